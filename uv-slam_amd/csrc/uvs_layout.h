// uvs_layout.h -- HBM layout of one sliding window ("blob") and of its solver workspace.
//
// Host packs a uvs_window into one contiguous blob (SoA, 8-byte aligned) that is
// read once per solve; the per-window workspace holds everything the persistent
// solve kernel spills outside LDS (landmark parameters cur/candidate, the
// Schur back-substitution store, whitened IMU Jacobians) and the outputs.
// All `d_*` offsets are in doubles, all `i_*` offsets in int32, from the blob
// base; `w_*` offsets are in doubles from the window's workspace base.
#pragma once
#include <stdint.h>

#define UVS_NF 11                 // frames in the window (WINDOW_SIZE + 1)
#define UVS_FD 16                 // padded per-frame dimension in the reduced system (15 + 1 dummy)
#define UVS_RD (UVS_NF * UVS_FD)  // 176: padded reduced dimension
#define UVS_NBLK (UVS_NF * (UVS_NF + 1) / 2)   // 66 lower 16x16 blocks
#define UVS_NBLKX (UVS_NBLK + UVS_NF + 1 + UVS_NF + 2)      // + the gather blocks of the two pseudo frames: 11 = time offset (ESTIMATE_TD): (td, f) = 66 + f, (td, td) = 77;
                                               // 12 = camera extrinsic (ESTIMATE_EXTRINSIC): (ex, f) = 78 + f, (ex, td) = 89, (ex, ex) = 90
#define UVS_EX_INDEX(a) (16 * (a) + 15)        // the 6 dofs of para_Ex_Pose sit in the spare 16th slots of frames 0..5 -- or, in a window with relocalization
                                               // blocks (estimator.cpp:944-978; only with a fixed extrinsic), the 6 dofs of relo_Pose: pseudo frame 12 is then an
                                               // ORDINARY second frame of a point observation whose S rows / columns are scattered to these slots
#define UVS_RELO_FRAME (UVS_NF + 1)            // frame index of relo_Pose in pt_fj
#define UVS_RELO2_BLOCKROW (UVS_NF + 2)        // relocalization blocks in a window with a FREE extrinsic (DevWin::relo2): 6 + 6 (+ 1) dofs do not fit the 11 spare slots, so the
                                               // gather blocks of relo_Pose get a block row of their own, 13 -- (13, f) = 91 + f, (13, td) = 102, (13, ex) = 103, (13, 13) = 104 --
                                               // which the assembly writes to a side buffer in the workspace instead of S; relo_Pose is then eliminated from the reduced system by a
                                               // rank-6 update before the factorization (k_solve only).  pt_fj of a relocalization block stays UVS_RELO_FRAME.
#define UVS_NBLKX2 (UVS_NBLKX + UVS_NF + 3)    // 105: gather blocks including block row 13 (the large-window kernels carry its 14 blocks as a tail behind the canonical partial: uvs_large_kernel.h LG_R2)
#define UVS_RELO2_DOUBLES 2304                 // side buffer: R[6][176] | Rrr[36] | g_r[6] | hd_r[6] | sc_r[6] | D_r[6] | Minv[36] | mg[6] | Z[6][176] | dr[6]
#define UVS_XDIM 192                           // frame state vector: pose[11][7] sb[11][9] ex[7] td relo_pose[7] pad
#define UVS_TD_INDEX (UVS_RD - 1)              // para_Td sits in the spare 16th slot of the last frame (index 175 of the padded reduced system)
#define UVS_BLK_LD 17             // padded row stride of a 16x16 LDS block (bank-conflict padding)
#define UVS_BLK_SZ (16 * UVS_BLK_LD)            // 272 doubles
#define UVS_S_DOUBLES (UVS_NBLK * UVS_BLK_SZ)   // 17952 doubles = 143616 B

#define UVS_IMU_STRIDE 296        // doubles per IMU block in the blob: 20 header + packed jac 48 + cov 225 (+3 pad).  The whitening matrix W (225 doubles, written by the device: setup_window)
                                  // lives in the WORKSPACE since round 4 (DevWin::w_imu_w, UVS_IMU_WS doubles per block): 18 KB per window less to pack and to send over PCIe
#define UVS_IMU_JAC 20            // the five 3x3 blocks of the pre-integration Jacobian the factor reads (dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg), 9 doubles each
#define UVS_IMU_COV 68
#define UVS_IMU_WS 226            // workspace doubles per block for W (15 x 15 row-major + 1 pad)
// packed index of jacobian(R + i, C + j) for (R, C) in {(0,9) (0,12) (3,12) (6,9) (6,12)}  (integration_base.h O_P/O_R/O_V rows, O_BA/O_BG columns)
#define UVS_IMU_JIDX(R, C, i, j) (9 * ((R) == 0 ? ((C) == 9 ? 0 : 1) : (R) == 3 ? 2 : ((C) == 9 ? 3 : 4)) + 3 * (i) + (j))

#define UVS_PT_REC 30             // LDS record per point observation: r[2] A[12] c|rc[2] B[12] rc[2]  (rc = Schur-corrected residual);
                                  // even stride and even field offsets: every Jacobian row is 16-byte aligned for ds_read_b128
#define UVS_PT_A 2
#define UVS_PT_C 14
#define UVS_PT_B 16
#define UVS_PT_ENTRY_A 0x8000      // bit 15 of a DIRECT gather entry of a point chunk (the offsets use 15 bits): the entry is an observation's A^T A term, whose Schur-corrected residual
                                   // is read from the record's rc slot (UVS_PT_RC2) at A + 26; every other diagonal entry finds it 12 doubles behind its first operand (B + 12 = rc slot)
#define UVS_PT_RC2 28
#define UVS_PT_TD 30               // d r / d td (2 doubles) + 2 zero pads: only in ESTIMATE_TD records (34 doubles)
#define UVS_PT_REC_TD 34
#define UVS_PT_EX 34               // d r / d ex_pose (2 x 6) : only in ESTIMATE_EXTRINSIC records (46 doubles)
#define UVS_PT_REC_EX 46
#define UVS_LN_REC 34             // LDS record per line observation: rl|rc[2] Jp[3][6] rv|rc2 pad Jl[3][4]  (row 2 = vanishing-point row)
#define UVS_LN_JP 2               // pose-Jacobian rows at 2, 8, 14
#define UVS_LN_RV 20              // VP residual, later its Schur-corrected value
#define UVS_LN_JL 22              // line-parameter Jacobian rows at 22, 26, 30
#define UVS_LN_EY 26              // LDS doubles per line observation in the staged E = J_l^T J_p and Y = H_ll^-1 E (4 rows of 6 + 2 pad): with the natural 24 (48 dwords) the
                                  // observations fall into only FOUR bank classes of the 64-bank LDS (gcd(48, 64) = 16) and the 32 gather groups of a wave -- and the 48 stores per lane
                                  // of pass B2 -- collided 8-fold; 52 dwords give the 16 classes a 16-byte access can have (round 4)
#ifndef UVS_NT
#define UVS_NT 256                // threads per workgroup of the solve kernels
#endif
#if UVS_NT != 256 && !defined(UVS_ALLOW_EXPERIMENTAL_NT)
#error "the library is built with 256 threads per workgroup; only uvs_solve512.hip instantiates the persistent kernel with 512 (it defines UVS_ALLOW_EXPERIMENTAL_NT)"
#endif
#ifndef UVS_GLANES
#define UVS_GLANES 2                // lanes per gather group: 2 = three rows of the 6x6 block per lane, 1 = all six rows in one lane
#endif
#define UVS_GROWS (6 / UVS_GLANES)  // block rows held by one lane
#if UVS_NT > 256
#define UVS_GT 256                  // threads that hold gather accumulators: in the 512-thread build waves 4..7 (waves 0..3 evaluate observations; uvs_solve_kernel.h: ROLES)
#else
#define UVS_GT UVS_NT
#endif
#define UVS_NGRP (UVS_GT / UVS_GLANES)      // gather groups; each owns one 6x6 pose block or one part of a split one

#define UVS_CHUNK_INTS 8
struct DevWin {
    int32_t n_points, n_pt_obs, n_lines, n_ln_obs, n_imu, prior_n, prior_nb, n_chunks;
    int32_t pt_stride, ln_stride;     // SoA strides of the measurement arrays
    int32_t d_frames;                 // pose[11][7], sb[11][9], ex[7], td, relo_pose[7], pad  (UVS_XDIM doubles)
    int32_t d_invd;                   // [n_points]
    int32_t d_ptmeas;                 // 6 x pt_stride : pi_x pi_y pi_z pj_x pj_y pj_z
    int32_t d_ptvel;                  // ESTIMATE_TD only: 6 x pt_stride : vel_i.xy vel_j.xy td_i td_j
    int32_t td_on;                    // options.estimate_td
    int32_t ex_on;                    // options.estimate_extrinsic
    int32_t n_relo;                   // number of relocalization blocks among the n_pt_obs point observations
    int32_t i_pt_eidx;                // relo_on only: [n_pt_obs] the caller's observation index of each packed observation, -1 = relocalization block
    int32_t relo_on;                  // the window carries relocalization blocks: point observations with pt_fj == UVS_RELO_FRAME
    int32_t relo2;                    // ... in a window with a free extrinsic: relo_Pose is a second-level block (UVS_RELO2_BLOCKROW), its side buffer at w_relo2
    int32_t w_relo2;
    int32_t pt_rec;                   // doubles per point record in the LDS staging area (30, or 34 with the td Jacobian)
    int32_t pt_xslots;                // Schur slots per point landmark beyond its observations: anchor (+ td) (+ ex)
    int32_t d_line;                   // [n_lines][4]
    int32_t d_lnmeas;                 // 9 x ln_stride : sp xyz, ep xyz, vp xyz
    int32_t d_imu;                    // n_imu x UVS_IMU_STRIDE
    int32_t d_prior;                  // J0[n*n] r0[n] pad[n] x0[144]   (round 4: no transposed copy -- the solve reads J0 once, in setup_window)
    int32_t i_pt_lm, i_pt_fi, i_pt_fj, i_pt_beg;      // obs arrays + CSR begin[n_points+1]
    int32_t i_ln_lm, i_ln_fj, i_ln_vp, i_ln_beg;      // obs arrays + CSR begin[n_lines+1]
    int32_t i_imu;                    // [n_imu][2] : frame_i, skip
    int32_t i_prior;                  // kind[16] frame[16] size[16] idx[16] x0off[16] colmap[96] inverse colmap[176] touched S blocks[66]
    int32_t i_chunks;                 // [n_chunks][UVS_CHUNK_INTS] : type(0 pt,1 ln), lm_begin, lm_end, offset of the chunk's gather lists in i_lists, their length, 0, first observation, observations
                                      // (the last two save the kernel two dependent loads from the CSR arrays at the head of every chunk: the whole descriptor is ONE 32-byte scalar load)
    int32_t i_wblk;                   // [UVS_NGRP] gather group -> pose block id | 256 (diagonal block) | part << 9 (4 bits, split blocks) | fa << 13 | fb << 17 | (parts - 1) << 21 (the parts of a block are consecutive groups); -1 = idle
    int32_t i_lists;                  // per chunk: schur_off[81] direct_off[81] entries[...]  (group-major, see pack_window in uvs_solver.hip)
    // workspace
    int32_t w_invd0, w_invd1, w_line0, w_line1;       // landmark parameters, two buffers (current / candidate)
    int32_t w_ltrig0, w_ltrig1;                        // sin/cos of the four orthonormal line angles, [n_lines][8], one per parameter buffer (k_solve only)
    int32_t w_scale_pt, w_scale_ln;                   // Jacobi scales of landmark parameters
    int32_t w_pt_E, w_pt_x;           // Einv store 6*(n_pt_obs+n_points) ; per point {ginv, g, dd, 0}
    int32_t w_ln_Y, w_ln_x;           // Y store 24*n_ln_obs ; per line UVS_LN_X doubles {Hinv*g[4], g[4], dd[4], H[10]}
    int32_t w_imu;                    // per block: Jraw[450] Jw[450] rraw[15] rw[15] (pad 936)
    int32_t w_imu_w;                  // per block: W[225] = chol(cov^-1)^T, upper triangular (UVS_IMU_WS doubles), written once per solve by setup_window
    int32_t w_out;                    // final state: frames[UVS_XDIM] | inv_depth[n_points] | line_orth[4 n_lines] (k_solve; the large path reads the cur buffers)
    int32_t w_prior_h0;               // the prior's quadratic form, written by setup_window: H0 = J0^T J0 dense [n][n] | g0 = J0^T r0 at UVS_PH_G0 | c0 = r0^T r0 / 2 at UVS_PH_C0 | diag(H0) by S index [176] at UVS_PH_HD
    int32_t n_pblk;                   // pose blocks of S the prior touches (ids in i_prior + 352)
    int32_t w_gacc;                   // 512-thread build only: the gather accumulators of the last linearization, [24][UVS_GT] (what a re-damping continues from; the 256-thread build keeps them in registers)
    int32_t n_cimg, w_cimg;           // entries of H0 that are structurally non-zero in S (pairs of prior columns a, b whose S indices satisfy i >= j): int32 index into the dense n x n H0 [n_cimg],
                                      // then S offset [n_cimg] -- in the WORKSPACE (w_cimg, as ints), generated by setup_window from the prior's column map (round 4: it was 21 KB of every blob)
    int32_t ws_doubles;
    int32_t blob_bytes;
    int32_t cur_sel;                  // written by the kernel: which landmark buffer holds the final state
    int32_t max_chunk_doubles;        // LDS doubles the fullest chunk occupies in the staging area (records + Schur factors + lists; <= UVS_S_DOUBLES; informational)
    int32_t n_parts;                  // largest number of parts any pose block is split into (informational; gacc_gather_parts sums them in one step)
    int32_t chol_half_ok;             // 1: blocks (i, j), j < i-1, of the reduced system are non-zero in rows {0..5, 15} only (true unless the prior keeps the speed / bias of a frame >= 2): the Cholesky pairs their rows
    int32_t reserved0;                // (round 5's opt-in dense landmark path lived behind this flag: tools/experiments/r05_dense_landmark_path.patch; always 0)
    int64_t out_host;                 // uvs_batch_stream: address (in the device's view) of this window's slot in the pinned result buffer of the host -- k_solve writes the final state there as well, so that
                                      // no gather kernel and no device-to-host copy follow the solve; 0 = none.  Patched into the staged header by upload_windows, not by pack_window.
    int32_t redamp_ok;                // 1: k_solve may re-damp the last linearization after a rejected step instead of linearizing again (no pseudo-frame blocks; every line chunk has room for the tables)
};

#define UVS_PH_G0(n) ((((n) * (n)) + 1) & ~1)
#define UVS_PH_C0(n) (UVS_PH_G0(n) + UVS_MAX_PRIOR_DIM)
#define UVS_PH_HD(n) (UVS_PH_C0(n) + 8)
#define UVS_PH_DOUBLES(n) (UVS_PH_HD(n) + UVS_RD)
#define UVS_WIMU_STRIDE 936
#define UVS_LN_X 24               // workspace doubles per line: Hinv g [4] | g [4] | damping [4] | undamped H = J_l^T J_l, lower packed [10] | 2 spare
