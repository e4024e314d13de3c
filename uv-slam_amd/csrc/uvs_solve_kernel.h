// uvs_solve_kernel.h -- the persistent Levenberg-Marquardt kernel: ONE workgroup solves ONE window.
//
// Replaces ceres::Solve(options, &problem, &summary) at estimator.cpp:992 for the
// problem built at estimator.cpp:763-927 (prior + IMU + point + line + VP blocks,
// SPARSE_SCHUR + LEVENBERG_MARQUARDT, Ceres defaults).
//
// MI355X mapping:
//   * the whole LM loop runs inside one launch; nothing returns to the host between iterations;
//   * the 11 frame states (15 DoF each), the reduced 176x176 system (66 lower 16x16 blocks,
//     padded rows) and all LM bookkeeping live in LDS (~155 KiB of the CU's 160 KiB);
//   * residual blocks are evaluated one lane per block straight from coalesced SoA arrays;
//   * landmark Schur elimination is an LDS-tiled, OUTPUT-STATIONARY reduction: every entry of the
//     pose-pose system is owned by one lane that sums the landmarks of the staged chunk in a fixed
//     order (no atomics => bitwise reproducible);
//   * the reduced system is factored by a blocked right-looking Cholesky on 16x16 LDS blocks.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/uvs_solver.h"
#include "uvs_layout.h"
#include "uvs_factors.h"

namespace uvsdev {


// threadIdx.x behind an optimization barrier.  With the plain intrinsic the compiler hoists every per-lane address computation of the
// LM iteration (base + k * tid ...) out of the iteration loop, keeps ~50 of them alive across the whole loop -- the kernel sits at the
// 512-register cap, so they are spilled to scratch once and RELOADED from scratch at each use, 75 global-memory loads per iteration whose
// latency one wavefront per SIMD cannot hide.  Re-deriving them from an opaque value costs a few integer instructions instead.
#ifndef UVS_X_PLAIN_TID
static __device__ __forceinline__ int lane_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
#else
static __device__ __forceinline__ int lane_tid() { return threadIdx.x; }
#endif
static constexpr int NT = UVS_NT;              // threads per workgroup
static constexpr int NW = NT / 64;
// Wave roles of the 512-thread build (two waves per SIMD, 256 registers per lane).  One body for all eight waves does not fit: the gather accumulators
// (48 VGPRs per lane, live from the first chunk to a possible re-damping) and the observation temporaries of the evaluation passes together are what
// spilled 500 registers in the round-3 experiment.  So the linearization is split by wave: waves 0..3 (EVALUATORS, one per SIMD) run the observation
// passes, the per-landmark Schur preparation and the IMU tiles exactly as the four waves of the 256-thread build do; waves 4..7 (GATHERERS, again one
// per SIMD) own the 128 gather groups.  Every other phase (assembly, factorization, back substitution, cost) uses all eight waves.
static constexpr bool ROLES = NT > 256;
static constexpr int ET = ROLES ? 256 : NT;       // evaluator threads = stride of the evaluation loops
static constexpr int EW = ET / 64;                // evaluator waves
static constexpr int GT0 = NT - UVS_GT;           // first gatherer thread (0: every thread is both)
UVS_DEV int wave_uniform() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
UVS_DEV bool role_eval() { return !ROLES || wave_uniform() < EW; }
// Lane order of the LINE loop of the back substitution, which walks points and lines one after the other with one lane per landmark.  A window has fewer
// points / lines than the 512-thread build has lanes, so both families would sit on the low waves and run back to back; started at wave 4, the lines run
// beside the points: 269 k -> 178 k cycles per solve.  (The cost pass keeps its lines on the low waves: its last wave carries the raw IMU residuals, and with
// the lines on waves 4..7 as well that wave became the longest -- measured, observations + IMU 121 k -> 143 k cycles per solve.)
#ifndef UVS_X_NO_LINE_ROTATE
UVS_DEV int line_lane() { return ROLES ? (lane_tid() ^ 256) : lane_tid(); }
#else
UVS_DEV int line_lane() { return lane_tid(); }
#endif
#define UVS_COST_LINE_LANE tid
// Loads in flight per lane.  One wave per SIMD (NT = 256) hides memory latency only with its own independent loads, so the streaming loops batch
// several observations per lane; with two resident waves per SIMD (NT = 512) the other wave covers the latency and the batches shrink to what
// fits 256 registers per lane.
#ifndef UVS_T_CP_PB
#define UVS_T_CP_PB 2
#define UVS_T_CP_LB 1
#define UVS_T_BS_LNB 4
#define UVS_T_PR_UN 26
#endif
static_assert(UVS_PT_C == UVS_PT_A + 12, "pt_anchor_pass reads A and c as one run");
static constexpr int CP_PB = NT > 256 ? UVS_T_CP_PB : 4;      // cost pass: point observations per batch
static constexpr int CP_LB = NT > 256 ? UVS_T_CP_LB : 2;      // cost pass: line observations per batch
static constexpr int BS_LNB = NT > 256 ? UVS_T_BS_LNB : 4;     // back substitution: line observations per batch
static constexpr int PR_UN = NT > 256 ? UVS_T_PR_UN : 26;    // prior mat-vec: rows of H0 in flight (26: the 25 rows per part of the n = 75 prior are ONE round trip, not 24 + 1)
static constexpr int PG_UN = NT > 256 ? 16 : 32;    // prior gradient: rows of J0 in flight

// ---- LDS map (in doubles).  The small arrays come FIRST and the big region (reduced system / staging area) LAST, so that a kernel that needs only part of
// the region -- the landmark-sharded kernels: k_large_backsub none of it beyond a little scratch, k_large_chunks a chunk's worth -- asks for less dynamic LDS and
// leaves room for a second workgroup on the compute unit (round 4); every offset below is the same in all kernels.
static constexpr int L_X = 0;                  // pose[77] sb[99] ex[7] (+1 pad)
static constexpr int L_XC = L_X + UVS_XDIM;
static constexpr int L_G = L_XC + UVS_XDIM;    // gradient of the frame block, padded index space (176)
static constexpr int L_DLT = L_G + UVS_RD;     // rhs / step; [192..197] = copy of the relo_Pose step (pseudo frame 12) for the back-substitution
static constexpr int L_HD = L_DLT + UVS_RD + 24;   // diag(J^T J) of frame parameters (before Schur, before damping)
static constexpr int L_SC = L_HD + UVS_RD;     // Jacobi scaling s_k
static constexpr int L_DD = L_SC + UVS_RD;     // LM damping added to the diagonal
static constexpr int L_DINV = L_DD + UVS_RD;   // 1 / L_kk of the Cholesky factor
static constexpr int L_RF = L_DINV + UVS_RD;   // 11 rotation matrices (row-major) of the CURRENT evaluation point; slot 12 (offset 108) = relo_Pose's
static constexpr int L_EX = L_RF + 120;        // ric[9] tic[3]
static constexpr int L_PDX = L_EX + 16;        // prior dx
static constexpr int L_PR = L_PDX + UVS_MAX_PRIOR_DIM;   // prior: y = H0 dx of the current point (k_evaluate: the residual vector r0 + J0 dx)
static constexpr int L_PRC = L_PR + UVS_MAX_PRIOR_DIM;   // y = H0 dx at the CANDIDATE (becomes the current one when the step is accepted)
static constexpr int L_RED = L_PRC + UVS_MAX_PRIOR_DIM;  // reduction scratch
static constexpr int L_CTRL = L_RED + (5 * NW > 24 ? 5 * NW : 24);      // block_reduce uses 5 doubles per wave (an 8-wave build wrote waves 5..7 into the control words: the wrong final costs of the 512-thread experiments)
static constexpr int L_PROF = L_CTRL + 32;      // per-phase cycle counters (debug launches only)
static constexpr int L_WPROF = L_PROF + 24;     // debug sub-timers: [0..3] candidate-cost phase (stage + dx, prior residual, observations, IMU), [4..7] busy cycles of each wave in the Cholesky column phase
static constexpr int L_LCOST = L_WPROF + 8;     // per-lane cost accumulator of the current linearization (kept in LDS, not in a register that
static constexpr int LACC = 1;                  // (one slot per EVALUATOR lane: the gatherer waves of the 512-thread build evaluate nothing)
static constexpr int L_LGMAX = L_LCOST + ET / LACC;    // would have to live across every phase) ; max |g_landmark| per WAVE (a maximum does not care how it is grouped: 8 doubles instead of one per lane)
static constexpr int L_PX0 = L_LGMAX + 8;              // the prior's linearization point, 9 doubles per kept block, and its block table (kind, frame, size, column offset: 4 x 16 ints) -- staged
static constexpr int L_PTAB = L_PX0 + 9 * UVS_MAX_PRIOR_BLOCKS;      // once per solve by k_solve (stage_prior_tables): prior_dx runs 21 times per solve and its global round trip was ~2.5 k cycles each
static constexpr int L_SMALL = (L_PTAB + 2 * UVS_MAX_PRIOR_BLOCKS + 1) & ~1;      // end of the small arrays (even: the region below holds 16-byte rows)
static constexpr int L_S = L_SMALL;            // the reduced system S (66 lower 16 x 16 blocks, 17-double rows) / the staging area of a landmark chunk / scratch of the frame phases
static constexpr int L_TOTAL = L_S + UVS_S_DOUBLES;
static_assert(L_TOTAL * 8 <= 160 * 1024, "LDS map exceeds the 160 KB of a CU");
enum { P_SETUP = 0, P_OBS, P_LMPREP, P_GATHER, P_ASSEMBLE, P_CHOL, P_TRSV, P_BACKSUB, P_COST, P_MISC, P_CH_DIAG, P_CH_PANEL, P_CH_TRAIL, P_AS_IMU, P_AS_ZERO, P_AS_ADD, P_LAST };
#define UVS_PROF(c, k) do { if ((c).o.debug && lane_tid() == 0) { const long long now_ = clock64(); (c).sh[L_PROF + (k)] += (double)(now_ - (long long)(c).sh[L_PROF + 23]); (c).sh[L_PROF + 23] = (double)now_; } } while (0)
static constexpr size_t LDS_BYTES = (size_t)L_TOTAL * 8;
// debug == 5 (UVS_DEBUG_LIN_TIMELINE, first window of a launch): every wave logs (stamp id, clock) pairs of the linearization's inner steps; tools/lin_timeline.py prints them
static constexpr int TL_PER_WAVE = 4096;
__device__ long long g_lin_tl[8 * TL_PER_WAVE * 2];
#define UVS_TLOG(c, id) do { if ((c).o.debug == 5 && (threadIdx.x & 63) == 0 && blockIdx.x == 0) { const int w_ = threadIdx.x >> 6; const int n_ = (int)(c).sh[L_WPROF + w_]; \
    if (n_ < TL_PER_WAVE) { g_lin_tl[2 * (w_ * TL_PER_WAVE + n_)] = (id); g_lin_tl[2 * (w_ * TL_PER_WAVE + n_) + 1] = clock64(); (c).sh[L_WPROF + w_] = (double)(n_ + 1); } } } while (0)

enum { C_COST = 0, C_RADIUS, C_DECR, C_XNORM, C_GMAX, C_CAND, C_MCC, C_STEP2, C_XC2, C_GO, C_IT, C_INVALID, C_CUR, C_FIRST,
       C_TERM, C_NSUCC, C_CHOLOK, C_GMAXLM, C_TIMEUP };

struct KOpts {   // device copy of uvs_options
    int max_it, ex_free, keep_cand, jacobi;
    double sqrt_info, line_factor, vp_factor, loss_pt, loss_ln, loss_vp, G[3];
    double r0, rmax, rmin, min_rel, dlo, dhi, ftol, gtol, ptol;
    int max_invalid;
    int debug;
    long long max_ticks;      // options.max_solver_time_in_seconds in ticks of the 100 MHz wall clock (wall_clock64); 0 = no cap
    int redamp;      // 1 (default): a rejected step is followed by a re-damping of the stored linearization; 0 (UVS_REDAMP=0 at uvs_create): by a new linearization
};

__constant__ unsigned char c_blk_fa[UVS_NBLK];
__constant__ unsigned char c_blk_fb[UVS_NBLK];

UVS_DEV int sidx(int i, int j) {   // i >= j, padded index space
    const int bi = i >> 4, bj = j >> 4;
    return (((bi * (bi + 1)) >> 1) + bj) * UVS_BLK_SZ + (i & 15) * UVS_BLK_LD + (j & 15);
}
UVS_DEV double bcast_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// LDS traffic of ONE wave is processed in order; this only stops the compiler from moving accesses across the hand-over point
UVS_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

UVS_DEV double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
UVS_DEV double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
// sqrt(x) and 1/sqrt(x) together from the hardware seed (v_rsq_f64) + two Goldschmidt steps (FMA only, no divide):
// a few ulp, which is all the Cholesky pivots need; x <= 0 or non-finite yields NaN and is caught by the caller.
UVS_DEV void rsqrt_pair(double x, double* sq, double* rsq) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, hh = 0.5 * y;
    double r = fma(-hh, g, 0.5);
    g = fma(g, r, g); hh = fma(hh, r, hh);
    r = fma(-hh, g, 0.5);
    g = fma(g, r, g); hh = fma(hh, r, hh);
    const double d = fma(-g, g, x);
    g = fma(d, hh, g);
    *sq = g; *rsq = 2.0 * hh;
}
// Deterministic block reductions of up to 4 sums + 1 max.  Result broadcast to every thread.
UVS_DEV void block_reduce(double* sh, double* s /*[4]*/, double* mx) {
    const int tid = lane_tid(), lane = tid & 63, wv = tid >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) { const double t = wave_sum(s[k]); if (lane == 0) sh[L_RED + wv * 5 + k] = t; }
    { const double t = wave_max(*mx); if (lane == 0) sh[L_RED + wv * 5 + 4] = t; }
    __syncthreads();
    // every lane adds the NW wave partials itself (broadcast LDS reads, wave order => the same sum everywhere): no single-lane stage and
    // no third barrier; the entry barrier of the next call keeps these reads ahead of its writes
    double a[4] = {0, 0, 0, 0}, m = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += sh[L_RED + w * 5 + k];
        m = fmax(m, sh[L_RED + w * 5 + 4]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = a[k];
    *mx = m;
}

// per-lane (or per lane pair) accumulators of the running linearization
UVS_DEV void lacc_set(double* sh, double cost, double gmax) {
    const int tid = lane_tid();
    if (tid < ET) sh[L_LCOST + tid] = cost;
    const double wm = wave_max(gmax);
    if ((tid & 63) == 0) sh[L_LGMAX + (tid >> 6)] = wm;
}
UVS_DEV void lacc_add(double* sh, double cost, double gmax) {
    const int tid = lane_tid();
    if (tid < ET) sh[L_LCOST + tid] += cost;
    const double wm = wave_max(gmax);
    if ((tid & 63) == 0) sh[L_LGMAX + (tid >> 6)] = fmax(sh[L_LGMAX + (tid >> 6)], wm);
}
UVS_DEV double lacc_cost(const double* sh) { const int tid = lane_tid(); return tid < ET ? sh[L_LCOST + tid] : 0.0; }
UVS_DEV double lacc_gmax(const double* sh) { return sh[L_LGMAX + (lane_tid() >> 6)]; }

struct Ctx {
    const DevWin* hdr;
    const double* bd;      // blob as doubles
    const int* bi;         // blob as ints
    double* ws;            // workspace of this window
    double* sh;            // LDS
    KOpts o;
    int ltrig_ok = 0;      // the sin/cos cache of the line parameters (w_ltrig0/1) is maintained (k_solve) -- the step-wise large-window kernels leave it off
    int ptab_ok = 0;       // the prior's block table and linearization point are staged in LDS (L_PTAB / L_PX0: k_solve) -- the other kernels read them from the blob
};

// sin/cos cache that belongs to the line-parameter buffer `line` (nullptr: compute on the fly)
UVS_DEV const double* line_trig_of(const Ctx& c, const double* line) {
    const DevWin& h = *c.hdr;
    if (!c.ltrig_ok) return nullptr;
    return line == c.ws + h.w_line0 ? c.ws + h.w_ltrig0 : (line == c.ws + h.w_line1 ? c.ws + h.w_ltrig1 : nullptr);
}

// rotation matrices of the evaluation point `x` (LDS, L_X or L_XC layout) -> L_RF / L_EX
UVS_DEV void stage_rotations(const Ctx& c, const double* x) {
    const int tid = lane_tid();
    if (tid < UVS_NF) quat_to_R(x + 7 * tid + 3, c.sh + L_RF + 9 * tid);
    else if (tid == UVS_NF) { quat_to_R(x + 176 + 3, c.sh + L_EX); c.sh[L_EX + 9] = x[176]; c.sh[L_EX + 10] = x[177]; c.sh[L_EX + 11] = x[178]; }
    else if (tid == UVS_RELO_FRAME && c.hdr->relo_on) quat_to_R(x + 184 + 3, c.sh + L_RF + 9 * UVS_RELO_FRAME);
}
// pose block of frame f in a state vector; f == UVS_RELO_FRAME is relo_Pose (its rotation sits in slot 12 of L_RF, so RF + 9 f needs no select)
UVS_DEV const double* pose_of(const double* x, int f) { return x + (f > UVS_NF ? 184 : 7 * f); }

// measurements of point residual block `o`; with ESTIMATE_TD the time-shifted ones of ProjectionTdFactor (projection_td_factor.cpp:51-52):
//   pts_i_td = pts_i - (td - td_i) * (vel_i, 0), same for j  (rolling-shutter term folded into td_i / td_j by the caller).  vij = vel_i.xy, vel_j.xy
UVS_DEV void load_point_obs(const Ctx& c, int o, double td, double* pi, double* pj, double* vij) {
    const DevWin& h = *c.hdr;
    const double* m = c.bd + h.d_ptmeas + o; const int st = h.pt_stride;
    pi[0] = m[0]; pi[1] = m[st]; pi[2] = m[2 * st]; pj[0] = m[3 * st]; pj[1] = m[4 * st]; pj[2] = m[5 * st];
    if (h.td_on) {
        const double* v = c.bd + h.d_ptvel + o;
        vij[0] = v[0]; vij[1] = v[st]; vij[2] = v[2 * st]; vij[3] = v[3 * st];
        const double di = td - v[4 * st], dj = td - v[5 * st];
        pi[0] -= di * vij[0]; pi[1] -= di * vij[1]; pj[0] -= dj * vij[2]; pj[1] -= dj * vij[3];
    }
}

// ------------------------------------------------------------------ residual-only cost at `x` (LDS) / landmark buffer `sel`
UVS_DEV void stage_prior_tables(const Ctx& c) {      // once per solve; the caller's next barrier precedes the first prior_dx
    const int tid = lane_tid();
    const DevWin& h = *c.hdr;
    if (h.prior_n <= 0) return;
    int* tab = (int*)(c.sh + L_PTAB);
    if (tid < 4 * UVS_MAX_PRIOR_BLOCKS) tab[tid] = c.bi[h.i_prior + tid];      // kind[16] frame[16] size[16] idx[16]
    if (tid < 9 * UVS_MAX_PRIOR_BLOCKS) c.sh[L_PX0 + tid] = c.bd[h.d_prior + h.prior_n * h.prior_n + 2 * h.prior_n + tid];
}
UVS_DEV void prior_dx(const Ctx& c, const double* x) {
    const int tid = lane_tid() - 64;      // the lanes of the SECOND wave: its callers stage the rotations on the first lanes of wave 0 in the same breath, and one wave would run the two one after the other
    const DevWin& h = *c.hdr;
    if (h.prior_n > 0 && tid >= 0 && tid < h.prior_nb) {
        int kind, frame, size, idx;
        double x0[9];
        if (c.ptab_ok) {
            const int* tab = (const int*)(c.sh + L_PTAB);
            kind = tab[tid]; frame = tab[16 + tid]; size = tab[32 + tid]; idx = tab[48 + tid];
#pragma unroll
            for (int k = 0; k < 9; ++k) x0[k] = c.sh[L_PX0 + 9 * tid + k];
        } else {
            const int* pt = c.bi + h.i_prior;
            kind = pt[tid]; frame = pt[16 + tid]; size = pt[32 + tid]; idx = pt[48 + tid];
            const double* x0g = c.bd + h.d_prior + h.prior_n * h.prior_n + 2 * h.prior_n + 9 * tid;      // stride 9 per block (pack_window): independent of the table
#pragma unroll
            for (int k = 0; k < 9; ++k) x0[k] = x0g[k];
        }
        const double* xb = (kind == UVS_BLOCK_POSE) ? x + 7 * frame : (kind == UVS_BLOCK_SPEEDBIAS) ? x + 77 + 9 * frame : (kind == UVS_BLOCK_TD) ? x + 183 : x + 176;
        double* dx = c.sh + L_PDX + idx;
        if (size != 7) { for (int k = 0; k < size; ++k) dx[k] = xb[k] - x0[k]; }
        else {   // marginalization_factor.cpp:352-362
            dx[0] = xb[0] - x0[0]; dx[1] = xb[1] - x0[1]; dx[2] = xb[2] - x0[2];
            double qi[4], e[4]; quat_inv(x0 + 3, qi); quat_mul(qi, xb + 3, e);
            const double sg = (e[3] >= 0.0) ? 2.0 : -2.0;
            dx[3] = sg * e[0]; dx[4] = sg * e[1]; dx[5] = sg * e[2];
        }
    }
}
// The prior inside the solve (marginalization_factor.cpp:333-381: r = r0 + J0 dx, Jacobian J0 constant) in its QUADRATIC form: with H0 = J0^T J0,
// g0 = J0^T r0, c0 = r0^T r0 / 2 (built once per solve by setup_window, in the workspace)
//     cost = c0 + g0 . dx + dx . (H0 dx) / 2 ,      gradient = g0 + H0 dx ,
// so ONE symmetric n x n mat-vec y = H0 dx serves the cost of an evaluation point and -- when that point is linearized -- the gradient; the
// n x n Jacobian is read once per solve instead of twice per LM iteration (J0^T for the residual, J0 for J0^T r: 2 x 45 KB of a window's
// 350 KB working set, and the transposed copy leaves the blob).  After prior_dx + barrier; fills sh[dst .. dst + n) with y and returns this
// thread's share of the prior cost.  Every row's dot product is split over up to 4 lanes with PR_UN loads in flight each (H0 comes from
// L2 / HBM on every evaluation; a 75-deep dependent chain would pay the memory latency 75 times).  Partials meet in the (dead) S region;
// contains one workgroup barrier, so all threads must call it.
UVS_DEV double prior_quad(const Ctx& c, int dst = L_PR) {
    const DevWin& h = *c.hdr;
    const int n = h.prior_n, tid = lane_tid();
    double cost = 0.0;
    if (n <= 0) return cost;
    const double* H0 = c.ws + h.w_prior_h0;
    const int parts = (NT / n) < 4 ? (NT / n) : 4;
    const int part = tid / n, row = tid - part * n;
    // g0 / c0 are requested with the rows of H0, not after the barrier below (where they were a second memory round trip with every wave waiting)
    const double g0v = tid < n ? c.ws[h.w_prior_h0 + UVS_PH_G0(n) + tid] : 0.0, c0v = tid == 0 ? c.ws[h.w_prior_h0 + UVS_PH_C0(n)] : 0.0;
    if (part < parts) {
        const int kb = (n * part) / parts, ke = (n * (part + 1)) / parts;
        double p8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = kb; k < ke; k += PR_UN) {
            double jv[PR_UN];
#pragma unroll
            for (int u = 0; u < PR_UN; ++u) jv[u] = H0[(k + u < ke ? k + u : kb) * n + row];      // H0 is symmetric: row k, entry `row` -- consecutive lanes, consecutive addresses (the packed lower triangle, 23 instead of 45 KB per evaluation, was measured: its scattered upper-half reads cost more than the bytes, -0.6 % on the batch)
#pragma unroll
            for (int u = 0; u < PR_UN; ++u) if (k + u < ke) p8[u & 7] += jv[u] * c.sh[L_PDX + k + u];
        }
        c.sh[L_S + 128 * part + row] = ((p8[0] + p8[1]) + (p8[2] + p8[3])) + ((p8[4] + p8[5]) + (p8[6] + p8[7]));
    }
    __syncthreads();
    if (tid < n) {
        double y = c.sh[L_S + tid];
        for (int p = 1; p < parts; ++p) y += c.sh[L_S + 128 * p + tid];
        c.sh[dst + tid] = y;
        cost = c.sh[L_PDX + tid] * (g0v + 0.5 * y);
        if (tid == 0) cost += c0v;
    }
    return cost;
}
// this thread's share of the prior cost at the point whose dx is staged in L_PDX and whose y = H0 dx sits in sh[src ..)
UVS_DEV double prior_cost_share(const Ctx& c, int src) {
    const DevWin& h = *c.hdr;
    const int n = h.prior_n, tid = lane_tid();
    double cost = 0.0;
    if (tid < n) {
        cost = c.sh[L_PDX + tid] * (c.ws[h.w_prior_h0 + UVS_PH_G0(n) + tid] + 0.5 * c.sh[src + tid]);
        if (tid == 0) cost += c.ws[h.w_prior_h0 + UVS_PH_C0(n)];
    }
    return cost;
}
// The residual vector itself, r = r0 + J0 dx (marginalization_factor.cpp:364), for uvs_evaluate / the host marginalization path: one lane per row,
// J0 read row-wise (not a hot path).  After prior_dx + barrier; fills L_PR, returns this thread's 0.5 r^2.
UVS_DEV double prior_residual_rows(const Ctx& c) {
    const DevWin& h = *c.hdr;
    const int n = h.prior_n, tid = lane_tid();
    double cost = 0.0;
    if (tid < n) {
        const double* Jr = c.bd + h.d_prior + (size_t)tid * n;
        double s = c.bd[h.d_prior + n * n + tid];
        for (int k = 0; k < n; ++k) s += Jr[k] * c.sh[L_PDX + k];
        c.sh[L_PR + tid] = s;
        cost = 0.5 * s * s;
    }
    return cost;
}

// cost of all residual blocks at the point staged in (x, RF/EX) with landmark buffers invd / line
// PB / LB: point / line observations per lane and batch (loads in flight); the defaults suit the workgroup size of the persistent kernel, a kernel that runs
// several workgroups per compute unit passes smaller ones (its other waves hide the latency, its register budget is smaller)
// raw IMU residual of block (lane - lane0) -> scratch in the S region (cost_pass whitens them after its barrier)
UVS_DEV void cost_imu_raw(const Ctx& c, const double* x, int lane0) {
    const DevWin& h = *c.hdr;
    const int b = lane_tid() - lane0;
    double* rs = c.sh + L_S + 1024;
    if (b >= 0 && b < h.n_imu) {
        const int fi = c.bi[h.i_imu + 2 * b], skip = c.bi[h.i_imu + 2 * b + 1];
        if (!skip) {
            const double* blk = c.bd + h.d_imu + (size_t)b * UVS_IMU_STRIDE;
            double r[15];
            imu_raw(blk, blk + UVS_IMU_JAC, c.o.G, x + 7 * fi, x + 77 + 9 * fi, x + 7 * (fi + 1), x + 77 + 9 * (fi + 1), r, nullptr);
#pragma unroll
            for (int i = 0; i < 15; ++i) rs[16 * b + i] = r[i];
        }
    }
}
template <int PB = CP_PB, int LB = CP_LB>
UVS_DEV double cost_pass(const Ctx& c, const double* x, const double* invd, const double* line, int po0, int po1, int lo0, int lo1, bool with_imu) {
    const DevWin& h = *c.hdr;
    const int tid = lane_tid();
    const double* RF = c.sh + L_RF; const double* ric = c.sh + L_EX; const double* tic = c.sh + L_EX + 9;
    const double* ltrig = line_trig_of(c, line);
    double cost = 0.0;
    long long tq_ = clock64();
#define UVS_CQ(slot) if (c.o.debug == 1 && tid == 0) { const long long t_ = clock64(); c.sh[L_WPROF + slot] += (double)(t_ - tq_); tq_ = t_; }
    // 512-thread build: the raw IMU residuals (one lane per block, a long serial chain of quaternion algebra) go to the LAST wave and run FIRST: that wave has
    // half the observations of waves 0..3 (one point, no line), so the chain that used to follow the observations on wave 0 -- with seven waves waiting at the
    // barrier below -- now runs beside them
    const int imu_lane0 = ROLES ? NT - 64 : 0;
#ifndef UVS_X_NO_IMU_WAVE
    if (ROLES && with_imu) cost_imu_raw(c, x, imu_lane0);
#endif
    // Observations in batches of four per lane: the index loads and the measurement loads of a batch go out together and the
    // landmark parameters (the only loads whose address depends on an index) follow as a second group, so a lane pays two HBM/L2
    // round trips per BATCH instead of two per observation (one wave per SIMD: nothing else hides that latency).
    for (int o0 = po0 + tid; o0 < po1; o0 += PB * NT) {
        int lm[PB], fi[PB], fj[PB]; bool in[PB];
        double pi[PB][3], pj[PB][3], vij[PB][4], idp[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int o = o0 + u * NT; in[u] = o < po1;
            const int oo = in[u] ? o : o0;
            lm[u] = c.bi[h.i_pt_lm + oo]; fi[u] = c.bi[h.i_pt_fi + oo]; fj[u] = c.bi[h.i_pt_fj + oo];
            load_point_obs(c, oo, x[183], pi[u], pj[u], vij[u]);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) idp[u] = invd[lm[u]];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (!in[u]) continue;
            double r[2];
            point_eval<false, false>(x + 7 * fi[u], RF + 9 * fi[u], pose_of(x, fj[u]), RF + 9 * fj[u], ric, tic, idp[u], pi[u], pj[u], c.o.sqrt_info, r, nullptr, nullptr, nullptr, nullptr);
            double sc; cost += 0.5 * cauchy(c.o.loss_pt, r[0] * r[0] + r[1] * r[1], &sc);
        }
    }
    // lines + vp, two per batch
    for (int o0 = lo0 + UVS_COST_LINE_LANE; o0 < lo1; o0 += LB * NT) {
        int lm[LB], fj[LB], hv[LB]; bool in[LB];
        double ms[LB][9], lp[LB][4] = {}, tg[LB][8];
        const int st = h.ln_stride;
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int o = o0 + u * NT; in[u] = o < lo1;
            const int oo = in[u] ? o : o0;
            lm[u] = c.bi[h.i_ln_lm + oo]; fj[u] = c.bi[h.i_ln_fj + oo]; hv[u] = c.bi[h.i_ln_vp + oo];
            const double* m = c.bd + h.d_lnmeas + oo;
#pragma unroll
            for (int q = 0; q < 9; ++q) ms[u][q] = m[q * st];
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            if (ltrig) {
#pragma unroll
                for (int q = 0; q < 8; ++q) tg[u][q] = ltrig[8 * lm[u] + q];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) lp[u][q] = line[4 * lm[u] + q];
            }
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            if (!in[u]) continue;
            const double sp[3] = {ms[u][0], ms[u][1], ms[u][2]}, ep[3] = {ms[u][3], ms[u][4], ms[u][5]}, vp[3] = {ms[u][6], ms[u][7], ms[u][8]};
            LineGeom g;
            line_geom<false>(x + 7 * fj[u], x + 7 * fj[u] + 3, RF + 9 * fj[u], ric, tic, lp[u], g, ltrig ? tg[u] : nullptr);
            double r[2], sc;
            line_residual<false>(g, sp, ep, c.o.line_factor, r, nullptr, nullptr);
            cost += 0.5 * cauchy(c.o.loss_ln, r[0] * r[0] + r[1] * r[1], &sc);
            if (hv[u]) { double rv; vp_residual<false>(g, vp, c.o.vp_factor, &rv, nullptr, nullptr); cost += 0.5 * cauchy(c.o.loss_vp, rv * rv, &sc); }
        }
    }
    UVS_CQ(2)
    // IMU: the raw 15-vector by one lane per block (serial quaternion algebra), then the 15x15 upper-triangular whitening with one lane
    // per (block, row): 120 dependent-latency loads of W per block no longer sit on a single lane.  Scratch: the S region, which holds
    // nothing live between the triangular solve and the next linearization (prior_residual uses its first 512 doubles the same way).
    if (with_imu) {
        double* rs = c.sh + L_S + 1024;
#ifndef UVS_X_NO_IMU_WAVE
        if (!ROLES)
#endif
        cost_imu_raw(c, x, 0);
        __syncthreads();
        const int b = tid >> 4, i = tid & 15;
        if (b < h.n_imu && i < 15 && !c.bi[h.i_imu + 2 * b + 1]) {
            const double* W = c.ws + h.w_imu_w + (size_t)b * UVS_IMU_WS + i * 15;
            double wv[15];
#pragma unroll
            for (int k = 0; k < 15; ++k) wv[k] = (k >= i) ? W[k] : 0.0;
            double sacc = 0.0;
#pragma unroll
            for (int k = 0; k < 15; ++k) sacc += wv[k] * rs[16 * b + k];
            cost += 0.5 * sacc * sacc;
        }
    }
    UVS_CQ(3)
    return cost;
}

// ------------------------------------------------------------------ blocked Cholesky of the LDS-resident reduced system
// S (lower 16x16 blocks) <- L ; strictly-upper part of every diagonal block <- (L_kk^-1)^T ; L_DINV <- 1/diag(L).
// Returns via CTRL[C_CHOLOK].
typedef double d4_t __attribute__((ext_vector_type(4)));

UVS_DEV double* sblk(double* sh, int i, int j) { return sh + L_S + (((i * (i + 1)) >> 1) + j) * UVS_BLK_SZ; }

// 1/x from the hardware seed (v_rcp_f64) + two Newton steps: the pivot reciprocal sits on the serial chain of the factorization.
UVS_DEV double rcp_newton(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0); y = fma(y, e, y);
    e = fma(-x, y, 1.0); y = fma(y, e, y);
    return y;
}

// LEFT-LOOKING blocked Cholesky with one column of LOOK-AHEAD:  S = L L^T in place, and the right-hand side carried as block
// row 11 (one real row), so L_DLT leaves as y = L^-1 rhs.  Wave 0 owns the serial part, the other waves ("workers") everything else:
//   wave 0, column k:  the diagonal block's last term (j = k-1) from the registers that still hold L(k, k-1)^T;  the 16 x 16 block factored
//       WITHOUT leaving the MFMA C layout, two pivots per link (one rank-2 MFMA), a second MFMA per link running the same elimination on
//       the identity => W = L_kk^-1; square roots only after the chain;  then the panel product of the block below, L(k+1, k)^T = W S(k+1, k)^T;
//   workers, column k: last terms of the blocks (i, k), i >= k+2, kept in registers (transposed) until W_k is published, then their panel
//       products L_ik = S_ik W^T (4 MFMAs instead of a 16-step substitution);  in between the LOOK-AHEAD: the terms j < k of block column k+1.
//       Blocks (i, j), j < i-1, have non-zero rows {0..5, 15} only, so two of them share every MFMA (chol_hr below).
//   No workgroup barrier inside the factorization (half-row path): LDS flags carry the four dependencies (W_k, S(k+1, k) complete, the next
//       diagonal block's look-ahead complete, L(k+1, k) stored); the workers meet at an LDS counter.
// C/D layout of the f64 MFMA: row = (lane >> 4) + 4 * reg, col = lane & 15;  A[i][k]: lane i + 16k;  B[k][j]: lane j + 16k.
struct MiniCtx { double* sh; struct { int debug; } o; };      // what UVS_PROF needs inside the dense-solve phases
template <bool TR = false>
UVS_DEV void chol_update_item(double* sh, int i, int cc, int j0, int j1, int lane, d4_t& acc) {
    // acc -= sum_{j in [j0, j1)} L_ij L_cj^T   (i == UVS_NF: the right-hand-side row, y_j^T in L_DLT);  TR: acc -= sum L_cj L_ij^T, the TRANSPOSED
    // block (the two MFMA operands have the same load pattern, so transposing the product is swapping them)
    const int li = lane & 15, lk = lane >> 4;
    const bool rhs = (i == UVS_NF);
    const double* Bj = sblk(sh, cc, j0) + li * UVS_BLK_LD + lk;
    const double* Ai = rhs ? sh + L_DLT + 16 * j0 + lk : sblk(sh, i, j0) + li * UVS_BLK_LD + lk;
    const int astep = rhs ? 16 : UVS_BLK_SZ;
    if (j0 >= j1) return;
    // Operands of term j+1 are fetched from LDS while the four 64-cycle MFMAs of term j run; two register sets in ping-pong (no copies).
    // Nothing touches a loaded value before its MFMA: the sign is the MFMA's own neg modifier (blgp bit 0 = -A for the f64 shapes), and
    // the right-hand-side row needs no masking -- row i of the product depends on row i of A only, the row-0 result is the only one stored,
    // and whatever accumulates in the other rows is never read.  (A negation or select on the prefetched registers makes the compiler
    // wait for the prefetch before issuing the current term's MFMAs: 437 instead of ~260 cycles per term.)
    // ONE accumulator chain: back-to-back MFMAs that feed their own result as SrcC issue at full rate, while a second chain makes the
    // compiler shuttle it between AGPRs and VGPRs across the loop back-edge (a pipeline drain plus 16 moves every other term).
    double a0[4], b0[4], a1[4], b1[4];
#define UVS_CH_LOAD(AV, BV) { _Pragma("unroll") for (int q = 0; q < 4; ++q) { AV[q] = Ai[4 * q]; BV[q] = Bj[4 * q]; } Bj += UVS_BLK_SZ; Ai += astep; }
#define UVS_CH_MFMA(AV, BV) { _Pragma("unroll") for (int q = 0; q < 4; ++q) acc = TR ? __builtin_amdgcn_mfma_f64_16x16x4f64(BV[q], AV[q], acc, 0, 0, 1) : __builtin_amdgcn_mfma_f64_16x16x4f64(AV[q], BV[q], acc, 0, 0, 1); }
    UVS_CH_LOAD(a0, b0)
    for (int j = j0;;) {
        if (j + 1 < j1) UVS_CH_LOAD(a1, b1)
        UVS_CH_MFMA(a0, b0)
        if (++j >= j1) break;
        if (j + 1 < j1) UVS_CH_LOAD(a0, b0)
        UVS_CH_MFMA(a1, b1)
        if (++j >= j1) break;
    }
#undef UVS_CH_LOAD
#undef UVS_CH_MFMA
}
// C-layout load / store of block (i, cc); the diagonal block is symmetrised from its stored lower triangle; the rhs row lives in L_DLT
UVS_DEV d4_t chol_load_item(double* sh, int i, int cc, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    d4_t acc;
    if (i == UVS_NF) { const double bv0 = sh[L_DLT + 16 * cc + li]; acc[0] = (lk == 0) ? bv0 : 0.0; acc[1] = 0.0; acc[2] = 0.0; acc[3] = 0.0; }
    else if (i == cc) {
        const double* Dk = sblk(sh, cc, cc);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int r = lk + 4 * q; const int hi_ = r >= li ? r : li, lo_ = r >= li ? li : r; acc[q] = Dk[hi_ * UVS_BLK_LD + lo_]; }
    } else {
        const double* Cb = sblk(sh, i, cc) + lk * UVS_BLK_LD + li;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = Cb[4 * q * UVS_BLK_LD];
    }
    return acc;
}
// S(i, cc)^T in the C layout = S(i, cc) in the A-operand layout of the panel product (register q: row li, column lk + 4q)
UVS_DEV d4_t chol_load_item_t(double* sh, int i, int cc, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    d4_t acc;
    if (i != UVS_NF) {
        const double* Ai = sblk(sh, i, cc) + li * UVS_BLK_LD + lk;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = Ai[4 * q];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const double yv = sh[L_DLT + 16 * cc + 4 * q + lk]; acc[q] = (li == 0) ? yv : 0.0; }
    }
    return acc;
}
UVS_DEV void chol_store_item(double* sh, int i, int cc, int lane, const d4_t& acc) {
    const int li = lane & 15, lk = lane >> 4;
    if (i == UVS_NF) { if (lk == 0) sh[L_DLT + 16 * cc + li] = acc[0]; }
    else if (i == cc) {      // diagonal block: only the lower triangle is storage of S (the upper one belongs to W)
        double* Dk = sblk(sh, cc, cc);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int r = lk + 4 * q; if (r >= li) Dk[r * UVS_BLK_LD + li] = acc[q]; }
    } else {
        double* Cb = sblk(sh, i, cc) + lk * UVS_BLK_LD + li;
#pragma unroll
        for (int q = 0; q < 4; ++q) Cb[4 * q * UVS_BLK_LD] = acc[q];
    }
}

// ---- HALF-ROW PAIRS.  The speed / bias rows (6..14) of frame i couple to frames i-1, i, i+1 only (IMU blocks; a prior keeps the speed / bias of
// frame 0 or 1), and in the frame-major elimination order no fill reaches them from an earlier frame: every neighbour of such a row lives in a
// frame >= i-1, so no path through lower-numbered unknowns connects it to a column of a frame j < i-1.  Hence block (i, j), j < i-1, of S AND of L
// is non-zero only in rows H = {0..5, 15} (pose + the spare slot), and EVERY term the workers apply -- L_ij L_cj^T with j <= c-1 <= i-2, the
// diagonal block's look-ahead terms j < c-1 included -- touches rows H only.  Two such blocks of one block column therefore share a 16-row MFMA:
// row x of the pair is row hr(x) of block (x < 8 ? i1 : i2); x = 6 and 7 both map to row 15 (the same value is computed and stored twice).
// This halves the matrix-core time of the workers (look-ahead, last terms and panel products), which were as loaded as the pivot chain.
// Host side: DevWin::chol_half_ok = 0 (a prior that keeps the speed / bias of a frame >= 2) selects the full-row path below.
UVS_DEV int chol_hr(int x) { return ((x & 7) < 6) ? (x & 7) : 15; }
// pair (i1, i2) of column cc in the A-operand layout (= transposed C layout): register q of lane (lk, li) = S(pair row li, column lk + 4q); never a diagonal block
// RHS (the first pair of a list): pair row 7 -- otherwise a second copy of row 15 of block i1 -- is the right-hand-side row (L_DLT), which so rides along for free
template <bool RHS>
UVS_DEV d4_t chol_load_pair_t(double* sh, int i1, int i2, int cc, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    const double* Ai = ((li < 8) ? sblk(sh, i1, cc) : sblk(sh, i2, cc)) + chol_hr(li) * UVS_BLK_LD + lk;
    if (RHS && li == 7) Ai = sh + L_DLT + 16 * cc + lk;
    d4_t acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = Ai[4 * q];
    return acc;
}
// pair in the C layout: register q = pair row lk + 4q (q < 2: block i1, else i2), column li; a diagonal block (i1 == cc) is read through its lower triangle
template <bool RHS>
UVS_DEV d4_t chol_load_pair(double* sh, int i1, int i2, int cc, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    d4_t acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int blk = q < 2 ? i1 : i2, r = chol_hr(lk + 4 * (q & 1));
        const double* Cb = sblk(sh, blk, cc);
        const int hi_ = r >= li ? r : li, lo_ = r >= li ? li : r;
        const double* src = (blk == cc) ? Cb + hi_ * UVS_BLK_LD + lo_ : Cb + r * UVS_BLK_LD + li;
        if (RHS && q == 1 && lk == 3) src = sh + L_DLT + 16 * cc + li;
        acc[q] = *src;
    }
    return acc;
}
template <bool RHS>
UVS_DEV void chol_store_pair(double* sh, int i1, int i2, int cc, int lane, const d4_t& acc) {
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int blk = q < 2 ? i1 : i2, r = chol_hr(lk + 4 * (q & 1));
        if (RHS && q == 1 && lk == 3) sh[L_DLT + 16 * cc + li] = acc[q];
        else if (blk != cc || r >= li) sblk(sh, blk, cc)[r * UVS_BLK_LD + li] = acc[q];
    }
}
// acc -= sum_{j in [j0, j1)} L_(pair)j L_cj^T  (TR: the transposed product), same pipelining as chol_update_item
template <bool TR, bool RHS>
UVS_DEV void chol_update_pair(double* sh, int i1, int i2, int cc, int j0, int j1, int lane, d4_t& acc) {
    const int li = lane & 15, lk = lane >> 4;
    if (j0 >= j1) return;
    const double* Bj = sblk(sh, cc, j0) + li * UVS_BLK_LD + lk;
    const double* Ai = ((li < 8) ? sblk(sh, i1, j0) : sblk(sh, i2, j0)) + chol_hr(li) * UVS_BLK_LD + lk;
    int astep = UVS_BLK_SZ;
    if (RHS && li == 7) { Ai = sh + L_DLT + 16 * j0 + lk; astep = 16; }      // y_j: 16 doubles per block column
    double a0[4], b0[4], a1[4], b1[4];
#define UVS_CH_LOAD(AV, BV) { _Pragma("unroll") for (int q = 0; q < 4; ++q) { AV[q] = Ai[4 * q]; BV[q] = Bj[4 * q]; } Bj += UVS_BLK_SZ; Ai += astep; }
#define UVS_CH_MFMA(AV, BV) { _Pragma("unroll") for (int q = 0; q < 4; ++q) acc = TR ? __builtin_amdgcn_mfma_f64_16x16x4f64(BV[q], AV[q], acc, 0, 0, 1) : __builtin_amdgcn_mfma_f64_16x16x4f64(AV[q], BV[q], acc, 0, 0, 1); }
    UVS_CH_LOAD(a0, b0)
    for (int j = j0;;) {
        if (j + 1 < j1) UVS_CH_LOAD(a1, b1)
        UVS_CH_MFMA(a0, b0)
        if (++j >= j1) break;
        if (j + 1 < j1) UVS_CH_LOAD(a0, b0)
        UVS_CH_MFMA(a1, b1)
        if (++j >= j1) break;
    }
#undef UVS_CH_LOAD
#undef UVS_CH_MFMA
}
template <bool RHS>
UVS_DEV void chol_panel_pair(double* sh, int i1, int i2, int k, int lane, const double* Bw, const d4_t& av) {      // av = chol_load_pair_t layout
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], Bw[q], acc, 0, 0, 0);
    chol_store_pair<RHS>(sh, i1, i2, k, lane, acc);
}

// (Round 2 tried two restructurings of this factorization; both are correct and both lost on MI355X, so the barrier version stays:
//  (1) the 16x16 diagonal block on the VALU, one matrix row per lane with DPP row broadcasts as FMA operands (tools/uvs_chol16.h,
//      tools/chol16_test.hip): 268 cycles per pivot without the inverse, 390 with it, against ~250 for the readlane -> rcp -> rank-1
//      MFMA chain below -- a dependent FP64 VALU operation costs 14 cycles, a DPP one 21 (tools/micro_dpp.hip), and a link of that
//      chain has about ten of them;
//  (2) the block columns as a task graph without workgroup barriers (LDS flags: wave 0 runs only the diagonal chain and the block
//      below it, the other three own block rows): 1.888 ms against 1.822 ms per 256-window launch in an A/B on one box -- three
//      workers carry as many cycles per column as the chain itself, so the chain waits for them a quarter of the time.)
// One panel item: L_ik = S_ik W_k^T (and y_k = b_k W_k^T for the right-hand-side row i == UVS_NF); Bw = the B operand W_k[c][m]
UVS_DEV void chol_panel_item(double* sh, int i, int k, int lane, const double* Bw) {
    const int li = lane & 15, lk = lane >> 4;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    double av[4];
    if (i != UVS_NF) {
        const double* Ai = sblk(sh, i, k) + li * UVS_BLK_LD + lk;
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = Ai[4 * q];
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const double yv = sh[L_DLT + 16 * k + 4 * q + lk]; av[q] = (li == 0) ? yv : 0.0; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], Bw[q], acc, 0, 0, 0);
    chol_store_item(sh, i, k, lane, acc);
}
UVS_DEV void chol_panel_from(double* sh, int i, int k, int lane, const double* Bw, const d4_t& av) {      // av = chol_load_item_t layout
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], Bw[q], acc, 0, 0, 0);
    chol_store_item(sh, i, k, lane, acc);
}
UVS_DEV void chol_panel_operand(double* sh, int k, int lane, double* Bw) {      // W_k[c][m]: strictly-upper slot (m, c) of the diagonal block, 1/L_cc on the diagonal
    const int li = lane & 15, lk = lane >> 4;
    const double* Dk = sblk(sh, k, k);
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // unconditional loads + selects (a conditional LDS read becomes a branch per element)
        const int m = 4 * q + lk;
        const double up = Dk[m * UVS_BLK_LD + li], dg = sh[L_DINV + 16 * k + li];
        Bw[q] = (m < li) ? up : (m == li ? dg : 0.0);
    }
}

// ONE workgroup barrier per block column.  Inside a column the rows are statically owned -- wave 0: the diagonal block (k, k) and the
// block (k+1, k) below it; worker w: rows k+2+w, k+2+w+nwork, ... -- so a block's last term (A), its panel solve (S3) and the store in
// between stay inside one wave, and the only cross-wave dependency left inside the column is W_k, which the workers wait for on an LDS
// flag after they have done their look-ahead (terms j < k of column k+1).  The two-barrier version made every wave wait for the
// slowest one twice per column and left the pivot chain idle during the whole panel phase.
template <bool half>
UVS_DEV void chol_factor_impl(double* sh, int debug) {
    MiniCtx c; c.sh = sh; c.o.debug = debug;
    const int tid = lane_tid(), lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform ON PURPOSE: item indices derived from it select code paths
    const int li = lane & 15, lk = lane >> 4;
    int* flg = (int*)(sh + L_XC);      // x_c is dead between the assembly and the back-substitution of the landmarks
    // [k]: W_k published; [16 + k]: S(k+1, k) carries its last term; half-row path (no workgroup barrier inside the factorization): [32 + c]: the
    // diagonal block (c, c) carries its look-ahead terms; [48 + k]: L(k+1, k) stored by wave 0; [64]: arrivals at the workers' own barrier
    if (tid < 80) flg[tid] = 0;
    if (tid == 0) sh[L_CTRL + C_CHOLOK] = 1.0;
    UVS_PROF(c, P_MISC);
#ifndef UVS_X_CHOL_ALL_WORKERS
    // 512-thread build: wave 4 shares its SIMD with the pivot chain (waves w and w + 4 sit on one SIMD) and stays out of the factorization: six workers
    constexpr bool chain_alone = ROLES;
#else
    constexpr bool chain_alone = false;
#endif
    constexpr int nwork = chain_alone ? NW - 2 : NW - 1;
    const int wrk = chain_alone ? (wv < 4 ? wv - 1 : wv - 2) : wv - 1;
    const bool idle_wave = chain_alone && wv == 4;
    long long tend_ = 0;
    d4_t Lt = {0.0, 0.0, 0.0, 0.0};      // wave 0: L(k, k-1)^T, the panel result of the previous column, = both operands of the diagonal block's last term
#define UVS_FLAG_WAIT(idx, val) while (__hip_atomic_load(flg + (idx), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (val)) __builtin_amdgcn_s_sleep(1);
#define UVS_FLAG_SET(idx) if (lane == 0) __hip_atomic_store(flg + (idx), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    for (int k = 0; k < UVS_NF; ++k) {
        double* Dk = sblk(sh, k, k);
        if (idle_wave) { if (!half && k > 0) __syncthreads(); continue; }
        // Half-row path: NO workgroup barrier per column.  The pivot chain (wave 0) only ever waits for two flags that are raised long before it
        // needs them (its diagonal block's look-ahead terms, the last term of the block below), so the critical path of the factorization is
        // chain -> W_k -> panel product of (k+1, k) -> first four MFMAs of the next diagonal block -> chain, without the workers' panel products
        // in between; the three workers meet at a barrier of their own (an LDS counter) and wait for L(k, k-1), which wave 0 stores.
        if (!half) { if (k > 0) __syncthreads(); }
        else if (wv == 0) { if (k >= 2) UVS_FLAG_WAIT(32 + k, 1) }
        else if (k > 0) {
            if (lane == 0) __hip_atomic_fetch_add(flg + 64, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            UVS_FLAG_WAIT(64, nwork * k)
            UVS_FLAG_WAIT(48 + k - 1, 1)
        }
        const long long tw0_ = debug ? clock64() : 0;
        long long tl_ = tw0_;
#define UVS_TL(slot) if (debug == 4 && lane == 0) { const long long t_ = clock64(); sh[L_WPROF + slot] += (double)(t_ - tl_); tl_ = t_; }
        if (debug == 4 && tid == 0 && k > 0) sh[L_WPROF + 7] += (double)(tw0_ - tend_);
        if (wv == 0) {
            // ---- A: last term (j = k-1) of the diagonal block (kept in MFMA registers) and of the block below it
            d4_t dacc = chol_load_item(sh, k, k, lane);
            if (k > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lt[q], Lt[q], dacc, 0, 0, 1);
            }
            // ---- S2: the diagonal block factored in registers (L -> lower triangle, W^T -> strictly upper, 1/L_jj -> L_DINV)
            d4_t T;
            UVS_TL(0)
#pragma unroll
            for (int q = 0; q < 4; ++q) T[q] = (lk + 4 * q == li) ? 1.0 : 0.0;
            double ap_prev = 0.0;
            double pivs[4] = {1.0, 1.0, 1.0, 1.0};      // pivs[q] = pivot of row lk + 4q (this lane's row of register q)
            // TWO pivots per link of the chain.  Rows j and j + 1 (j even) are k-slots s0 = j & 3 and s0 + 1 of register j >> 2, so the elimination of
            // the 2 x 2 pivot block is ONE rank-2 MFMA whose A operand holds, for every row c below the block,
            //     u1[c] = -a'[j+1][c] / d2                       a'[j+1][.] = a[j+1][.] - l10 a[j][.]   (row j + 1 after pivot j),  d2 = a'[j+1][j+1]
            //     u0[c] = -a[j][c] / a00 - u1[c] l10             l10 = a[j+1][j] / a00
            // i.e. exactly the two sequential eliminations composed (no 2 x 2 determinant: same stability as one pivot at a time), and for
            // c = j + 1 the plain multiplier -l10, so that row j + 1 leaves the MFMA as a'[j+1][.], the row the factor needs.  An FP64 MFMA blocks
            // the wave's FP64 VALU for its 64 cycles (the units are shared), so a link costs its MFMAs PLUS its scalar chain: 336 cycles per
            // pair against 2 x 230 (tools/micro_chain.hip variants 0 / 6; results equal to 7e-16).  Both rows are made visible in both of their
            // 16-lane rows by v_permlane16_swap; the inverse's elimination trails by one link as before.
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const int reg = j >> 2, s0 = j & 3, s1 = s0 + 1;
                const double a00 = bcast_lane(dacc[reg], 16 * s0 + j), a10 = bcast_lane(dacc[reg], 16 * s0 + j + 1), a11 = bcast_lane(dacc[reg], 16 * s1 + j + 1);
                if (j > 0) T = __builtin_amdgcn_mfma_f64_16x16x4f64(ap_prev, T[(j - 2) >> 2], T, 0, 0, 0);
                const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(dacc[reg]), __double2loint(dacc[reg]), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(dacc[reg]), __double2hiint(dacc[reg]), false, false);
                const double rj = __hiloint2double(hi[0], lo[0]), rj1 = __hiloint2double(hi[1], lo[1]);      // a[j][c], a[j+1][c] at column c = li (lanes of rows s0, s1)
                // 1 / a00: hardware seed y0 (2^-24) and one third-order correction 1 / x = y0 (1 + e + e^2), e = 1 - x y0, arranged so that the
                // seed-only products run beside the error term
                const double y0 = __builtin_amdgcn_rcp(a00);
                const double e0 = fma(-a00, y0, 1.0), l0 = a10 * y0, w0s = -rj * y0;
                const double p0 = fma(e0, e0, e0);
                const double l10 = fma(l0, p0, l0), w0 = fma(w0s, p0, w0s);
                const double d2 = fma(-l10, a10, a11);
                const double t = fma(-l10, rj, rj1);
                const double y1 = __builtin_amdgcn_rcp(d2);
                const double e1 = fma(-d2, y1, 1.0), u1s = -t * y1;
                const double p1 = fma(e1, e1, e1);
                double u1 = fma(u1s, p1, u1s);
                u1 = (li > j + 1) ? u1 : 0.0;
                const double u0 = fma(-u1, l10, w0);
                const double ap = (lk == s0) ? ((li > j) ? u0 : 0.0) : ((lk == s1) ? u1 : 0.0);
                if (lk == s0) pivs[reg] = a00;
                if (lk == s1) pivs[reg] = d2;
                dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, dacc[reg], dacc, 0, 0, 0);
                ap_prev = ap;
            }
            T = __builtin_amdgcn_mfma_f64_16x16x4f64(ap_prev, T[3], T, 0, 0, 0);      // row 15 of W: the multiplier -l10 of the last pair
            // the block below the diagonal (its last term was the first thing worker 0 did in this column) is fetched under the square roots
            if (k > 0) UVS_FLAG_WAIT(16 + k, 1)
            const d4_t av = chol_load_item_t(sh, k + 1, k, lane);
            // square roots + scaling, lane-parallel and off the chain: register q of this lane belongs to row j = lk + 4q
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = lk + 4 * q;
                double ljj, inv; rsqrt_pair(pivs[q], &ljj, &inv);
                Dk[li * UVS_BLK_LD + j] = (li > j) ? dacc[q] * inv : (li == j ? ljj : T[q] * inv);      // L[c][j] | W[j][m] at (m, j)
                if (li == j) { sh[L_DINV + 16 * k + j] = inv; if (!(pivs[q] > 0.0)) sh[L_CTRL + C_CHOLOK] = 0.0; }
            }
            if (lane == 0) __hip_atomic_store(flg + k, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            UVS_TL(1)
            // ---- S3 for the block below the diagonal (the right-hand-side row for the last column)
            UVS_TL(2)
            // the product TRANSPOSED, L(k+1, k)^T = W_k S(k+1, k)^T (operands swapped): its C layout is the operand layout of the next
            // column's first four MFMAs, which therefore need no LDS round trip on the critical path
            double Bw[4]; chol_panel_operand(sh, k, lane, Bw);
            {
                Lt = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 4; ++q) Lt = __builtin_amdgcn_mfma_f64_16x16x4f64(Bw[q], av[q], Lt, 0, 0, 0);
                if (k + 1 < UVS_NF) {
                    double* Lb = sblk(sh, k + 1, k) + li * UVS_BLK_LD + lk;
#pragma unroll
                    for (int q = 0; q < 4; ++q) Lb[4 * q] = Lt[q];
                } else if (li == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) sh[L_DLT + 16 * k + lk + 4 * q] = Lt[q];
                }
                UVS_FLAG_SET(48 + k)
            }
            UVS_TL(3)
            tend_ = tl_;
            if (debug == 1 && lane == 0) sh[L_WPROF + 4] += (double)(clock64() - tw0_);
        } else if (half) {
            // ---- workers, half-row pairs (see chol_hr): the items of a column are dealt round-robin over ONE list
            //   [ (k+1, k) alone | last-term items t = 0..: pairs of rows k+2.. | look-ahead items u = 0..: pairs of rows k+1.. ]
            // the right-hand-side row rides in the first pair of each of the two lists (alone only in column NF-2, which has no rows below k+1);
            // a last-term item stays in registers, transposed, until W_k is published and its panel product can run (no store / reload in between)
            const int nPA = (UVS_NF - k - 1) / 2;      // pairs of rows k+2 .. NF-1
            const int nAi = nPA > 0 ? nPA : ((k + 2 <= UVS_NF) ? 1 : 0);      // (the lone rhs item)
            const int nPL = (UVS_NF - k) / 2;          // pairs of rows k+1 .. NF-1 (look-ahead on column k+1)
            // Who does what, by deadline: the look-ahead pair that holds the NEXT diagonal block is the longest item (k terms) and the first thing the
            // pivot chain waits for, so worker 1 starts with it; worker 0 starts with the block below the diagonal (wave 0's panel product needs it
            // when the chain of this column ends); the last-term items go round-robin from worker 2 (owner(t) = (t + 2) % nwork, panel products
            // included); the remaining look-ahead pairs go to whoever is least loaded (every worker runs the same scalar bookkeeping).
            constexpr int w_la0 = 1 % nwork, w_single = 0;
            const int t0 = (wrk + nwork - (2 % nwork)) % nwork;
            d4_t hold[2];
            static_assert((UVS_NF / 2 + nwork - 1) / nwork <= 2, "last-term items of a column per worker");
            if (k > 0) {
                if (wrk == w_la0 && k + 1 < UVS_NF) {
                    const int i1 = k + 1, i2 = (i1 + 1 < UVS_NF) ? i1 + 1 : i1;
                    d4_t acc = chol_load_pair<true>(sh, i1, i2, k + 1, lane);
                    chol_update_pair<false, true>(sh, i1, i2, k + 1, 0, k, lane, acc);
                    chol_store_pair<true>(sh, i1, i2, k + 1, lane, acc);
                    UVS_FLAG_SET(32 + k + 1)
                }
                if (wrk == w_single) {
                    if (k + 1 < UVS_NF) {
                        d4_t acc = chol_load_pair<false>(sh, k + 1, k + 1, k, lane);
                        chol_update_pair<false, false>(sh, k + 1, k + 1, k, k - 1, k, lane, acc);
                        chol_store_pair<false>(sh, k + 1, k + 1, k, lane, acc);
                    } else {
                        d4_t acc = chol_load_item(sh, k + 1, k, lane);
                        chol_update_item(sh, k + 1, k, k - 1, k, lane, acc);
                        chol_store_item(sh, k + 1, k, lane, acc);
                    }
                    UVS_FLAG_SET(16 + k)
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int t = t0 + n * nwork;
                    const int i1 = k + 2 + 2 * t, i2 = (i1 + 1 < UVS_NF) ? i1 + 1 : i1;
                    if (t < nPA) {
                        if (t == 0) { hold[n] = chol_load_pair_t<true>(sh, i1, i2, k, lane); chol_update_pair<true, true>(sh, i1, i2, k, k - 1, k, lane, hold[n]); }
                        else { hold[n] = chol_load_pair_t<false>(sh, i1, i2, k, lane); chol_update_pair<true, false>(sh, i1, i2, k, k - 1, k, lane, hold[n]); }
                    } else if (t < nAi) {
                        hold[n] = chol_load_item_t(sh, UVS_NF, k, lane);
                        chol_update_item<true>(sh, UVS_NF, k, k - 1, k, lane, hold[n]);
                    }
                }
                if (k + 1 < UVS_NF) {
                    // loads in units of one MFMA (64 cycles): item overheads ~5, a term 4, a panel product ~8
                    int load[nwork];
#pragma unroll
                    for (int w = 0; w < nwork; ++w) {
                        const int tw = (w + nwork - (2 % nwork)) % nwork;
                        load[w] = (tw < nAi ? (nAi - tw + nwork - 1) / nwork : 0) * 17 + (w == w_la0 ? 5 + 4 * k : 0) + (w == w_single ? 9 : 0);
                    }
                    for (int u = 1; u < nPL; ++u) {
                        int best = 0;
#pragma unroll
                        for (int w = 1; w < nwork; ++w) if (load[w] < load[best]) best = w;
#pragma unroll
                        for (int w = 0; w < nwork; ++w) if (w == best) load[w] += 5 + 4 * k;
                        if (best == wrk) {
                            const int i1 = k + 1 + 2 * u, i2 = (i1 + 1 < UVS_NF) ? i1 + 1 : i1;
                            d4_t acc = chol_load_pair<false>(sh, i1, i2, k + 1, lane);
                            chol_update_pair<false, false>(sh, i1, i2, k + 1, 0, k, lane, acc);
                            chol_store_pair<false>(sh, i1, i2, k + 1, lane, acc);
                        }
                    }
                }
            }
            if (debug == 1 && lane == 0) sh[L_WPROF + 4 + wv] += (double)(clock64() - tw0_);
            if (wv == 1) UVS_TL(4)
            while (__hip_atomic_load(flg + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
            if (wv == 1) UVS_TL(5)
            double Bw[4]; chol_panel_operand(sh, k, lane, Bw);
            if (k > 0) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int t = t0 + n * nwork;
                    const int i1 = k + 2 + 2 * t, i2 = (i1 + 1 < UVS_NF) ? i1 + 1 : i1;
                    if (t < nPA) { if (t == 0) chol_panel_pair<true>(sh, i1, i2, k, lane, Bw, hold[n]); else chol_panel_pair<false>(sh, i1, i2, k, lane, Bw, hold[n]); }
                    else if (t < nAi) chol_panel_from(sh, UVS_NF, k, lane, Bw, hold[n]);
                }
            } else {
                for (int t = t0; t < nPA; t += nwork) {
                    const int i1 = 2 + 2 * t, i2 = (i1 + 1 < UVS_NF) ? i1 + 1 : i1;
                    if (t == 0) { const d4_t av = chol_load_pair_t<true>(sh, i1, i2, 0, lane); chol_panel_pair<true>(sh, i1, i2, 0, lane, Bw, av); }
                    else { const d4_t av = chol_load_pair_t<false>(sh, i1, i2, 0, lane); chol_panel_pair<false>(sh, i1, i2, 0, lane, Bw, av); }
                }
            }
            if (wv == 1) UVS_TL(6)
        } else {
            // (full-row path: windows whose prior keeps the speed / bias of a frame >= 2)
            // ---- A: last term of this wave's rows of column k, TRANSPOSED and kept in registers (= the A operand of the panel product: no
            // store / reload between the two);  LA: terms j < k of column k + 1 (all rows, the diagonal block included), rows dealt in the
            // opposite worker order so that the worker with the most rows of column k has the fewest look-ahead items
            d4_t hold[3];
            static_assert((UVS_NF - 2 + nwork - 1) / nwork <= 3, "rows of a column (k > 0) per worker");
            if (k > 0) {
                // the lightest worker takes the block below the diagonal (whose panel solve is wave 0's) first
                int light = 0, best = 1 << 30;
#pragma unroll
                for (int w = 0; w < nwork; ++w) {
                    const int na = UVS_NF - k - 1 - w, nl = (k + 1 < UVS_NF) ? UVS_NF - k - (nwork - 1 - w) : 0;
                    const int ca = na > 0 ? (na + nwork - 1) / nwork : 0, cl = nl > 0 ? (nl + nwork - 1) / nwork : 0;
                    const int load = ca * 17 + cl * (5 + 4 * k);      // in units of one MFMA (64 cycles): item overheads ~5, a term 4, a panel product ~8
                    if (load < best) { best = load; light = w; }
                }
                if (wrk == light) {
                    d4_t acc = chol_load_item(sh, k + 1, k, lane);
                    chol_update_item(sh, k + 1, k, k - 1, k, lane, acc);
                    chol_store_item(sh, k + 1, k, lane, acc);
                    if (lane == 0) __hip_atomic_store(flg + 16 + k, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    const int i = k + 2 + wrk + n * nwork;
                    if (i <= UVS_NF) { hold[n] = chol_load_item_t(sh, i, k, lane); chol_update_item<true>(sh, i, k, k - 1, k, lane, hold[n]); }
                }
                if (k + 1 < UVS_NF) {
                    for (int i = k + 1 + (nwork - 1 - wrk); i <= UVS_NF; i += nwork) {
                        d4_t acc = chol_load_item(sh, i, k + 1, lane);
                        chol_update_item(sh, i, k + 1, 0, k, lane, acc);
                        chol_store_item(sh, i, k + 1, lane, acc);
                    }
                }
            }
            if (debug == 1 && lane == 0) sh[L_WPROF + 4 + wv] += (double)(clock64() - tw0_);
            // ---- S3: panel of this wave's rows, once the diagonal wave has published W_k
            if (wv == 1) UVS_TL(4)
            while (__hip_atomic_load(flg + k, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(1);
            if (wv == 1) UVS_TL(5)
            double Bw[4]; chol_panel_operand(sh, k, lane, Bw);
            if (k > 0) {
#pragma unroll
                for (int n = 0; n < 3; ++n) { const int i = k + 2 + wrk + n * nwork; if (i <= UVS_NF) chol_panel_from(sh, i, k, lane, Bw, hold[n]); }
            } else {
                for (int i = 2 + wrk; i <= UVS_NF; i += nwork) chol_panel_item(sh, i, k, lane, Bw);
            }
            if (wv == 1) UVS_TL(6)
        }
        UVS_PROF(c, P_CH_DIAG);
    }
    __syncthreads();
}
// The dense solve is a REAL call (not inlined into the 500-register linearization code of k_solve / k_large_solve): its register
// allocation is then independent of the gather / factor code around it, and changes in here cannot perturb that code's allocation
// (a build at the 512-register cap once produced a wrong cost).  The LDS base is re-declared inside, so the callee still addresses
// LDS with ds_* instructions (a `double*` parameter would degrade to flat loads).
// Two instantiations (half-row pairs / full rows), each a call of its own: together in one function they need 248 VGPRs + 64 AGPRs, which reaches into the
// callee-saved registers -- 50 scratch stores and loads per lane around every factorization, 0.18 GB of traffic per 256-window launch.
__device__ __attribute__((noinline)) void chol_factor_call(int debug) {
    extern __shared__ __attribute__((aligned(16))) double sh_chol[];
    chol_factor_impl<true>(sh_chol, debug);
}
__device__ __attribute__((noinline)) void chol_factor_call_full_rows(int debug) {
    extern __shared__ __attribute__((aligned(16))) double sh_chol[];
    chol_factor_impl<false>(sh_chol, debug);
}
UVS_DEV void chol_factor(const Ctx& c) { if (c.hdr->chol_half_ok) chol_factor_call(c.o.debug); else chol_factor_call_full_rows(c.o.debug); }

// back substitution L^T x = y in place (y in L_DLT, produced by chol_factor); the diagonal solves are mat-vecs with W^T.
// One wave does all of it: the chain x_k -> (update of the rows above) -> x_k-1 is serial anyway, and inside a single wave it
// needs no workgroup barrier (22 of them otherwise).
UVS_DEV void chol_solve_impl(double* sh) {
    const int tid = lane_tid();
    double* b = sh + L_DLT;
    __syncthreads();
    if (tid < 64) {
        for (int k = UVS_NF - 1; k >= 0; --k) {
            const double* Dk = sblk(sh, k, k);
            // x_k[c] = sum_{m >= c} W[m][c] r[m] ; W[m][c] (m > c) sits at (c, m).  All loads unconditional and up front,
            // four partial sums: the only serial part left is LDS latency
            const int c16 = tid & 15;
            double wv_[16], rv[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) { wv_[m] = Dk[c16 * UVS_BLK_LD + m]; rv[m] = b[16 * k + m]; }
            const double dinv = sh[L_DINV + 16 * k + c16];
            double p4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < 16; ++m) { const double w = (m > c16) ? wv_[m] : (m == c16 ? dinv : 0.0); p4[m & 3] += w * rv[m]; }
            const double s = (p4[0] + p4[1]) + (p4[2] + p4[3]);
            wave_sync();
            if (tid < 16) b[16 * k + tid] = s;
            wave_sync();
            double xk[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) xk[r] = b[16 * k + r];
            for (int t = tid; t < 16 * k; t += 64) {
                const int j = t >> 4, cc = t & 15;
                const double* B = sblk(sh, k, j) + cc;
                double bv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bv[r] = B[r * UVS_BLK_LD];
                double u4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 16; ++r) u4[r & 3] += bv[r] * xk[r];
                b[16 * j + cc] -= (u4[0] + u4[1]) + (u4[2] + u4[3]);
            }
            wave_sync();
        }
    }
    __syncthreads();
}
UVS_DEV void chol_solve(const Ctx& c) { chol_solve_impl(c.sh); }

// ------------------------------------------------------------------ linearization: builds S (damped, Schur-reduced), G, HD, cost, gmax
// Gather work split: the lanes form UVS_NGRP GROUPS of 2 lanes (32 per wave).  A group owns one lower 6x6 pose block -- or one
// part of it: the host splits the heavy blocks (water-filling, pack_window) so that all groups carry similar work -- and lane t of
// the group owns ROWS 3t..3t+2 of the block in registers, on diagonal blocks also those rows' gradient and diag(J^T J) entries.
// Per landmark chunk the host packed, for every group, two index lists:
//   Schur list : (E row of frame a, Einv/Y row of frame b) for each landmark observed in both frames   -> - E_a^T H_ll^-1 E_b
//   direct list: the observation's Jacobian blocks contributing J^T J to the block
// Rows-per-lane is what makes this LDS-bandwidth friendly: three 8-byte reads of the lane's own operands and three 16-byte
// broadcast reads of the other row feed 18 FMAs (element-per-lane needed 2 reads per FMA and was LDS-bound; so was row-per-lane).
// A group walks its lists front to back, so every sum has a fixed order (bitwise reproducible, no atomics).
// List entries are pre-expanded by the host into LDS offsets (two 15-bit fields, doubles relative to the staging base):
//   Schur entry : offset of the E row (frame a) | offset of the E*H_ll^-1 (points) / H_ll^-1 E (lines) row (frame b) << 16
//   direct entry: points: offset of the first Jacobian block | offset of the second << 16 ; lines: record offset
// Gradient entries need no Schur list: pass B stores the Schur-CORRECTED residual rc = r - J_l H_ll^-1 g_l in every record,
// and sum_o J_p^T rc IS the reduced gradient.
static constexpr int GRP_PER_WAVE = 64 / UVS_GLANES;
static constexpr int GR = UVS_GROWS;      // block rows per lane
static constexpr int LIST_HDR = 2 * (UVS_NGRP + 1);
typedef double d2_t __attribute__((ext_vector_type(2)));

struct GAcc { double v[6 * GR], g[GR], hd[GR]; };     // v[6r + c]: rows r0 + r of the block (r0 = GR * lane-in-group); gradient and diag(J^T J) of those rows

UVS_DEV d2_t lds2(const double* p) { return *(const d2_t*)p; }
// Sum of a split block's parts into its part-0 group, through LDS scratch at `scr` (8 GR + 1 doubles per lane; must be free: the caller
// brackets the call with barriers as documented).  Part order => the same fixed summation order as one add round per part, but
// ONE barrier-separated step instead of up to 16 rounds.  All lanes must call it.
UVS_DEV void gacc_gather_parts(GAcc& A, int grp, double* scr) {
    const int tid = lane_tid() - GT0;
    const int part = grp >= 0 ? (grp >> 9) & 15 : 0, np = grp >= 0 ? ((grp >> 21) & 15) + 1 : 1;
    constexpr int NV = 8 * GR, LD = NV + 1;      // 24 values per lane (18 block entries, 3 gradient, 3 diagonal), odd stride
    double* D = scr + LD * tid;
    if (np > 1) {
#pragma unroll
        for (int q = 0; q < 6 * GR; ++q) D[q] = A.v[q];
#pragma unroll
        for (int q = 0; q < GR; ++q) { D[6 * GR + q] = A.g[q]; D[7 * GR + q] = A.hd[q]; }
    }
    __syncthreads();
    // every lane of a split block sums a SLICE of the 24 values over all parts (value q belongs to the part q mod np), in part order, into the part-0
    // lane's slot: np lanes share the np x 24 loads that the part-0 lane alone used to issue (up to 15 x 24 dependent LDS round trips: 8.8 k cycles
    // per linearization, the largest single piece of the assembly).  The value order of every sum is unchanged: part 0 + part 1 + ... .
    if (np > 1) {
        const double* D0 = D - LD * UVS_GLANES * part;      // the part-0 lane of this lane's row half
        for (int q = part; q < NV; q += np) {
            double sacc = D0[q];
            for (int p = 1; p < np; ++p) sacc += D0[LD * UVS_GLANES * p + q];      // (all 16 possible parts in flight with clamped loads + selects was measured: slower, 86 k against 63 k cycles per solve -- issue slots, not latency)
            const_cast<double*>(D0)[q] = sacc;              // only this lane reads or writes column q of the block's slots
        }
    }
    __syncthreads();
    if (np > 1 && part == 0) {
#pragma unroll
        for (int q = 0; q < 6 * GR; ++q) A.v[q] = D[q];
#pragma unroll
        for (int q = 0; q < GR; ++q) { A.g[q] = D[6 * GR + q]; A.hd[q] = D[7 * GR + q]; }
    }
}
UVS_DEV void gacc_zero(GAcc& A) {
#pragma unroll
    for (int q = 0; q < 6 * GR; ++q) A.v[q] = 0.0;
#pragma unroll
    for (int q = 0; q < GR; ++q) { A.g[q] = 0.0; A.hd[q] = 0.0; }
}
// A.v[6r + ..] += s * (row of 6 doubles held as 3 x d2_t)
UVS_DEV void row_fma(double* v, double s, const d2_t* q) {
    v[0] += s * q[0].x; v[1] += s * q[0].y; v[2] += s * q[1].x; v[3] += s * q[1].y; v[4] += s * q[2].x; v[5] += s * q[2].y;
}

// this lane's group descriptor (-1 = idle group), see uvs_layout.h: i_wblk
UVS_DEV int gather_group(const Ctx& c) { const int t = lane_tid() - GT0; return t >= 0 ? c.bi[c.hdr->i_wblk + (t / UVS_GLANES)] : -1; }

// Walks entries [e0, e1) of a gather list, K per stage, SOFTWARE PIPELINED over two register sets: while the FMAs of stage t issue, the
// LDS loads of stage t + 1 (addresses from the entries fetched during stage t - 1) and the entry words of stage t + 2 are already in
// flight.  One wave per SIMD executes in order and an LDS round trip is ~130 cycles: the un-pipelined loops paid two of them per stage
// (entry word, then operands) with nothing to overlap, 275 cycles per Schur entry against ~145 of FMA issue.
//   load(first entry index of the stage, entry words[K], Ops&)   -- issues the operand loads (entries beyond e1 are clamped by the caller)
//   use(first entry index of the stage, Ops&)                    -- the FMAs; must ignore entries >= e1 itself when K > 1
template <int K, class Ops, class Load, class Use>
UVS_DEV void gather_walk(const int* ent, int e0, int e1, Load load, Use use) {
    if (e0 >= e1) return;
    Ops A, B;
    int wa[K], wb[K];
    const auto fetch = [&](int i, int* w) {
#pragma unroll
        for (int u = 0; u < K; ++u) w[u] = ent[i + u < e1 ? i + u : e0];      // clamped: always a valid entry of this group
    };
    fetch(e0, wa); load(e0, wa, A); fetch(e0 + K, wb);
    for (int i = e0;;) {
        load(i + K, wb, B); fetch(i + 2 * K, wa);
        use(i, A);
        i += K; if (i >= e1) break;
        load(i + K, wa, A); fetch(i + 2 * K, wb);
        use(i, B);
        i += K; if (i >= e1) break;
    }
}

// EXT = the window has pseudo-frame blocks (ESTIMATE_TD / ESTIMATE_EXTRINSIC); the default instantiation folds all their special cases away
// DELTA (re-damping after a rejected step, see redamp_chunk): only the Schur walk runs -- the staged "E H_ll^-1" rows hold E times the CHANGE of
// H_ll^-1, so the walk adds the change of the Schur complement -- and the diagonal blocks also take the change of the reduced gradient,
// - E_a (H_ll^-1 g_l)_new + E_a (H_ll^-1 g_l)_old, from the rows staged `goff` doubles from the E rows (a diagonal block's Schur entries are
// exactly one (slot, slot) pair per landmark seen in its frame).
// exrows: block row 12 is the camera extrinsic (its Jacobian sits in the record's own J_ex field); false in a window with relocalization blocks,
// where pseudo frame 12 is relo_Pose, an ordinary second frame
template <bool EXT, bool DELTA = false>
UVS_DEV void gather_points(int grp, const int* lists, const double* S0, GAcc& A, int goff = 0, bool exrows = true) {
    const int g = (lane_tid() - GT0) / UVS_GLANES, r0 = GR * (lane_tid() % UVS_GLANES);
    const bool on = grp >= 0;
    const bool diag = on && ((grp >> 8) & 1);
    const bool tdg = EXT && on && ((grp >> 13) & 15) == UVS_NF;       // block row of the time offset: J1 = (J_td[0], J_td[1]) adjacent, residual 16 doubles below
    const bool exg = EXT && exrows && on && ((grp >> 13) & 15) == UVS_NF + 1;   // block rows of the camera extrinsic: J1 = the 2 x 6 J_ex block of the record
    const bool tdcol = exg && ((grp >> 17) & 15) == UVS_NF;           // (ex, td): J2 is the adjacent J_td pair and only column 0 is real
    const int p1off = tdg ? 1 : 6, rcoff = tdg ? UVS_PT_RC2 - UVS_PT_TD : exg ? UVS_PT_RC2 - UVS_PT_EX : 12;      // the corrected residual is read where pass B left it (the record's rc slot)
    const bool dirv = !(tdg && diag) && !tdcol;                 // (td, td): the direct term is the scalar J_td . J_td = the hd accumulator
    const int* ent = lists + LIST_HDR;
    // ---- Schur: acc[r][c] -= E_a[r0 + r] * Einv_b[c], two entries per stage
    {
        struct Ops { double ea[2][GR]; d2_t q[2][3]; double gg[2][DELTA ? GR : 1]; };
        const int e0 = on ? lists[g] : 0, e1 = on ? lists[g + 1] : 0;
        gather_walk<2, Ops>(ent, e0, e1,
            [&](int, const int* w, Ops& o) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const double* pa = S0 + (w[u] & 0x7fff) + r0;
                    const double* pb = S0 + ((unsigned)w[u] >> 16);
#pragma unroll
                    for (int r = 0; r < GR; ++r) o.ea[u][r] = pa[r];
                    o.q[u][0] = lds2(pb); o.q[u][1] = lds2(pb + 2); o.q[u][2] = lds2(pb + 4);
                    if (DELTA) {
#pragma unroll
                        for (int r = 0; r < GR; ++r) o.gg[u][r] = pa[goff + r];
                    }
                }
            },
            [&](int i, Ops& o) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool ok = i + u < e1;
#pragma unroll
                    for (int r = 0; r < GR; ++r) row_fma(A.v + 6 * r, ok ? -o.ea[u][r] : 0.0, o.q[u]);
                    if (DELTA) {
#pragma unroll
                        for (int r = 0; r < GR; ++r) A.g[r] -= (ok && diag) ? o.gg[u][r] : 0.0;
                    }
                }
            });
    }
    if (DELTA) return;
    // ---- direct: acc[r][c] += J1[0][r0+r] J2[0][c] + J1[1][r0+r] J2[1][c] ; diagonal blocks (J1 == J2) also g and diag(J^T J)
    {
        struct Ops { double p0[GR], p1[GR]; d2_t q0[3], q1[3], rc; };
        const int e0 = on ? lists[UVS_NGRP + 1 + g] : 0, e1 = on ? lists[UVS_NGRP + 2 + g] : 0;
        gather_walk<1, Ops>(ent, e0, e1,
            [&](int, const int* w, Ops& o) {
                const int lo = w[0] & 0x7fff;
                const double* pa = S0 + lo + r0;
                const double* pb = S0 + ((unsigned)w[0] >> 16);
#pragma unroll
                for (int r = 0; r < GR; ++r) { o.p0[r] = pa[r]; o.p1[r] = pa[p1off + r]; }
#pragma unroll
                for (int k = 0; k < 3; ++k) { o.q0[k] = lds2(pb + 2 * k); o.q1[k] = lds2(pb + 6 + 2 * k); }
                o.rc = lds2(S0 + lo + ((w[0] & UVS_PT_ENTRY_A) ? UVS_PT_RC2 - UVS_PT_A : rcoff));
            },
            [&](int, Ops& o) {
#pragma unroll
                for (int r = 0; r < GR; ++r) {
                    if (dirv) { row_fma(A.v + 6 * r, o.p0[r], o.q0); row_fma(A.v + 6 * r, o.p1[r], o.q1); }
                    if (tdcol) A.v[6 * r] += o.p0[r] * o.q0[0].x + o.p1[r] * o.q0[0].y;
                    if (diag) { A.g[r] += o.p0[r] * o.rc.x + o.p1[r] * o.rc.y; A.hd[r] += o.p0[r] * o.p0[r] + o.p1[r] * o.p1[r]; }
                }
            });
    }
}

// DELTA: as for the points; the gradient rows are one 6-vector per observation at S0 + goff + (E offset - eoff) / 4 (E rows are 24 doubles per observation)
template <bool DELTA = false>
UVS_DEV void gather_lines(int grp, const int* lists, const double* S0, GAcc& A, int goff = 0, int eoff = 0) {
    const int g = (lane_tid() - GT0) / UVS_GLANES, r0 = GR * (lane_tid() % UVS_GLANES);
    const bool on = grp >= 0;
    const bool diag = on && ((grp >> 8) & 1);
    const int* ent = lists + LIST_HDR;
    // ---- Schur: acc[r][c] -= sum_q E_a[q][r0 + r] * Y_b[q][c]
    {
        struct Ops { double ea[4][GR]; d2_t y[4][3]; double gg[DELTA ? GR : 1]; };
        const int e0 = on ? lists[g] : 0, e1 = on ? lists[g + 1] : 0;
        gather_walk<1, Ops>(ent, e0, e1,
            [&](int, const int* w, Ops& o) {
                const int lo = w[0] & 0x7fff;
                const double* pa = S0 + lo + r0;
                const double* pb = S0 + ((unsigned)w[0] >> 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int r = 0; r < GR; ++r) o.ea[q][r] = pa[6 * q + r];
                    o.y[q][0] = lds2(pb + 6 * q); o.y[q][1] = lds2(pb + 6 * q + 2); o.y[q][2] = lds2(pb + 6 * q + 4);
                }
                if (DELTA) {
                    const double* pg = S0 + goff + 6 * ((lo - eoff) / UVS_LN_EY) + r0;
#pragma unroll
                    for (int r = 0; r < GR; ++r) o.gg[r] = pg[r];
                }
            },
            [&](int, Ops& o) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < GR; ++r) row_fma(A.v + 6 * r, -o.ea[q][r], o.y[q]);
                if (DELTA) {
#pragma unroll
                    for (int r = 0; r < GR; ++r) A.g[r] -= diag ? o.gg[r] : 0.0;
                }
            });
    }
    if (DELTA) return;
    // ---- direct (always a diagonal block): 3 pose-Jacobian rows (line, line, vanishing point) + corrected residuals
    {
        struct Ops { double p[3][GR]; d2_t q[3][3], rc01; double rc2; };
        const int e0 = on ? lists[UVS_NGRP + 1 + g] : 0, e1 = on ? lists[UVS_NGRP + 2 + g] : 0;
        gather_walk<1, Ops>(ent, e0, e1,
            [&](int, const int* w, Ops& o) {
                const double* R = S0 + w[0];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
#pragma unroll
                    for (int r = 0; r < GR; ++r) o.p[k][r] = R[UVS_LN_JP + 6 * k + r0 + r];
                    o.q[k][0] = lds2(R + UVS_LN_JP + 6 * k); o.q[k][1] = lds2(R + UVS_LN_JP + 6 * k + 2); o.q[k][2] = lds2(R + UVS_LN_JP + 6 * k + 4);
                }
                o.rc01 = lds2(R); o.rc2 = R[UVS_LN_RV];
            },
            [&](int, Ops& o) {
#pragma unroll
                for (int r = 0; r < GR; ++r) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { row_fma(A.v + 6 * r, o.p[k][r], o.q[k]); A.hd[r] += o.p[k][r] * o.p[k][r]; }
                    A.g[r] += o.p[0][r] * o.rc01.x + o.p[1][r] * o.rc01.y + o.p[2][r] * o.rc2;
                }
            });
    }
}

// ---- linearization, part 1: frame-only terms at x (rotations, prior residual, IMU normal equations). Returns this lane's cost share.
// IMU blocks go through the FP64 matrix cores: per block the raw 15 x 30 Jacobian and the raw residual are laid out as one
// frame-padded 16 x 32 operand  Jaug = [ J_i | 0 | J_j | r ]  (column 16 f + dof, residual in column 31), whitened with
// T = W Jaug (W = chol(cov^-1)^T, imu_factor.h:64-66; 8 MFMAs) and squared, N = T^T T (lower 16 x 16 tiles (0,0) (1,0) (1,1);
// 12 MFMAs).  N holds J^T J in its frame blocks, J^T r in row 31 and r^T r in (31,31); the three accumulator tiles stay in
// registers until lin_assemble adds them to the pose blocks (i,i) (j,i) (j,j) of S.
static constexpr int IMU_JLD = 48;                       // row stride of Jaug / T in LDS (16 rows; 48 = 16 mod 32: no bank conflicts between k-groups)
static constexpr int IMU_WOFF = 16 * IMU_JLD;            // W as [16][17] after the operand tile
static constexpr int IMU_BLK = IMU_WOFF + UVS_BLK_SZ;    // 1040 doubles of LDS staging per block
#ifndef UVS_IMU_SLOT_SWAP
#define UVS_IMU_SLOT_SWAP 1
#endif
static constexpr int IMU_SLOTS = (UVS_NF - 1 + NW - 1) / NW;   // IMU blocks per wave (block b -> wave b % NW, slot b / NW): the MFMA stages and the tile adds use ALL waves (the staging before them only the evaluators)
// which IMU block a wave holds in slot s.  Eight waves, ten blocks: the two second-slot blocks go to waves whose FIRST block has the other parity (block 8 to wave 1, block 9 to wave 0),
// so that in each of asm_imu's two rounds (even blocks, odd blocks) every wave adds at most ONE block's tiles (with b = wv + 8 s waves 0 and 1 added two each, 3 k of the 6.5 k cycles of
// that step).  Four waves: blocks wv, wv + 4, wv + 8 as before.  NF - 1 or more = no block.
UVS_DEV int imu_block_of(int wv, int s) {
    if (NW == 8 && UVS_IMU_SLOT_SWAP) return s == 0 ? wv : (wv == 0 ? 9 : wv == 1 ? 8 : UVS_NF);
    return wv + s * NW;
}
struct ImuN { d4_t n00[IMU_SLOTS], n10[IMU_SLOTS], n11[IMU_SLOTS]; int fi[IMU_SLOTS]; bool act[IMU_SLOTS]; };      // fi / act: first frame of the wave's blocks, block present (asm_imu)

// rotations of the evaluation point + prior residual (L_PR); returns this lane's share of the prior cost
// mode 0: everything.  The persistent kernel knows more: after an ACCEPTED step (mode 1) x is the candidate the cost pass has just
// evaluated, so the rotations staged in L_RF / L_EX are already x's and the prior's y = H0 dx sits in L_PRC (the 75 x 75
// mat-vec from HBM is not repeated); after a REJECTED or invalid step (mode 2) x and L_PR are unchanged, the rotations and dx are restaged.
UVS_DEV double lin_prep(const Ctx& c, const double* x, int mode = 0) {
    UVS_PROF(c, P_MISC);
    const int tid = lane_tid(), n = c.hdr->prior_n;
    if (mode == 0) {
        stage_rotations(c, x);
        prior_dx(c, x);
        __syncthreads();
        return prior_quad(c);
    }
    if (mode == 2) { stage_rotations(c, x); prior_dx(c, x); __syncthreads(); }      // L_PDX held the REJECTED candidate's dx; y = H0 dx of x is still in L_PR
    else if (tid < n) c.sh[L_PR + tid] = c.sh[L_PRC + tid];                          // mode 1: L_PDX and L_PRC are the accepted candidate's
    __syncthreads();
    return prior_cost_share(c, L_PR);
}
// IMU normal-equation tiles; staged in the S region, so it runs when no landmark chunk is staged there (after the last gather,
// right before the assembly: the 9 accumulator tiles per wave then live only across lin_assemble)
// lin_imu_stage: W + raw residuals / Jacobians -> the operand tiles in LDS (evaluator waves; THREE workgroup barriers, the last one at its end);
// lin_imu_tiles: the MFMA stages of this wave's blocks (every wave), returns this lane's cost share
UVS_DEV void lin_imu_stage(const Ctx& c, const double* x) {
    const DevWin& h = *c.hdr;
    double* sh = c.sh;
    const int tid = lane_tid(), lane = tid & 63, wv = tid >> 6;
    __syncthreads();      // previous users of the S region are done
    UVS_PROF(c, P_GATHER);
    double* IM = sh + L_S;
    // the whitening matrices W (global, written by setup_window) are requested first and stored last: their latency runs beside the
    // blocks' own loads and the raw evaluation instead of after them
    constexpr int WPL = ((UVS_NF - 1) * 225 + ET - 1) / ET;
    double wreg[WPL];
#pragma unroll
    for (int q = 0; q < WPL; ++q) {
        const int t = tid + q * ET, tc = t < h.n_imu * 225 ? t : 0, b = tc / 225, e = tc - 225 * b;
        wreg[q] = c.ws[h.w_imu_w + (size_t)b * UVS_IMU_WS + e];
    }
    for (int t = tid; t < h.n_imu * IMU_BLK; t += ET) IM[t] = 0.0;      // operand tiles are mostly structural zeros
    UVS_TLOG(c, 40);
    __syncthreads();
    UVS_TLOG(c, 41);
    // raw residual + Jacobian of block `lane`, its four Jacobian groups on four different waves (one lane doing all of it was a 10 k-cycle
    // serial chain with 246 lanes idle; divergent parts inside one wave would serialise just the same)
    if (lane < h.n_imu && wv < 4 && !c.bi[h.i_imu + 2 * lane + 1]) {
        const int fi = c.bi[h.i_imu + 2 * lane];
        const double* blk = c.bd + h.d_imu + (size_t)lane * UVS_IMU_STRIDE;
        double* T = IM + IMU_BLK * lane;
        const double* pi_ = x + 7 * fi; const double* si_ = x + 77 + 9 * fi; const double* pj_ = x + 7 * (fi + 1); const double* sj_ = x + 77 + 9 * (fi + 1);
        double r[15];
        if (EW < 4) { imu_raw<IMU_JLD, 1, false>(blk, blk + UVS_IMU_JAC, c.o.G, pi_, si_, pj_, sj_, r, T); for (int i = 0; i < 15; ++i) T[i * IMU_JLD + 31] = r[i]; }
        else if (wv == 0) { imu_raw<IMU_JLD, 1, false, 1>(blk, blk + UVS_IMU_JAC, c.o.G, pi_, si_, pj_, sj_, r, T); for (int i = 0; i < 15; ++i) T[i * IMU_JLD + 31] = r[i]; }
        else if (wv == 1) imu_raw<IMU_JLD, 1, false, 2>(blk, blk + UVS_IMU_JAC, c.o.G, pi_, si_, pj_, sj_, r, T);
        else if (wv == 2) imu_raw<IMU_JLD, 1, false, 4>(blk, blk + UVS_IMU_JAC, c.o.G, pi_, si_, pj_, sj_, r, T);
        else imu_raw<IMU_JLD, 1, false, 8>(blk, blk + UVS_IMU_JAC, c.o.G, pi_, si_, pj_, sj_, r, T);
    }
#pragma unroll
    for (int q = 0; q < WPL; ++q) {
        const int t = tid + q * ET;
        if (t < h.n_imu * 225) { const int b = t / 225, e = t - 225 * b, i = e / 15, k = e - 15 * i; IM[IMU_BLK * b + IMU_WOFF + i * UVS_BLK_LD + k] = wreg[q]; }
    }
    UVS_TLOG(c, 42);
    __syncthreads();
    UVS_TLOG(c, 43);
}
UVS_DEV double lin_imu_tiles(const Ctx& c, ImuN& N) {
    const DevWin& h = *c.hdr;
    double* sh = c.sh;
    const int tid = lane_tid(), lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    double cost = 0.0;
    double* IM = sh + L_S;
    // stage 1 for all of this wave's blocks, ONE wave-level hand-over, then stage 2: the LDS round trips and MFMA drains of the slots overlap
    bool act[IMU_SLOTS];
#pragma unroll
    for (int s = 0; s < IMU_SLOTS; ++s) {
        const int b = imu_block_of(wv, s);
        act[s] = b < h.n_imu && !c.bi[h.i_imu + 2 * (b < h.n_imu ? b : 0) + 1];
        N.act[s] = act[s]; N.fi[s] = c.bi[h.i_imu + 2 * (b < h.n_imu ? b : 0)];
        if (act[s]) {
            double* Jb = IM + IMU_BLK * b; const double* Wb = Jb + IMU_WOFF;
            d4_t t0 = {0.0, 0.0, 0.0, 0.0}, t1 = t0;
            double wa[4], j0[4], j1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { wa[q] = Wb[li * UVS_BLK_LD + 4 * q + lk]; j0[q] = Jb[(4 * q + lk) * IMU_JLD + li]; j1[q] = Jb[(4 * q + lk) * IMU_JLD + 16 + li]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[q], j0[q], t0, 0, 0, 0); t1 = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[q], j1[q], t1, 0, 0, 0); }
            // T overwrites Jaug in place (all of this wave's reads of it are consumed above), C layout: row = lk + 4q, col = li
#pragma unroll
            for (int q = 0; q < 4; ++q) { Jb[(lk + 4 * q) * IMU_JLD + li] = t0[q]; Jb[(lk + 4 * q) * IMU_JLD + 16 + li] = t1[q]; }
        }
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < IMU_SLOTS; ++s) {
        const int b = imu_block_of(wv, s);
        d4_t n00 = {0.0, 0.0, 0.0, 0.0}, n10 = n00, n11 = n00;
        if (act[s]) {
            const double* Jb = IM + IMU_BLK * b;
            double j0[4], j1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { j0[q] = Jb[(4 * q + lk) * IMU_JLD + li]; j1[q] = Jb[(4 * q + lk) * IMU_JLD + 16 + li]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                n00 = __builtin_amdgcn_mfma_f64_16x16x4f64(j0[q], j0[q], n00, 0, 0, 0);
                n10 = __builtin_amdgcn_mfma_f64_16x16x4f64(j1[q], j0[q], n10, 0, 0, 0);
                n11 = __builtin_amdgcn_mfma_f64_16x16x4f64(j1[q], j1[q], n11, 0, 0, 0);
            }
            if (lane == 63) cost += 0.5 * n11[3];       // r^T W^T W r sits at N[31][31] = tile (1,1), row 15, col 15
        }
        N.n00[s] = n00; N.n10[s] = n10; N.n11[s] = n11;
    }
    UVS_PROF(c, P_AS_IMU);
    return cost;
}
UVS_DEV double lin_imu(const Ctx& c, const double* x, ImuN& N) { lin_imu_stage(c, x); return lin_imu_tiles(c, N); }
UVS_DEV double lin_frames(const Ctx& c, const double* x, ImuN& N) { const double pc = lin_prep(c, x); return pc + lin_imu(c, x, N); }

// Inverse of a damped 4 x 4 line block (lower packed H) through its Cholesky factor, written to X[4][4] (may be LDS), and hg = H^-1 g.
// One reciprocal square root per pivot (an FP64 division is ~30 instructions; the factor + explicit inverse had 26 of them on ONE lane per
// line while the other lanes of the workgroup wait): sqrt and reciprocal sqrt together from the hardware seed + FMA-only refinement.
UVS_DEV void spd4_inverse(const double* H, const double* gl, double* X, double* hg) {
    double L[10];
    double i0, i1, i2, i3;
    rsqrt_pair(H[0], &L[0], &i0);
    L[1] = H[1] * i0; rsqrt_pair(H[2] - L[1] * L[1], &L[2], &i1);
    L[3] = H[3] * i0; L[4] = (H[4] - L[3] * L[1]) * i1; rsqrt_pair(H[5] - L[3] * L[3] - L[4] * L[4], &L[5], &i2);
    L[6] = H[6] * i0; L[7] = (H[7] - L[6] * L[1]) * i1; L[8] = (H[8] - L[6] * L[3] - L[7] * L[4]) * i2;
    rsqrt_pair(H[9] - L[6] * L[6] - L[7] * L[7] - L[8] * L[8], &L[9], &i3);
    hg[0] = 0.0; hg[1] = 0.0; hg[2] = 0.0; hg[3] = 0.0;
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx) {
        double e[4] = {0, 0, 0, 0}; e[cidx] = 1.0;
        e[0] = e[0] * i0;
        e[1] = (e[1] - L[1] * e[0]) * i1;
        e[2] = (e[2] - L[3] * e[0] - L[4] * e[1]) * i2;
        e[3] = (e[3] - L[6] * e[0] - L[7] * e[1] - L[8] * e[2]) * i3;
        e[3] = e[3] * i3;
        e[2] = (e[2] - L[8] * e[3]) * i2;
        e[1] = (e[1] - L[4] * e[2] - L[7] * e[3]) * i1;
        e[0] = (e[0] - L[1] * e[1] - L[3] * e[2] - L[6] * e[3]) * i0;
        X[0 * 4 + cidx] = e[0]; X[1 * 4 + cidx] = e[1]; X[2 * 4 + cidx] = e[2]; X[3 * 4 + cidx] = e[3];
#pragma unroll
        for (int a = 0; a < 4; ++a) hg[a] += e[a] * gl[cidx];
    }
}

// ---- linearization, part 2: one landmark chunk: stage -> per-landmark Schur prep -> list-driven gather into acc[]
// The evaluation half (observation passes, per-landmark Schur preparation, the chunk's lists -> staging area; chunk_eval_barriers() workgroup barriers -- three for a point chunk, four for a line chunk --, the first before
// anything is written) and the gather half (no barrier).  In the 256-thread build every thread runs both, one after the other (lin_chunk); in the
// 512-thread build the evaluator waves run the first and the gatherer waves the second behind four barriers of their own (linearize).
// 512-thread build: the gatherer waves copy a chunk's lists into the staging area themselves, between the entry barrier and the three that follow (they have
// nothing else to do while the evaluators run passes A / B, and the evaluators' own loads no longer queue behind the list words)
#ifndef UVS_X_NO_LISTS_BY_GATHERERS
static constexpr bool LISTS_BY_GATHERERS = ROLES;
#else
static constexpr bool LISTS_BY_GATHERERS = false;
#endif
struct ChunkDesc { int type, k0, k1, o0, nob, nlm, nlist; const int* glists; };
UVS_DEV ChunkDesc chunk_desc(const Ctx& c, int ch) {
    const DevWin& h = *c.hdr;
    const int* chunks = c.bi + h.i_chunks;
    ChunkDesc d;
    d.type = chunks[UVS_CHUNK_INTS * ch]; d.k0 = chunks[UVS_CHUNK_INTS * ch + 1]; d.k1 = chunks[UVS_CHUNK_INTS * ch + 2];
    d.glists = c.bi + h.i_lists + chunks[UVS_CHUNK_INTS * ch + 3];      // gather lists of this chunk (HBM)
    d.nlist = chunks[UVS_CHUNK_INTS * ch + 4];
    d.nlm = d.k1 - d.k0;
    d.o0 = chunks[UVS_CHUNK_INTS * ch + 6]; d.nob = chunks[UVS_CHUNK_INTS * ch + 7];
    return d;
}
// where the gather lists of a staged chunk sit (after the records and the Schur factors)
UVS_DEV int* chunk_lists(const Ctx& c, const ChunkDesc& d) {
    const DevWin& h = *c.hdr;
    double* rec = c.sh + L_S;
    if (d.type == 0) return (int*)(rec + (size_t)d.nob * h.pt_rec + (size_t)(d.nob + h.pt_xslots * d.nlm) * 12);
    return (int*)(rec + (size_t)d.nob * (UVS_LN_REC + 2 * UVS_LN_EY) + 20 * d.nlm);
}
// What the landmark as a whole contributes to a staged point chunk, beside its observations' own slots: the scalars of the back substitution, the anchor-frame
// slot E_0 = sum_o c_o A_o (and the time-offset / extrinsic slots).  One lane per landmark: the lane of its first observation in the 256-thread build, a lane of
// the otherwise idle gatherer waves in the 512-thread build (pt_anchor_pass).
UVS_DEV void pt_landmark_slots(const Ctx& c, const ChunkDesc& d, int k, int b0, int b1, double hd, double gl, double dd, double hinv, double ginv, double* gmax_lm) {
    const DevWin& h = *c.hdr;
    const int PREC = h.pt_rec, XS = h.pt_xslots, li = k - d.k0;
    double* rec = c.sh + L_S;
    double* Eb = rec + (size_t)d.nob * PREC;
    double* EIb = Eb + (size_t)(d.nob + XS * d.nlm) * 6;
    double* E = Eb + (size_t)(b0 + XS * li) * 6; double* EI = EIb + (size_t)(b0 + XS * li) * 6;
    double* Eg = c.ws + h.w_pt_E + 6 * (size_t)(d.o0 + b0 + XS * k);
        double* px = c.ws + h.w_pt_x + 4 * (size_t)k; px[0] = ginv; px[1] = gl; px[2] = dd; px[3] = hd;
        *gmax_lm = fmax(*gmax_lm, fabs(gl));
        double e0[6] = {0, 0, 0, 0, 0, 0}, etd = 0.0;
        for (int o = b0; o < b1; ++o) {
            const double* Ro = rec + (size_t)o * PREC;
#pragma unroll
            for (int a = 0; a < 6; ++a) e0[a] += Ro[UVS_PT_C] * Ro[UVS_PT_A + a] + Ro[UVS_PT_C + 1] * Ro[UVS_PT_A + 6 + a];
            if (h.td_on) etd += Ro[UVS_PT_C] * Ro[UVS_PT_TD] + Ro[UVS_PT_C + 1] * Ro[UVS_PT_TD + 1];
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) { E[a] = e0[a]; EI[a] = e0[a] * hinv; Eg[a] = e0[a] * hinv; }
        if (h.td_on) {      // slot after the observations: the time-offset "row" J_l^T J_td (a 6-vector whose first entry is the only real one)
            const int st_ = 6 * (b1 - b0 + 1);
#pragma unroll
            for (int a = 0; a < 6; ++a) { const double e = a == 0 ? etd : 0.0; E[st_ + a] = e; EI[st_ + a] = e * hinv; Eg[st_ + a] = e * hinv; }
        }
        if (h.ex_on) {      // last slot: the extrinsic row J_l^T J_ex
            double ex6[6] = {0, 0, 0, 0, 0, 0};
            for (int o = b0; o < b1; ++o) {
                const double* Ro = rec + (size_t)o * PREC;
#pragma unroll
                for (int a = 0; a < 6; ++a) ex6[a] += Ro[UVS_PT_C] * Ro[UVS_PT_EX + a] + Ro[UVS_PT_C + 1] * Ro[UVS_PT_EX + 6 + a];
            }
            const int st_ = 6 * (b1 - b0 + 1 + (h.td_on ? 1 : 0));
#pragma unroll
            for (int a = 0; a < 6; ++a) { E[st_ + a] = ex6[a]; EI[st_ + a] = ex6[a] * hinv; Eg[st_ + a] = ex6[a] * hinv; }
        }
}
// hd = sum c^2, gl = sum c . r over the observations [b0, b1) of a landmark (every lane that needs them runs this same loop: identical values)
UVS_DEV void pt_landmark_hd_gl(const double* rec, int PREC, int b0, int b1, double* hd_, double* gl_) {
    double hd = 0.0, gl = 0.0;
    for (int o = b0; o < b1; ++o) { const double* R = rec + (size_t)o * PREC; hd += R[UVS_PT_C] * R[UVS_PT_C] + R[UVS_PT_C + 1] * R[UVS_PT_C + 1]; gl += R[UVS_PT_C] * R[0] + R[UVS_PT_C + 1] * R[1]; }
    *hd_ = hd; *gl_ = gl;
}
#ifndef UVS_X_NO_ANCHOR_BY_GATHERERS
static constexpr bool ANCHOR_BY_GATHERERS = ROLES;
#else
static constexpr bool ANCHOR_BY_GATHERERS = false;
#endif
// 512-thread build, gatherer waves, between the second and the third barrier of a point chunk (beside pass B): one lane per landmark
// what the anchor lane of the FIRST round needs from global memory (its landmark's CSR range and Jacobi scale): requested by the gatherer lane while pass A is
// still running (pt_anchor_pre, before the barrier that ends pass A), so the pass itself starts on LDS data
struct AnchorPre { int b0, b1; double sc; };
UVS_DEV void pt_anchor_pre(const Ctx& c, const ChunkDesc& d, bool first, AnchorPre& ap) {
    const DevWin& h = *c.hdr;
    const int li = lane_tid() - GT0;
    ap.b0 = 0; ap.b1 = 0; ap.sc = 1.0;
    if (li >= 0 && li < d.nlm) {
        const int k = d.k0 + li;
        ap.b0 = c.bi[h.i_pt_beg + k] - d.o0; ap.b1 = c.bi[h.i_pt_beg + k + 1] - d.o0;
        if (!first) ap.sc = c.ws[h.w_scale_pt + k];
    }
}
UVS_DEV void pt_anchor_pass(const Ctx& c, const ChunkDesc& d, bool first, double radius, const AnchorPre& ap) {
    const DevWin& h = *c.hdr;
    const int* beg = c.bi + h.i_pt_beg;
    const double* rec = c.sh + L_S;
    double gmax_lm = 0.0;
    for (int li = lane_tid() - GT0; li < d.nlm; li += UVS_GT) {
        const int k = d.k0 + li;
        const bool pre = li == lane_tid() - GT0;      // first round: from pt_anchor_pre
        const int b0 = pre ? ap.b0 : beg[k] - d.o0, b1 = pre ? ap.b1 : beg[k + 1] - d.o0;
        const double sc_old = pre ? ap.sc : (first ? 1.0 : c.ws[h.w_scale_pt + k]);
        if (b1 <= b0) continue;
        if (h.td_on | h.ex_on) {
            double hd, gl; pt_landmark_hd_gl(rec, h.pt_rec, b0, b1, &hd, &gl);
            double sc;
            if (first) { sc = c.o.jacobi ? 1.0 / (1.0 + sqrt(hd)) : 1.0; c.ws[h.w_scale_pt + k] = sc; } else sc = sc_old;
            const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
            const double hinv = 1.0 / (hd + dd), ginv = gl * hinv;
            pt_landmark_slots(c, d, k, b0, b1, hd, gl, dd, hinv, ginv, &gmax_lm);
            continue;
        }
        // the usual window (no pseudo-frame slots): h_ll, g_l and the anchor slot in ONE walk over the landmark's records (the same sums in the same order as
        // pt_landmark_hd_gl / pt_landmark_slots; half the dependent LDS round trips)
        const int PREC = h.pt_rec;
        double hd = 0.0, gl = 0.0, e0[6] = {0, 0, 0, 0, 0, 0};
        for (int o = b0; o < b1; o += 2) {      // two records per trip: their 32 LDS reads are one round trip (the sums still take the records in order)
            const double* R = rec + (size_t)o * PREC;
            const bool two = o + 1 < b1;
            const double* R2 = two ? R + PREC : R;
            double v[16], w[16];
            v[0] = R[0]; v[1] = R[1]; w[0] = R2[0]; w[1] = R2[1];
#pragma unroll
            for (int a = 0; a < 14; ++a) { v[2 + a] = R[UVS_PT_A + a]; w[2 + a] = R2[UVS_PT_A + a]; }      // A[12] then c[2] (UVS_PT_C = UVS_PT_A + 12)
            {
                const double c0 = v[14], c1 = v[15];
                hd += c0 * c0 + c1 * c1; gl += c0 * v[0] + c1 * v[1];
#pragma unroll
                for (int a = 0; a < 6; ++a) e0[a] += c0 * v[2 + a] + c1 * v[8 + a];
            }
            if (two) {
                const double c0 = w[14], c1 = w[15];
                hd += c0 * c0 + c1 * c1; gl += c0 * w[0] + c1 * w[1];
#pragma unroll
                for (int a = 0; a < 6; ++a) e0[a] += c0 * w[2 + a] + c1 * w[8 + a];
            }
        }
        double sc;
        if (first) { sc = c.o.jacobi ? 1.0 / (1.0 + sqrt(hd)) : 1.0; c.ws[h.w_scale_pt + k] = sc; } else sc = sc_old;
        const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
        const double hinv = 1.0 / (hd + dd), ginv = gl * hinv;
        double* px = c.ws + h.w_pt_x + 4 * (size_t)k; px[0] = ginv; px[1] = gl; px[2] = dd; px[3] = hd;
        gmax_lm = fmax(gmax_lm, fabs(gl));
        double* Eg = c.ws + h.w_pt_E + 6 * (size_t)(d.o0 + b0 + k);
        double* Eb = c.sh + L_S + (size_t)d.nob * PREC;
        double* E = Eb + (size_t)(b0 + li) * 6; double* EI = E + (size_t)(d.nob + d.nlm) * 6;
#pragma unroll
        for (int a = 0; a < 6; ++a) { E[a] = e0[a]; EI[a] = e0[a] * hinv; Eg[a] = e0[a] * hinv; }
    }
    lacc_add(c.sh, 0.0, gmax_lm);
}
// 512-thread build: while the gatherer waves walk chunk t, an evaluator lane READS what passes A / B of chunk t + 1 will want from global memory for its first
// observation (index words, measurements, the landmark's parameters, CSR range and scale) and throws it away: the lines then sit in the compute unit's vector
// L1, which nothing else uses during a gather walk (LDS traffic only), and the dependent loads at the head of the next passes hit there instead of in L2.
UVS_DEV void chunk_touch(const Ctx& c, const ChunkDesc& d, const double* invd, const double* line) {
    const DevWin& h = *c.hdr;
    const int tid = lane_tid();
    if (tid >= d.nob) return;
    const int o = d.o0 + tid;
    double acc = 0.0; int iacc = 0;
    if (d.type == 0) {
        const int lm = c.bi[h.i_pt_lm + o]; iacc = c.bi[h.i_pt_fi + o] + c.bi[h.i_pt_fj + o];
        const double* m = c.bd + h.d_ptmeas + o; const int st = h.pt_stride;
#pragma unroll
        for (int q = 0; q < 6; ++q) acc += m[q * st];
        acc += invd[lm] + c.ws[h.w_scale_pt + lm];
        iacc += c.bi[h.i_pt_beg + lm] + c.bi[h.i_pt_beg + lm + 1];
    } else {
        const int lm = c.bi[h.i_ln_lm + o]; iacc = c.bi[h.i_ln_fj + o] + c.bi[h.i_ln_vp + o];
#ifndef UVS_X_NO_TOUCH_B1
        iacc += c.bi[h.i_ln_beg + lm] + c.bi[h.i_ln_beg + lm + 1];      // what pass B1 asks for per line: its CSR range, its four Jacobi scales (one 32-byte run)
        acc += c.ws[h.w_scale_ln + 4 * lm] + c.ws[h.w_scale_ln + 4 * lm + 3];
#endif
        const double* m = c.bd + h.d_lnmeas + o; const int st = h.ln_stride;
#pragma unroll
        for (int q = 0; q < 9; ++q) acc += m[q * st];
        const double* ltrig = line_trig_of(c, line);
        if (ltrig) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += ltrig[8 * lm + q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += line[4 * lm + q];
        }
    }
    asm volatile("" :: "v"(acc), "v"(iacc));
}
// the gatherer waves' copy of a chunk's lists into the staging area: LG_UN words per lane in flight (a plain copy loop compiles to batches of four loads and a
// one-load-per-trip remainder; the lists of a full chunk of a large window are ~12 words per lane, i.e. several memory round trips at HBM latency)
static constexpr int LG_UN = 16;
UVS_DEV void copy_lists_gatherers(const Ctx& c, const ChunkDesc& d) {
    int* lists = chunk_lists(c, d);
    const int t0 = lane_tid() - GT0;
    for (int tb = t0; tb < d.nlist; tb += LG_UN * UVS_GT) {
        int w[LG_UN];
#pragma unroll
        for (int u = 0; u < LG_UN; ++u) { const int t = tb + u * UVS_GT; w[u] = d.glists[t < d.nlist ? t : tb]; }
#pragma unroll
        for (int u = 0; u < LG_UN; ++u) { const int t = tb + u * UVS_GT; if (t < d.nlist) lists[t] = w[u]; }
    }
}
UVS_DEV void chunk_eval(const Ctx& c, const ChunkDesc& d, const double* x, const double* invd, const double* line, bool first, double radius) {
    const DevWin& h = *c.hdr;
    double* sh = c.sh;
    const int tid = lane_tid();
    const double* RF = sh + L_RF; const double* ric = sh + L_EX; const double* tic = sh + L_EX + 9;
    double cost = 0.0, gmax_lm = 0.0;      // this chunk's share; flushed to the per-lane LDS accumulators before the gather
    {
        const int type = d.type, k0 = d.k0, k1 = d.k1;
        const int* glists = d.glists;
        const int nlist = d.nlist;
        const int nlm = d.nlm;
        UVS_TLOG(c, 1);
        __syncthreads();     // previous users of the S region are done
        UVS_PROF(c, P_GATHER);
        UVS_TLOG(c, 2);
        if (type == 0) {
            const int* beg = c.bi + h.i_pt_beg;
            const int o0 = d.o0, nob = d.nob, o1 = o0 + nob;
            const int PREC = h.pt_rec, XS = h.pt_xslots;
            double* rec = sh + L_S;                                  // [nob][PREC]
            double* Eb = rec + (size_t)nob * PREC;                   // [(nob + XS * nlm)][6]   slots per landmark: anchor, observations, (td)
            double* EIb = Eb + (size_t)(nob + XS * nlm) * 6;         // [(nob + XS * nlm)][6]  Einv = E / h_ll
            int* lists = (int*)(EIb + (size_t)(nob + XS * nlm) * 6); // gather lists staged in LDS (one HBM latency per chunk)
            if (!LISTS_BY_GATHERERS) for (int t = tid; t < nlist; t += ET) lists[t] = glists[t];
            // pass A: one lane per observation
            for (int o = o0 + tid; o < o1; o += ET) {
                const int lm = c.bi[h.i_pt_lm + o], fi = c.bi[h.i_pt_fi + o], fj = c.bi[h.i_pt_fj + o];
                double pi[3], pj[3], vij[4] = {0.0, 0.0, 0.0, 0.0};
                load_point_obs(c, o, x[183], pi, pj, vij);
                double r[2], A[12], B[12], cl[2], jtd[2] = {0.0, 0.0};
                point_eval<true, false>(x + 7 * fi, RF + 9 * fi, pose_of(x, fj), RF + 9 * fj, ric, tic, invd[lm], pi, pj, c.o.sqrt_info, r, A, B, cl, nullptr,
                                        vij, vij + 2, h.td_on ? jtd : nullptr);
                double sc; cost += 0.5 * cauchy(c.o.loss_pt, r[0] * r[0] + r[1] * r[1], &sc);
                double* R = rec + (size_t)(o - o0) * PREC;
                R[0] = sc * r[0]; R[1] = sc * r[1];
#pragma unroll
                for (int q = 0; q < 12; ++q) { R[UVS_PT_A + q] = sc * A[q]; R[UVS_PT_B + q] = sc * B[q]; }
                R[UVS_PT_C] = sc * cl[0]; R[UVS_PT_C + 1] = sc * cl[1];      // d r / d lambda (the Schur-corrected residual goes to its own slot in pass B)
                if (h.td_on || h.ex_on) { R[UVS_PT_TD] = sc * jtd[0]; R[UVS_PT_TD + 1] = sc * jtd[1]; R[UVS_PT_TD + 2] = 0.0; R[UVS_PT_TD + 3] = 0.0; }
                if (h.ex_on) {      // ESTIMATE_EXTRINSIC only: the 2 x 6 block d r / d ex_pose from a second evaluation, in its own scope so that the
                                    // default path keeps its register footprint (the kernel sits at the 512-register cap)
                    double r2[2], A2[12], B2[12], cl2[2], jex[12];
                    point_eval<true, true>(x + 7 * fi, RF + 9 * fi, pose_of(x, fj), RF + 9 * fj, ric, tic, invd[lm], pi, pj, c.o.sqrt_info, r2, A2, B2, cl2, jex);
#pragma unroll
                    for (int q = 0; q < 12; ++q) R[UVS_PT_EX + q] = sc * jex[q];
                }
            }
            UVS_TLOG(c, 3);
            __syncthreads();
            UVS_PROF(c, P_OBS);
            UVS_TLOG(c, 4);
            // pass B: one lane per observation (its landmark's h_ll / g_l are recomputed per lane, cheap);
            // the lane of a landmark's first observation also owns the anchor slot and the per-landmark scalars.
            // The corrected residual goes to the record's own rc slot (UVS_PT_RC2); the d r / d lambda columns other lanes of the landmark still read stay as they are.
            for (int ol = tid; ol < nob; ol += ET) {
                const int k = c.bi[h.i_pt_lm + o0 + ol], li = k - k0, b0 = beg[k] - o0, b1 = beg[k + 1] - o0;
                double hd, gl; pt_landmark_hd_gl(rec, PREC, b0, b1, &hd, &gl);
                const bool lead = ol == b0;
                double sc;
                if (first) { sc = c.o.jacobi ? 1.0 / (1.0 + sqrt(hd)) : 1.0; if (lead && !ANCHOR_BY_GATHERERS) c.ws[h.w_scale_pt + k] = sc; } else sc = c.ws[h.w_scale_pt + k];
                const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
                const double hinv = 1.0 / (hd + dd), ginv = gl * hinv;
                const int s = ol - b0 + 1;
                double* E = Eb + (size_t)(b0 + XS * li) * 6; double* EI = EIb + (size_t)(b0 + XS * li) * 6;
                double* Eg = c.ws + h.w_pt_E + 6 * (size_t)(o0 + b0 + XS * k);
                double* R = rec + (size_t)ol * PREC;
                const double c0 = R[UVS_PT_C], c1 = R[UVS_PT_C + 1], rr0 = R[0], rr1 = R[1];
                double Bv[12];      // all LDS reads of the record BEFORE the first LDS write (the compiler must assume E / EI alias it)
#pragma unroll
                for (int a = 0; a < 12; ++a) Bv[a] = R[UVS_PT_B + a];
#pragma unroll
                for (int a = 0; a < 6; ++a) { const double e = c0 * Bv[a] + c1 * Bv[6 + a]; E[6 * s + a] = e; EI[6 * s + a] = e * hinv; Eg[6 * s + a] = e * hinv; }
                R[UVS_PT_RC2] = rr0 - c0 * ginv; R[UVS_PT_RC2 + 1] = rr1 - c1 * ginv;       // rc = r - J_l h^-1 g_l (slot nobody reads in this pass)
                if (lead && !ANCHOR_BY_GATHERERS) pt_landmark_slots(c, d, k, b0, b1, hd, gl, dd, hinv, ginv, &gmax_lm);
            }
            UVS_TLOG(c, 5);
            __syncthreads();
            UVS_PROF(c, P_LMPREP);
            UVS_TLOG(c, 7);
            lacc_add(sh, cost, gmax_lm);
        } else {
            const int* beg = c.bi + h.i_ln_beg;
            const int o0 = d.o0, nob = d.nob, o1 = o0 + nob;
            double* rec = sh + L_S;                                  // [nob][33]
            double* Eb = rec + (size_t)nob * UVS_LN_REC;             // [nob][UVS_LN_EY]  E[c][a] = (J_l^T J_p)   (24 used, see uvs_layout.h)
            double* Yb = Eb + (size_t)nob * UVS_LN_EY;               // [nob][UVS_LN_EY]  Y = Hinv E
            double* Xb = Yb + (size_t)nob * UVS_LN_EY;               // [nlm][20] : Hinv[16], Hinv*g[4]
            int* lists = (int*)(Xb + 20 * nlm);
            if (!LISTS_BY_GATHERERS) for (int t = tid; t < nlist; t += ET) lists[t] = glists[t];
            // pass A
            const double* ltrig = line_trig_of(c, line);
            for (int o = o0 + tid; o < o1; o += ET) {
                const int lm = c.bi[h.i_ln_lm + o], fj = c.bi[h.i_ln_fj + o], hv = c.bi[h.i_ln_vp + o];
                const double* m = c.bd + h.d_lnmeas + o; const int st = h.ln_stride;
                const double sp[3] = {m[0], m[st], m[2 * st]}, ep[3] = {m[3 * st], m[4 * st], m[5 * st]}, vp[3] = {m[6 * st], m[7 * st], m[8 * st]};
                LineGeom g;
                line_geom<true>(x + 7 * fj, x + 7 * fj + 3, RF + 9 * fj, ric, tic, line + 4 * lm, g, ltrig ? ltrig + 8 * lm : nullptr);
                double* R = rec + (size_t)(o - o0) * UVS_LN_REC;
                double r[2], Jp[12], Jl[8], sc;
                line_residual<true>(g, sp, ep, c.o.line_factor, r, Jp, Jl);
                cost += 0.5 * cauchy(c.o.loss_ln, r[0] * r[0] + r[1] * r[1], &sc);
                R[0] = sc * r[0]; R[1] = sc * r[1];
#pragma unroll
                for (int q = 0; q < 12; ++q) R[UVS_LN_JP + q] = sc * Jp[q];
#pragma unroll
                for (int q = 0; q < 8; ++q) R[UVS_LN_JL + q] = sc * Jl[q];
                R[UVS_LN_RV + 1] = (double)((lm - k0) | (fj << 10));      // pad slot: the chunk-local line index (< 1024: pack_window) and the frame, for pass B2
                if (hv) {
                    double rv, Jvp[6], Jvl[4];
                    vp_residual<true>(g, vp, c.o.vp_factor, &rv, Jvp, Jvl);
                    cost += 0.5 * cauchy(c.o.loss_vp, rv * rv, &sc);
                    R[UVS_LN_RV] = sc * rv;
#pragma unroll
                    for (int q = 0; q < 6; ++q) R[UVS_LN_JP + 12 + q] = sc * Jvp[q];
#pragma unroll
                    for (int q = 0; q < 4; ++q) R[UVS_LN_JL + 8 + q] = sc * Jvl[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 6; ++q) R[UVS_LN_JP + 12 + q] = 0.0;
                    R[UVS_LN_RV] = 0.0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) R[UVS_LN_JL + 8 + q] = 0.0;
                }
            }
            UVS_TLOG(c, 3);
            __syncthreads();
            UVS_PROF(c, P_OBS);
            UVS_TLOG(c, 4);
            // pass B1: one lane per line: H_ll, g_l, damping, 4x4 inverse
            for (int li = tid; li < nlm; li += ET) {
                const int k = k0 + li, b0 = beg[k] - o0, b1 = beg[k + 1] - o0;
                double H[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, gl[4] = {0, 0, 0, 0};   // lower packed (0,0)(1,0)(1,1)(2,0)...
                double scl[4] = {1.0, 1.0, 1.0, 1.0};      // Jacobi scales: requested before the accumulation loop, used after it
                if (!first) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) scl[a] = c.ws[h.w_scale_ln + 4 * k + a];
                }
                for (int o = b0; o < b1; ++o) {
                    const double* R = rec + (size_t)o * UVS_LN_REC;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        gl[a] += R[UVS_LN_JL + a] * R[0] + R[UVS_LN_JL + 4 + a] * R[1] + R[UVS_LN_JL + 8 + a] * R[UVS_LN_RV];
#pragma unroll
                        for (int b = 0; b <= a; ++b) H[(a * (a + 1)) / 2 + b] += R[UVS_LN_JL + a] * R[UVS_LN_JL + b] + R[UVS_LN_JL + 4 + a] * R[UVS_LN_JL + 4 + b] + R[UVS_LN_JL + 8 + a] * R[UVS_LN_JL + 8 + b];
                    }
                }
                double* lx = c.ws + h.w_ln_x + UVS_LN_X * (size_t)k;
#pragma unroll
                for (int q = 0; q < 10; ++q) lx[12 + q] = H[q];      // undamped: what a re-damping after a rejected step starts from
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const double hd = H[(a * (a + 1)) / 2 + a];
                    double sc;
                    if (first) { sc = c.o.jacobi ? 1.0 / (1.0 + sqrt(hd)) : 1.0; c.ws[h.w_scale_ln + 4 * k + a] = sc; } else sc = scl[a];
                    const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
                    H[(a * (a + 1)) / 2 + a] = hd + dd;
                    lx[4 + a] = gl[a]; lx[8 + a] = dd;
                    gmax_lm = fmax(gmax_lm, fabs(gl[a]));
                }
                // Cholesky of the damped 4x4 and explicit inverse
                double* X = Xb + 20 * li;
                double hg[4];
                spd4_inverse(H, gl, X, hg);
#pragma unroll
                for (int a = 0; a < 4; ++a) { X[16 + a] = hg[a]; lx[a] = hg[a]; }
            }
            UVS_TLOG(c, 5);
            __syncthreads();
            UVS_TLOG(c, 6);
            // pass B2: one lane per line observation: E and Y = Hinv E
            for (int o = tid; o < nob; o += ET) {
                double* R = rec + (size_t)o * UVS_LN_REC;
                const int li = (int)R[UVS_LN_RV + 1] & 1023;
                double* E = Eb + (size_t)o * UVS_LN_EY; double* Y = Yb + (size_t)o * UVS_LN_EY;
                double* Yg = c.ws + h.w_ln_Y + 24 * (size_t)(o0 + o);
                // all LDS reads BEFORE the first LDS write (the compiler must assume the E / Y stores alias the record)
                double Xv[20], Jl[12], Jp[18];
#pragma unroll
                for (int q = 0; q < 20; ++q) Xv[q] = Xb[20 * li + q];
#pragma unroll
                for (int q = 0; q < 12; ++q) Jl[q] = R[UVS_LN_JL + q];
#pragma unroll
                for (int q = 0; q < 18; ++q) Jp[q] = R[UVS_LN_JP + q];
                const double rr0 = R[0], rr1 = R[1], rr2 = R[UVS_LN_RV];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    double e[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { e[q] = Jl[q] * Jp[a] + Jl[4 + q] * Jp[6 + a] + Jl[8 + q] * Jp[12 + a]; E[6 * q + a] = e[q]; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const double y = Xv[4 * q] * e[0] + Xv[4 * q + 1] * e[1] + Xv[4 * q + 2] * e[2] + Xv[4 * q + 3] * e[3]; Y[6 * q + a] = y; Yg[6 * q + a] = y; }
                }
                // Schur-corrected residual rc = r - J_l (H_ll^-1 g_l) for the 2 line rows and the VP row
                R[0] = rr0 - (Jl[0] * Xv[16] + Jl[1] * Xv[17] + Jl[2] * Xv[18] + Jl[3] * Xv[19]);
                R[1] = rr1 - (Jl[4] * Xv[16] + Jl[5] * Xv[17] + Jl[6] * Xv[18] + Jl[7] * Xv[19]);
                R[UVS_LN_RV] = rr2 - (Jl[8] * Xv[16] + Jl[9] * Xv[17] + Jl[10] * Xv[18] + Jl[11] * Xv[19]);
            }
            __syncthreads();
            UVS_PROF(c, P_LMPREP);
            UVS_TLOG(c, 7);
            lacc_add(sh, cost, gmax_lm);
        }
    }
}
UVS_DEV int chunk_eval_barriers(const ChunkDesc& d) { return d.type == 0 ? 3 : 4; }      // workgroup barriers inside chunk_eval
UVS_DEV void chunk_gather(const Ctx& c, const ChunkDesc& d, int grp, GAcc& acc) {
    const DevWin& h = *c.hdr;
    const double* rec = c.sh + L_S;
    const int* lists = chunk_lists(c, d);
    UVS_TLOG(c, 8);
    if (d.type == 0) {
        const long long tg0_ = clock64();
        if (h.td_on | h.ex_on) gather_points<true>(grp, lists, rec, acc, 0, h.ex_on != 0); else gather_points<false>(grp, lists, rec, acc);
        if (c.o.debug == 2 && (lane_tid() & 63) == 0) c.sh[L_WPROF + 4 + ((lane_tid() >> 6) & 3)] += (double)(clock64() - tg0_);
    } else gather_lines(grp, lists, rec, acc);
    UVS_TLOG(c, 10);
}
UVS_DEV void lin_chunk(const Ctx& c, int ch, const double* x, const double* invd, const double* line, bool first, double radius,
                       int grp, GAcc& acc) {
    const ChunkDesc d = chunk_desc(c, ch);
    chunk_eval(c, d, x, invd, line, first, radius);
    chunk_gather(c, d, grp, acc);
}

// ---- re-damping: what a REJECTED step needs instead of a new linearization.  x has not moved, so every Jacobian, the cost and the direct
// J^T J sums are what they were; only the trust-region radius -- the damping of the landmark blocks and of the frame diagonal -- is new
// (Ceres does not re-evaluate the Jacobian after an unsuccessful step either).  The Schur complement is linear in H_ll^-1:
//     S_new = S_old - sum_l E_a^T ((H_ll + D_new)^-1 - (H_ll + D_old)^-1) E_b ,   g_new = g_old - sum_l E_a^T ((H_ll + D)^-1 g_l)_{new - old}
// and E = J_l^T J_p comes back from the back-substitution store of the last linearization (E (H_ll + D_old)^-1 per slot for points,
// (H_ll + D_old)^-1 E per observation for lines, whose undamped 4 x 4 H_ll sits in the workspace).  Per chunk: recover E, stage E and E times
// the CHANGE of the inverse where lin_chunk stages E and E H^-1, run the Schur half of the gather (DELTA), rewrite the store and the
// per-landmark scalars for the back-substitution.  No observation is evaluated, no direct term is gathered: ~7 k cycles per chunk against ~25 k.
// redamp_prep: the staging half (2 workgroup barriers for a point chunk, 3 for a line chunk: redamp_barriers); redamp_gather: the Schur walk.
UVS_DEV int redamp_barriers(const ChunkDesc& d) { return d.type == 0 ? 2 : 3; }
UVS_DEV void redamp_prep(const Ctx& c, const ChunkDesc& d, double radius) {
    const DevWin& h = *c.hdr;
    double* sh = c.sh;
    const int tid = lane_tid();
    const int type = d.type, k0 = d.k0;
    const int* glists = d.glists;
    const int nlist = d.nlist;
    const int nlm = d.nlm;
    __syncthreads();     // previous users of the S region are done
    double* rec = sh + L_S;
    if (type == 0) {
        const int* beg = c.bi + h.i_pt_beg;
        const int o0 = d.o0, nob = d.nob;
        const int PREC = h.pt_rec, XS = h.pt_xslots;      // (XS == 1: no pseudo-frame slots, see DevWin::redamp_ok)
        double* Gb = rec;                                        // [(nob + XS nlm)][6]  E times the change of H_ll^-1 g_l
        double* Eb = rec + (size_t)nob * PREC;                   // same places as in lin_chunk: the lists address them
        double* EIb = Eb + (size_t)(nob + XS * nlm) * 6;
        int* lists = (int*)(EIb + (size_t)(nob + XS * nlm) * 6);
        for (int t = tid; t < nlist; t += ET) lists[t] = glists[t];
        // one lane per slot: first the anchor slots (one per landmark with observations), then
        // one lane per observation slot.  Slot of observation o of landmark li: (o - o0) + XS li + 1; every lane recomputes its landmark's scalars.
        for (int t = tid; t < nlm + nob; t += ET) {
            const bool anchor = t < nlm;
            const int ol = anchor ? 0 : t - nlm;
            const int k = anchor ? k0 + t : c.bi[h.i_pt_lm + o0 + ol], li = k - k0;
            const int nobs = beg[k + 1] - beg[k];
            if (nobs == 0) continue;      // a landmark without observations has no entries in the lists and no scalars in the workspace
            double* px = c.ws + h.w_pt_x + 4 * (size_t)k;
            const double gl = px[1], dd_old = px[2], hd = px[3], sc = c.ws[h.w_scale_pt + k];
            const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
            const double hdo = hd + dd_old, hinv = 1.0 / (hd + dd), dh = hinv - 1.0 / hdo;
            const int slot = anchor ? (beg[k] - o0) + XS * li : ol + XS * li + 1;
            double* Eg = c.ws + h.w_pt_E + 6 * (size_t)(anchor ? beg[k] + XS * k : o0 + ol + XS * k + 1);
            double e[6];
#pragma unroll
            for (int a = 0; a < 6; ++a) e[a] = Eg[a] * hdo;
#pragma unroll
            for (int a = 0; a < 6; ++a) { Eb[6 * slot + a] = e[a]; EIb[6 * slot + a] = e[a] * dh; Gb[6 * slot + a] = e[a] * dh * gl; Eg[a] = e[a] * hinv; }
        }
        __syncthreads();
        // the landmark scalars are rewritten only now: the lanes above read the OLD damping of their landmark
        for (int li = tid; li < nlm; li += ET) {
            const int k = k0 + li;
            if (beg[k + 1] == beg[k]) continue;
            double* px = c.ws + h.w_pt_x + 4 * (size_t)k;
            const double gl = px[1], hd = px[3], sc = c.ws[h.w_scale_pt + k];
            const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
            px[0] = gl * (1.0 / (hd + dd)); px[2] = dd;
        }
    } else {
        const int o0 = d.o0, nob = d.nob;
        double* T = rec;                                         // [nlm][34]: (H + D_new)^-1 [16] | change of (H + D)^-1 g [4] | H [10] | D_old [4]
        double* Gb = rec + (size_t)nlm * 34;                     // [nob][6]
        double* Eb = rec + (size_t)nob * UVS_LN_REC;             // same places as in lin_chunk
        double* Yb = Eb + (size_t)nob * UVS_LN_EY;
        double* Xb = Yb + (size_t)nob * UVS_LN_EY;
        int* lists = (int*)(Xb + 20 * nlm);
        for (int t = tid; t < nlist; t += ET) lists[t] = glists[t];
        for (int li = tid; li < nlm; li += ET) {
            const int k = k0 + li;
            double* lx = c.ws + h.w_ln_x + UVS_LN_X * (size_t)k;
            double H[10], gl[4], ddo[4], hgo[4], hg[4];
#pragma unroll
            for (int q = 0; q < 10; ++q) H[q] = lx[12 + q];
#pragma unroll
            for (int a = 0; a < 4; ++a) { hgo[a] = lx[a]; gl[a] = lx[4 + a]; ddo[a] = lx[8 + a]; }
            double* t = T + 34 * li;
#pragma unroll
            for (int q = 0; q < 10; ++q) t[20 + q] = H[q];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const double hd = H[(a * (a + 1)) / 2 + a], sc = c.ws[h.w_scale_ln + 4 * k + a];
                const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
                H[(a * (a + 1)) / 2 + a] = hd + dd;
                t[30 + a] = ddo[a]; lx[8 + a] = dd;
            }
            spd4_inverse(H, gl, t, hg);
#pragma unroll
            for (int a = 0; a < 4; ++a) { t[16 + a] = hg[a] - hgo[a]; lx[a] = hg[a]; }
        }
        __syncthreads();
        for (int o = tid; o < nob; o += ET) {
            const int li = c.bi[h.i_ln_lm + o0 + o] - k0;
            double tv[34], yo[24];
#pragma unroll
            for (int q = 0; q < 34; ++q) tv[q] = T[34 * li + q];
            double* Yg = c.ws + h.w_ln_Y + 24 * (size_t)(o0 + o);
#pragma unroll
            for (int q = 0; q < 24; ++q) yo[q] = Yg[q];
            double* E = Eb + (size_t)o * UVS_LN_EY; double* Y = Yb + (size_t)o * UVS_LN_EY; double* G = Gb + (size_t)o * 6;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {      // E = (H + D_old) Y_old
                    double v = tv[30 + q] * yo[6 * q + a];
#pragma unroll
                    for (int p2 = 0; p2 < 4; ++p2) { const int hi_ = q > p2 ? q : p2, lo_ = q > p2 ? p2 : q; v += tv[20 + (hi_ * (hi_ + 1)) / 2 + lo_] * yo[6 * p2 + a]; }
                    e[q] = v; E[6 * q + a] = v;
                }
                double ga = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double yn = tv[4 * q] * e[0] + tv[4 * q + 1] * e[1] + tv[4 * q + 2] * e[2] + tv[4 * q + 3] * e[3];
                    Y[6 * q + a] = yn - yo[6 * q + a]; Yg[6 * q + a] = yn;
                    ga += e[q] * tv[16 + q];
                }
                G[a] = ga;
            }
        }
        __syncthreads();
    }
}
UVS_DEV void redamp_gather(const Ctx& c, const ChunkDesc& d, int grp, GAcc& acc) {
    const double* rec = c.sh + L_S;
    const int* lists = chunk_lists(c, d);
    if (d.type == 0) gather_points<false, true>(grp, lists, rec, acc, -d.nob * c.hdr->pt_rec);
    else gather_lines<true>(grp, lists, rec, acc, d.nlm * 34, d.nob * UVS_LN_REC);
}
UVS_DEV void redamp_chunk(const Ctx& c, int ch, double radius, int grp, GAcc& acc) {
    const ChunkDesc d = chunk_desc(c, ch);
    redamp_prep(c, d, radius);
    redamp_gather(c, d, grp, acc);
}
// ---- relo_Pose as a SECOND-LEVEL block (DevWin::relo2: relocalization blocks in a window with a free extrinsic; uvs_layout.h UVS_RELO2_BLOCKROW).
// The gather blocks of block row 13 land in a side buffer of the workspace: R = S(relo, frame dofs) [6][176], Rrr = S(relo, relo), its gradient
// and diag(J^T J).  With M = Rrr + D_r (D_r: the same Jacobi-scaled LM damping every other dof gets) the reduced system loses the block before it is
// factored -- S -= R^T M^-1 R, g -= R^T M^-1 g_r -- and gets it back after the solve, d_r = -M^-1 (g_r + R d_f): the same exact solve of the damped
// system as with the block inside S (what happens when the extrinsic is fixed and relo_Pose fits the spare slots), at the price of a rank-6 update
// through global memory.  R couples relo_Pose to pose / extrinsic / time-offset dofs only (73 indices), never to speed / bias rows, so the
// half-row structure of the Cholesky survives.
static constexpr int R2_R = 0, R2_RR = 1056, R2_G = 1092, R2_HD = 1098, R2_SC = 1104, R2_DD = 1110, R2_MI = 1116, R2_MG = 1152, R2_Z = 1158, R2_DR = 2214;
static_assert(R2_DR + 6 <= UVS_RELO2_DOUBLES, "side buffer");
UVS_DEV int relo2_index(int p) { return p < 66 ? 16 * (p / 6) + p % 6 : (p < 72 ? UVS_EX_INDEX(p - 66) : UVS_TD_INDEX); }      // the 73 S indices R can touch
UVS_DEV void relo2_eliminate(const Ctx& c, const double* x, bool first, double radius, double& gmax) {
    const DevWin& h = *c.hdr;
    double* sh = c.sh; double* W = c.ws + h.w_relo2;
    const int tid = lane_tid();
    __syncthreads();      // the block rows written by the assembly are visible
    if (tid == 0) {
        double M[36], L[36], Li[36];
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const double hd = W[R2_HD + a];
            if (first) W[R2_SC + a] = c.o.jacobi ? 1.0 / (1.0 + sqrt(hd)) : 1.0;
            const double sc = W[R2_SC + a];
            const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
            W[R2_DD + a] = dd;
#pragma unroll
            for (int b = 0; b < 6; ++b) M[6 * a + b] = W[R2_RR + (a >= b ? 6 * a + b : 6 * b + a)] + (a == b ? dd : 0.0);
        }
        // M = L L^T, Li = L^-1, M^-1 = Li^T Li
#pragma unroll
        for (int q = 0; q < 36; ++q) { L[q] = 0.0; Li[q] = 0.0; }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double dj = M[6 * j + j];
#pragma unroll
            for (int k = 0; k < 6; ++k) if (k < j) dj -= L[6 * j + k] * L[6 * j + k];
            if (!(dj > 0.0)) ok = false;
            double ljj, inv; rsqrt_pair(dj, &ljj, &inv);
            L[6 * j + j] = ljj;
#pragma unroll
            for (int i = 0; i < 6; ++i) if (i > j) { double v = M[6 * i + j]; for (int k = 0; k < j; ++k) v -= L[6 * i + k] * L[6 * j + k]; L[6 * i + j] = v * inv; }
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {      // column j of L^-1 by forward substitution
#pragma unroll
            for (int i = 0; i < 6; ++i) if (i >= j) { double v = (i == j) ? 1.0 : 0.0; for (int k = j; k < i; ++k) v -= L[6 * i + k] * Li[6 * k + j]; Li[6 * i + j] = v / L[6 * i + i]; }
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double mg = 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) { double v = 0.0; for (int k = 0; k < 6; ++k) v += Li[6 * k + a] * Li[6 * k + b]; W[R2_MI + 6 * a + b] = v; mg += v * W[R2_G + b]; }
            W[R2_MG + a] = mg;
        }
        if (!ok) sh[L_CTRL + C_CHOLOK] = 0.0;      // (reset by chol_factor; a failed block shows up there as well: the update below then makes S indefinite)
    }
    if (tid == UVS_NF + 1) {      // projected-gradient measure of relo_Pose (its landmark-reduced gradient, like every frame block's)
        double d[6], xp[7];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = -W[R2_G + k];
        pose_plus(x + 184, d, xp);
#pragma unroll
        for (int k = 0; k < 7; ++k) gmax = fmax(gmax, fabs(x[184 + k] - xp[k]));
    }
    __syncthreads();
    if (tid < UVS_RD) {      // Z = M^-1 R (column tid), g -= R^T M^-1 g_r
        double r[6], gs = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) { r[a] = W[R2_R + UVS_RD * a + tid]; gs += r[a] * W[R2_MG + a]; }
#pragma unroll
        for (int a = 0; a < 6; ++a) { double z = 0.0; for (int b = 0; b < 6; ++b) z += W[R2_MI + 6 * a + b] * r[b]; W[R2_Z + UVS_RD * a + tid] = z; }
        sh[L_G + tid] -= gs;
    }
    __syncthreads();
    for (int e = tid; e < 73 * 73; e += NT) {      // S -= R^T Z on the 73 x 73 index set (lower triangle of S; the product is symmetric)
        const int p = e / 73, q = e - 73 * p;
        if (q > p) continue;
        const int i = relo2_index(p), j = relo2_index(q);
        double acc = 0.0;
#pragma unroll
        for (int a = 0; a < 6; ++a) acc += W[R2_R + UVS_RD * a + i] * W[R2_Z + UVS_RD * a + j];
        const int hi_ = i >= j ? i : j, lo_ = i >= j ? j : i;
        sh[L_S + sidx(hi_, lo_)] -= acc;
    }
    __syncthreads();
}
// the step of relo_Pose after the reduced solve (d_f in L_DLT): d_r = -(M^-1 g_r + Z d_f); mirrored to where an observation of pseudo frame 12 looks for it
UVS_DEV void relo2_backsub(const Ctx& c) {
    const DevWin& h = *c.hdr;
    double* sh = c.sh; double* W = c.ws + h.w_relo2;
    const int tid = lane_tid();
    if (tid < 6) {
        double acc = W[R2_MG + tid];
        for (int p = 0; p < 73; ++p) { const int i = relo2_index(p); acc += W[R2_Z + UVS_RD * tid + i] * sh[L_DLT + i]; }
        W[R2_DR + tid] = -acc;
        sh[L_DLT + 16 * UVS_RELO_FRAME + tid] = -acc;
    }
    __syncthreads();
}

// ---- linearization, part 3: assemble the damped reduced system in LDS from the gathered pose blocks + IMU + prior
// mode 0: everything (k_solve).  The large-window kernels split the work over two workgroups that run concurrently with the landmark
// chunks resp. after them: mode 1 = the FRAME image only (zero, IMU tiles, prior; no landmark blocks, no damping: k_large_chunks' extra
// workgroup writes S / G / HD to global memory), mode 2 = landmark blocks + damping / scaling / gradient norm ONTO an image already in LDS
// (k_large_solve; `Ain` must hold the complete sums in the part-0 groups, nothing is gathered from other parts).
// ---- pieces of the assembly (shared by lin_assemble and the role-split linearization of the 512-thread build)
// zero the image: S, G, HD (and the side buffer of a second-level relo_Pose)
UVS_DEV void asm_zero(const Ctx& c) {
    const DevWin& h = *c.hdr; double* sh = c.sh; const int tid = lane_tid();
    { const d2_t z2 = {0.0, 0.0}; for (int i = tid; i < UVS_S_DOUBLES / 2; i += NT) *(d2_t*)(sh + L_S + 2 * i) = z2; }      // ds_write_b128
    if (tid < UVS_RD) { sh[L_G + tid] = 0.0; sh[L_HD + tid] = 0.0; }
    if (h.relo2) for (int t = tid; t < R2_SC; t += NT) c.ws[h.w_relo2 + t] = 0.0;
}
// the part-0 group of every pose block adds its rows (one writer per block: a single round)
UVS_DEV void asm_part0(const Ctx& c, int grp, const GAcc& A) {
    const DevWin& h = *c.hdr; double* sh = c.sh; const int tid = lane_tid();
    if (grp >= 0 && ((grp >> 9) & 15) == 0) {
        const int r0 = GR * (tid % UVS_GLANES);
        const int fa = (grp >> 13) & 15, fb = (grp >> 17) & 15;
        if (fa == UVS_RELO2_BLOCKROW) {      // relo_Pose beside a free extrinsic: its rows go to the side buffer (relo2_eliminate)
            double* W = c.ws + h.w_relo2;
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                const int a = r0 + r;
                if (fb < UVS_NF) {
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) W[R2_R + UVS_RD * a + 16 * fb + cc] = A.v[6 * r + cc];
                } else if (fb == UVS_NF) W[R2_R + UVS_RD * a + UVS_TD_INDEX] = A.v[6 * r];
                else if (fb == UVS_NF + 1) {
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) W[R2_R + UVS_RD * a + UVS_EX_INDEX(cc)] = A.v[6 * r + cc];
                } else {
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) if (cc <= a) W[R2_RR + 6 * a + cc] = A.v[6 * r + cc];
                    W[R2_G + a] = A.g[r]; W[R2_HD + a] = A.hd[r];
                }
            }
        } else if (fa == UVS_NF + 1) {  // camera-extrinsic rows (ESTIMATE_EXTRINSIC): dof a of Ex_Pose sits at S index 16 a + 15, anywhere relative to the column
#pragma unroll
            for (int r = 0; r < GR; ++r) {
                const int a = r0 + r, i = UVS_EX_INDEX(a);
                if (fb < UVS_NF) {
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) { const int j = 16 * fb + cc; sh[L_S + (i >= j ? sidx(i, j) : sidx(j, i))] += A.v[6 * r + cc]; }
                } else if (fb == UVS_NF) sh[L_S + sidx(UVS_TD_INDEX, i)] += A.v[6 * r];
                else {
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) if (cc <= a) sh[L_S + sidx(i, UVS_EX_INDEX(cc))] += A.v[6 * r + cc];
                    sh[L_G + i] += A.g[r]; sh[L_HD + i] += A.hd[r];
                }
            }
        } else if (fa == UVS_NF) {      // time-offset row (ESTIMATE_TD): row UVS_TD_INDEX of S, only row 0 of lane 0 of the group is real
            if (r0 == 0) {
                if (fb < UVS_NF) {
                    double* row = sh + L_S + sidx(UVS_TD_INDEX, 16 * fb);
                    double cur[6];
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) cur[cc] = row[cc];
#pragma unroll
                    for (int cc = 0; cc < 6; ++cc) row[cc] = cur[cc] + A.v[cc];
                } else {
                    sh[L_S + sidx(UVS_TD_INDEX, UVS_TD_INDEX)] += A.v[0] + A.hd[0];      // Schur part + J_td . J_td
                    sh[L_G + UVS_TD_INDEX] += A.g[0]; sh[L_HD + UVS_TD_INDEX] += A.hd[0];
                }
            }
        } else {
        const bool dg = fa == fb;
        double* row0 = sh + L_S + sidx(16 * fa + r0, 16 * fb);
        double cur[6 * GR], cg[GR], chd[GR];      // all reads before the first write (every "+=" to LDS otherwise waits for the one before)
#pragma unroll
        for (int r = 0; r < GR; ++r) {
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) cur[6 * r + cc] = row0[r * UVS_BLK_LD + cc];
            cg[r] = sh[L_G + 16 * fa + r0 + r]; chd[r] = sh[L_HD + 16 * fa + r0 + r];
        }
#pragma unroll
        for (int r = 0; r < GR; ++r) {
#pragma unroll
            for (int cc = 0; cc < 6; ++cc) if (!dg || cc <= r0 + r) row0[r * UVS_BLK_LD + cc] = cur[6 * r + cc] + A.v[6 * r + cc];
            if (dg) { sh[L_G + 16 * fa + r0 + r] = cg[r] + A.g[r]; sh[L_HD + 16 * fa + r0 + r] = chd[r] + A.hd[r]; }
        }
        }
    }
}
// IMU normal-equation tiles from the registers of lin_imu (even blocks, then odd: consecutive blocks share a diagonal frame block); TWO workgroup barriers
UVS_DEV void asm_imu(const Ctx& c, const ImuN& N) {
    const DevWin& h = *c.hdr; double* sh = c.sh; const int tid = lane_tid();
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, lk = lane >> 4;
    int fis[IMU_SLOTS]; bool act[IMU_SLOTS];
#pragma unroll
    for (int s = 0; s < IMU_SLOTS; ++s) {      // (the block table came with the tiles: lin_imu)
        act[s] = N.act[s] && li < 15;
        fis[s] = N.fi[s];
    }
    for (int par = 0; par < 2; ++par) {
#pragma unroll
        for (int s = 0; s < IMU_SLOTS; ++s) {
            if (act[s] && (fis[s] & 1) == par) {      // (blocks of one parity share no frame: fi' - fi >= 2)
                const int fi = fis[s], fj = fi + 1;
                // C layout: row = lk + 4q, col = li.  Rows < 15 go to S (lower triangles of the diagonal tiles), row 15 is J^T r.
                double* b10 = sblk(sh, fj, fi) + lk * UVS_BLK_LD + li;
                double* b00 = sblk(sh, fi, fi) + lk * UVS_BLK_LD + li;
                double* b11 = sblk(sh, fj, fj) + lk * UVS_BLK_LD + li;
                double c10[4], c00[4], c11[4];      // reads first, then writes
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = lk + 4 * q;
                    const bool isg = row == 15;
                    c10[q] = isg ? sh[L_G + 16 * fi + li] : b10[4 * q * UVS_BLK_LD];
                    c11[q] = isg ? sh[L_G + 16 * fj + li] : b11[4 * q * UVS_BLK_LD];
                    c00[q] = b00[4 * q * UVS_BLK_LD];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = lk + 4 * q;
                    if (row < 15) {
                        b10[4 * q * UVS_BLK_LD] = c10[q] + N.n10[s][q];
                        if (li <= row) {
                            b00[4 * q * UVS_BLK_LD] = c00[q] + N.n00[s][q];
                            b11[4 * q * UVS_BLK_LD] = c11[q] + N.n11[s][q];
                            if (li == row) { sh[L_HD + 16 * fi + row] += N.n00[s][q]; sh[L_HD + 16 * fj + row] += N.n11[s][q]; }
                        }
                    } else {
                        sh[L_G + 16 * fi + li] = c10[q] + N.n10[s][q];
                        sh[L_G + 16 * fj + li] = c11[q] + N.n11[s][q];
                    }
                }
            }
        }
        __syncthreads();
    }
}
// prior: H0 = J0^T J0 and g = g0 + H0 dx (y = H0 dx came with the cost of this point: prior_quad)
UVS_DEV void asm_prior(const Ctx& c) {
    const DevWin& h = *c.hdr; double* sh = c.sh; const int tid = lane_tid();
    if (h.prior_n > 0) {
        const int n = h.prior_n;
        const int* cm = c.bi + h.i_prior + 80;
        {   // H0 entries that are structurally non-zero in S: host table of (index into the dense n x n H0, S offset)
            const double* H0 = c.ws + h.w_prior_h0;
            const int tot = h.n_cimg;
            const int* src = (const int*)(c.ws + h.w_cimg); const int* off = src + tot;
            for (int t0 = tid; t0 < tot; t0 += 16 * NT) {      // up to 16 independent load pairs in flight per trip (one trip for the 10-frame prior)
                int idx[16], sr[16]; double v[16], cur[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int t = t0 + u * NT; const bool in = t < tot; idx[u] = in ? off[t] : -1; sr[u] = in ? src[t] : 0; }
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = H0[sr[u]];
#pragma unroll
                for (int u = 0; u < 16; ++u) cur[u] = sh[L_S + (idx[u] >= 0 ? idx[u] : 0)];
#pragma unroll
                for (int u = 0; u < 16; ++u) if (idx[u] >= 0) sh[L_S + idx[u]] = cur[u] + v[u];
            }
            if (tid < UVS_RD) sh[L_HD + tid] += c.ws[h.w_prior_h0 + UVS_PH_HD(n) + tid];      // diag(J0^T J0) by S index (setup_window)
        }
        if (tid < n && cm[tid] >= 0) sh[L_G + cm[tid]] += c.ws[h.w_prior_h0 + UVS_PH_G0(n) + tid] + sh[L_PR + tid];      // one writer per S index (the IMU adds ended with a barrier)
    }
}
// The same in two halves (512-thread build): everything asm_prior reads from global memory -- two dependent round trips, table then H0 entry -- is requested
// BEFORE the image is zeroed and crosses the zero / landmark / IMU steps of the assembly in registers; the adds then only touch LDS.
static constexpr int PA_UN = 8;      // entries per lane held ahead (the 10-frame prior has ~2.6 k structural entries: 6 per lane)
struct PriorAdd { int idx[PA_UN]; double v[PA_UN]; double hd, g; int cmi; };
UVS_DEV void asm_prior_load(const Ctx& c, PriorAdd& pa) {
    const DevWin& h = *c.hdr; const int tid = lane_tid();
    pa.hd = 0.0; pa.g = 0.0; pa.cmi = -1;
#pragma unroll
    for (int u = 0; u < PA_UN; ++u) { pa.idx[u] = -1; pa.v[u] = 0.0; }
    if (h.prior_n <= 0) return;
    const int n = h.prior_n, tot = h.n_cimg;
    const double* H0 = c.ws + h.w_prior_h0;
    const int* src = (const int*)(c.ws + h.w_cimg); const int* off = src + tot;
    int sr[PA_UN];
#pragma unroll
    for (int u = 0; u < PA_UN; ++u) { const int t = tid + u * NT; const bool in = t < tot; pa.idx[u] = in ? off[t] : -1; sr[u] = in ? src[t] : 0; }
#pragma unroll
    for (int u = 0; u < PA_UN; ++u) pa.v[u] = H0[sr[u]];
    if (tid < UVS_RD) pa.hd = c.ws[h.w_prior_h0 + UVS_PH_HD(n) + tid];
    if (tid < n) { pa.cmi = c.bi[h.i_prior + 80 + tid]; pa.g = c.ws[h.w_prior_h0 + UVS_PH_G0(n) + tid]; }
}
UVS_DEV void asm_prior_add(const Ctx& c, const PriorAdd& pa) {
    const DevWin& h = *c.hdr; double* sh = c.sh; const int tid = lane_tid();
    if (h.prior_n <= 0) return;
    double cur[PA_UN];
#pragma unroll
    for (int u = 0; u < PA_UN; ++u) cur[u] = sh[L_S + (pa.idx[u] >= 0 ? pa.idx[u] : 0)];
#pragma unroll
    for (int u = 0; u < PA_UN; ++u) if (pa.idx[u] >= 0) sh[L_S + pa.idx[u]] = cur[u] + pa.v[u];
    {   // a prior with more structural entries than PA_UN per lane: the rest as in asm_prior
        const double* H0 = c.ws + h.w_prior_h0;
        const int tot = h.n_cimg;
        const int* src = (const int*)(c.ws + h.w_cimg); const int* off = src + tot;
        for (int t = tid + PA_UN * NT; t < tot; t += NT) sh[L_S + off[t]] += H0[src[t]];
    }
    if (tid < UVS_RD) sh[L_HD + tid] += pa.hd;
    if (pa.cmi >= 0) sh[L_G + pa.cmi] += pa.g + sh[L_PR + tid];
}
// frame damping, Jacobi scaling (first linearization only), dummy pivots, projected-gradient max norm, the linearization's cost -> control words
UVS_DEV void asm_finish(const Ctx& c, const double* x, bool first, double radius, double cost, double gmax_lm, int mode) {
    const DevWin& h = *c.hdr; double* sh = c.sh; const int tid = lane_tid();
    double gmax = gmax_lm;
    if (tid < UVS_RD) {
        const int k = tid & 15;
        if (k < 15 || (h.td_on && tid == UVS_TD_INDEX) || ((h.ex_on | h.relo_on) && tid < 96)) {      // spare slots in use: td at 175, Ex_Pose (or relo_Pose) dofs at 15, 31, ... 95
            const double hd = sh[L_HD + tid];
            if (first) sh[L_SC + tid] = c.o.jacobi ? 1.0 / (1.0 + sqrt(hd)) : 1.0;
            const double sc = sh[L_SC + tid];
            const double dd = fmin(fmax(sc * sc * hd, c.o.dlo), c.o.dhi) / (radius * sc * sc);
            sh[L_DD + tid] = dd;
            sh[L_S + sidx(tid, tid)] += dd;
            if (k >= 6 && !(k == 15 && tid < 96)) gmax = fmax(gmax, fabs(sh[L_G + tid]));      // Euclidean blocks (the Ex_Pose slots are a manifold block, below)
        } else { sh[L_S + sidx(tid, tid)] = 1.0; sh[L_DD + tid] = 0.0; sh[L_G + tid] = 0.0; sh[L_SC + tid] = 1.0; }
    }
    const int pt = tid - (NT - 64);      // the projected-gradient measures on the LAST wave, beside the damping on the first three (one wave would run them one after the other)
    if (pt >= 0 && pt < UVS_NF) {   // || x - Plus(x, -g) ||_inf on the pose block
        double d[6], xp[7];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = -sh[L_G + 16 * pt + k];
        pose_plus(x + 7 * pt, d, xp);
#pragma unroll
        for (int k = 0; k < 7; ++k) gmax = fmax(gmax, fabs(x[7 * pt + k] - xp[k]));
    }
    if ((h.ex_on | h.relo_on) && pt == UVS_NF) {   // same projected-gradient measure for the extrinsic pose block / relo_Pose
        const double* xa = x + ((h.relo_on && !h.relo2) ? 184 : 176);
        double d[6], xp[7];
#pragma unroll
        for (int k = 0; k < 6; ++k) d[k] = -sh[L_G + UVS_EX_INDEX(k)];
        pose_plus(xa, d, xp);
#pragma unroll
        for (int k = 0; k < 7; ++k) gmax = fmax(gmax, fabs(xa[k] - xp[k]));
    }
    if (h.relo2) relo2_eliminate(c, x, first, radius, gmax);      // (mode 0: k_solve; mode 2: k_large_solve, whose side buffer was filled from the reduced tail of block row 13)
    double s4[4] = {cost, 0.0, 0.0, 0.0};
    block_reduce(sh, s4, &gmax);
    if (tid == 0) { sh[L_CTRL + C_COST] = s4[0]; sh[L_CTRL + C_GMAX] = gmax; }
    __syncthreads();
}
UVS_DEV void lin_assemble(const Ctx& c, const double* x, bool first, double radius, int grp, const GAcc& Ain, const ImuN& N, double cost, double gmax_lm, int mode = 0) {
    const DevWin& h = *c.hdr;
    double* sh = c.sh;
    const int tid = lane_tid();
    __syncthreads();
    UVS_PROF(c, P_GATHER);
    // ---- assemble the reduced system in LDS
    // parts of split blocks -> their part-0 group (the staging area is free now; S is zeroed only after the sums are in registers)
    GAcc A = Ain;
    long long tz_ = clock64();
#define UVS_TZ(slot) if (c.o.debug == 3 && tid == 0) { const long long t_ = clock64(); sh[L_WPROF + slot] += (double)(t_ - tz_); tz_ = t_; }
    if (mode == 0) { gacc_gather_parts(A, grp, sh + L_S); __syncthreads(); }
    UVS_TZ(4)
    if (mode != 2) {
        asm_zero(c);
        __syncthreads();
    }
    UVS_TZ(5)
    // the part-0 group of every pose block adds its rows (one writer per block: a single round)
    if (mode != 1) {
        asm_part0(c, grp, A);
        __syncthreads();
    }
    UVS_TZ(6)
#undef UVS_TZ
    UVS_PROF(c, P_AS_ZERO);
    // IMU normal-equation tiles from the registers of lin_imu (even blocks, then odd: consecutive blocks share a diagonal frame block)
    if (mode != 2) {
        asm_imu(c, N);
    }
    // prior: H0 = J0^T J0 and g = g0 + H0 dx (y = H0 dx came with the cost of this point: prior_quad)
    if (mode != 2) asm_prior(c);
    __syncthreads();
    UVS_PROF(c, P_AS_ADD);
    if (mode == 1) return;
    asm_finish(c, x, first, radius, cost, gmax_lm, mode);
    UVS_PROF(c, P_ASSEMBLE);
}

// ---- the role-split linearization of the 512-thread build (ROLES).  Evaluator and gatherer waves run DIFFERENT code between the same workgroup
// barriers (s_barrier counts arriving waves, not program counters); the two branches below must therefore execute the same NUMBER of barriers:
//     per chunk      evaluators: chunk_eval (3 / 4 barriers inside)        gatherers: as many barriers (the lists and the anchor slots in between), then the gather walk
//     frame terms    evaluators: lin_imu (3 barriers: entry, zeroed, raw)  gatherers: entry barrier, then gacc_gather_parts (2 barriers) beside the IMU staging
//     assembly       barrier | zero | barrier | part-0 rows (gatherers) | barrier | IMU tiles (evaluators, 2 barriers) | prior (all) | barrier | asm_finish (all)
// The gather accumulators exist only in the gatherer branch and the IMU tiles only in the evaluator branch, so neither occupies registers where the
// other branch's temporaries live.  redamp = true: re-damping of the stored linearization (relinearize_damping) instead of a new one.
static constexpr int ROLE_PARTS_OFF = (UVS_NF - 1) * IMU_BLK;      // the part sums of split blocks meet behind the IMU staging tiles
static_assert(ROLE_PARTS_OFF + (8 * GR + 1) * UVS_GT <= UVS_S_DOUBLES, "IMU staging + part sums exceed the S region");
UVS_DEV void role_barriers(int n) { for (int i = 0; i < n; ++i) __syncthreads(); }
// the accumulators of the last linearization in the workspace, component-major (one 512-byte run per wave and component)
UVS_DEV void gacc_store(const Ctx& c, const GAcc& A) {
    double* W = c.ws + c.hdr->w_gacc + (lane_tid() - GT0);
#pragma unroll
    for (int q = 0; q < 6 * GR; ++q) __builtin_nontemporal_store(A.v[q], W + q * UVS_GT);
#pragma unroll
    for (int q = 0; q < GR; ++q) { __builtin_nontemporal_store(A.g[q], W + (6 * GR + q) * UVS_GT); __builtin_nontemporal_store(A.hd[q], W + (7 * GR + q) * UVS_GT); }
}
UVS_DEV void gacc_load(const Ctx& c, GAcc& A) {
    const double* W = c.ws + c.hdr->w_gacc + (lane_tid() - GT0);
#pragma unroll
    for (int q = 0; q < 6 * GR; ++q) A.v[q] = W[q * UVS_GT];
#pragma unroll
    for (int q = 0; q < GR; ++q) { A.g[q] = W[(6 * GR + q) * UVS_GT]; A.hd[q] = W[(7 * GR + q) * UVS_GT]; }
}
UVS_DEV void linearize_roles(const Ctx& c, const double* x, const double* invd, const double* line, bool first, double radius, int prep_mode, bool redamp) {
    const DevWin& h = *c.hdr;
    const bool ev = role_eval();
    { const double pc = lin_prep(c, x, redamp ? 2 : prep_mode); if (!redamp) lacc_set(c.sh, pc, 0.0); }
    double ic = 0.0;
    UVS_TLOG(c, 20);
    PriorAdd pa;
    ImuN N;
    if (ev) {
        ChunkDesc d = chunk_desc(c, 0);
        for (int ch = 0; ch < h.n_chunks; ++ch) {
            if (redamp) redamp_prep(c, d, radius);
            else {
                chunk_eval(c, d, x, invd, line, first, radius);
            }
            if (ch + 1 < h.n_chunks) {
                d = chunk_desc(c, ch + 1);
#ifdef UVS_CHUNK_TOUCH
                if (!redamp) chunk_touch(c, d, invd, line);
#endif
            }
        }
        UVS_TLOG(c, 21);
        lin_imu_stage(c, x);
#ifndef UVS_X_NO_PRIOR_AHEAD
        asm_prior_load(c, pa);      // (before the MFMA stages: a workgroup barrier waits for outstanding loads, so they have to be in flight beside real work)
#endif
        ic = lin_imu_tiles(c, N);
        UVS_TLOG(c, 22);
        __syncthreads();
        asm_zero(c);
        __syncthreads();
        UVS_TLOG(c, 23);
        __syncthreads();
        UVS_TLOG(c, 24);
        asm_imu(c, N);
        UVS_TLOG(c, 25);
    } else {
        const int grp = gather_group(c);
        GAcc A;
        if (redamp) gacc_load(c, A); else gacc_zero(A);
        ChunkDesc d = chunk_desc(c, 0);
        for (int ch = 0; ch < h.n_chunks; ++ch) {
            if (redamp) { role_barriers(redamp_barriers(d)); redamp_gather(c, d, grp, A); }
            else {
                __syncthreads();      // the chunk's entry barrier: the staging area is free
                if (LISTS_BY_GATHERERS) copy_lists_gatherers(c, d);
                AnchorPre ap; ap.b0 = 0; ap.b1 = 0; ap.sc = 1.0;
                if (ANCHOR_BY_GATHERERS && d.type == 0) pt_anchor_pre(c, d, first, ap);
                __syncthreads();      // pass A is done: the records are complete
                if (ANCHOR_BY_GATHERERS && d.type == 0) pt_anchor_pass(c, d, first, radius, ap);
                role_barriers(chunk_eval_barriers(d) - 2);
                chunk_gather(c, d, grp, A);
            }
            if (ch + 1 < h.n_chunks) d = chunk_desc(c, ch + 1);
        }
        __syncthreads();      // (lin_imu's entry barrier: every gather walk is done, the staging area is free)
        if (h.redamp_ok && c.o.redamp) gacc_store(c, A);      // per part: a re-damping continues from these
        GAcc& T = A;
        gacc_gather_parts(T, grp, c.sh + L_S + ROLE_PARTS_OFF);
#ifndef UVS_X_NO_PRIOR_AHEAD
        asm_prior_load(c, pa);
#endif
        ic = lin_imu_tiles(c, N);      // (the operand tiles are complete since gacc_gather_parts' second barrier = the last one of lin_imu_stage)
        __syncthreads();
        asm_zero(c);
        __syncthreads();
        asm_part0(c, grp, T);
        __syncthreads();
        asm_imu(c, N);
    }
#ifndef UVS_X_NO_PRIOR_AHEAD
    asm_prior_add(c, pa);
#else
    asm_prior(c);
#endif
    __syncthreads();
    UVS_TLOG(c, 26);
    asm_finish(c, x, first && !redamp, radius, lacc_cost(c.sh) + ic, lacc_gmax(c.sh), 0);
    UVS_TLOG(c, 27);
}

// `A`: the gather accumulators, owned by the caller (k_solve keeps them in registers between a linearization and a possible re-damping)
UVS_DEV void linearize(const Ctx& c, const double* x, const double* invd, const double* line, bool first, double radius, int prep_mode, GAcc& A) {
    if (ROLES) { linearize_roles(c, x, invd, line, first, radius, prep_mode, false); return; }
    const DevWin& h = *c.hdr;
    const int grp = gather_group(c);       // this lane's gather group: pose block | flags (uvs_layout.h: i_wblk)
    gacc_zero(A);
    { const double pc = lin_prep(c, x, prep_mode); lacc_set(c.sh, pc, 0.0); }
    {
        ChunkDesc d = chunk_desc(c, 0);
        for (int ch = 0; ch < h.n_chunks; ++ch) {
            chunk_eval(c, d, x, invd, line, first, radius);
            chunk_gather(c, d, grp, A);
            if (ch + 1 < h.n_chunks) d = chunk_desc(c, ch + 1);
        }
    }
    ImuN N;
    const double ic = lin_imu(c, x, N);
    lin_assemble(c, x, first, radius, grp, A, N, lacc_cost(c.sh) + ic, lacc_gmax(c.sh));
}

// After a rejected / invalid step: the same point, a smaller radius.  The per-lane cost / gradient-norm accumulators of the linearization
// (L_LCOST / L_LGMAX) still hold their values.
UVS_DEV void relinearize_damping(const Ctx& c, const double* x, double radius, GAcc& A) {
    if (ROLES) { linearize_roles(c, x, nullptr, nullptr, false, radius, 2, true); return; }
    const DevWin& h = *c.hdr;
    const int grp = gather_group(c);
    (void)lin_prep(c, x, 2);
    {
        ChunkDesc d = chunk_desc(c, 0);
        for (int ch = 0; ch < h.n_chunks; ++ch) {
            redamp_prep(c, d, radius);
            redamp_gather(c, d, grp, A);
            if (ch + 1 < h.n_chunks) d = chunk_desc(c, ch + 1);
        }
    }
    ImuN N;
    const double ic = lin_imu(c, x, N);
    lin_assemble(c, x, false, radius, grp, A, N, lacc_cost(c.sh) + ic, lacc_gmax(c.sh));
}

// ------------------------------------------------------------------ back-substitution + candidate + model terms
// frames: XC = X (+) DLT ; landmarks: cand = cur + delta.  Accumulates into CTRL: MCC, STEP2, XC2.
// PSB: Schur slots of a point per batch of loads; STREAM (the landmark-sharded kernel, whose lanes pay HBM latency per dependent load): the per-landmark scalars are
// requested with the CSR range instead of after the slot loop, and the lines run one lane per (line, parameter) as in the 512-thread persistent kernel
#ifndef UVS_BS_FRAME_LANE0
#define UVS_BS_FRAME_LANE0 192
#endif
template <int LNBT = BS_LNB, int PSB = 4, bool STREAM = false>
UVS_DEV void backsub_candidate(const Ctx& c, const double* invd, const double* line, double* invd_c, double* line_c,
                                  int pk0, int pk1, int lk0, int lk1, bool with_frames, double* sums_out) {
    const DevWin& h = *c.hdr;
    double* sh = c.sh;
    const int tid = lane_tid();
    const double* d = sh + L_DLT;
    double gd = 0.0, dd2 = 0.0, step2 = 0.0, xc2 = 0.0;
    const bool td_on = h.td_on != 0;
    const bool ex_on = h.ex_on != 0, relo_on = h.relo_on != 0;
    double* ltrig_c = const_cast<double*>(line_trig_of(c, line_c));
    const bool relo2 = h.relo2 != 0;
    if (relo2) relo2_backsub(c);
    else if (relo_on) {      // the step of relo_Pose where an observation of pseudo frame 12 looks for it: d[16 * 12 + a]
        if (tid < 6) sh[L_DLT + 16 * UVS_RELO_FRAME + tid] = d[UVS_EX_INDEX(tid)];
        __syncthreads();
    }
    if (with_frames && tid < UVS_RD && ((tid & 15) < 15 || (td_on && tid == UVS_TD_INDEX) || ((ex_on | relo_on) && tid < 96))) { gd += sh[L_G + tid] * d[tid]; dd2 += sh[L_DD + tid] * d[tid] * d[tid]; }
    // the frame blocks' candidates on the lanes of wave 3 (no landmark work there in a canonical window: points sit on the first waves, line parameters from wave 4 on), not in front
    // of the point loop of wave 0
    const int ft = tid - UVS_BS_FRAME_LANE0;
    if (with_frames && ft >= 0 && ft < UVS_NF) {
        double xp[7];
        pose_plus(sh + L_X + 7 * ft, d + 16 * ft, xp);
#pragma unroll
        for (int k = 0; k < 7; ++k) { sh[L_XC + 7 * ft + k] = xp[k]; const double e = xp[k] - sh[L_X + 7 * ft + k]; step2 += e * e; xc2 += xp[k] * xp[k]; }
#pragma unroll
        for (int k = 0; k < 9; ++k) { const double v = sh[L_X + 77 + 9 * ft + k] + d[16 * ft + 6 + k]; sh[L_XC + 77 + 9 * ft + k] = v; step2 += d[16 * ft + 6 + k] * d[16 * ft + 6 + k]; xc2 += v * v; }
    }
    if (with_frames && ft == UVS_NF) {   // Ex_Pose moves only with ESTIMATE_EXTRINSIC, para_Td only with ESTIMATE_TD
        if (ex_on) {
            double de[6], xp[7];
#pragma unroll
            for (int k = 0; k < 6; ++k) de[k] = d[UVS_EX_INDEX(k)];
            pose_plus(sh + L_X + 176, de, xp);
#pragma unroll
            for (int k = 0; k < 7; ++k) { sh[L_XC + 176 + k] = xp[k]; const double e = xp[k] - sh[L_X + 176 + k]; step2 += e * e; xc2 += xp[k] * xp[k]; }
        } else for (int k = 0; k < 7; ++k) sh[L_XC + 176 + k] = sh[L_X + 176 + k];
        const double dtd = td_on ? d[UVS_TD_INDEX] : 0.0, tdc = sh[L_X + 183] + dtd;
        sh[L_XC + 183] = tdc;
        if (td_on) { step2 += dtd * dtd; xc2 += tdc * tdc; }
        if (relo_on) {      // relo_Pose: a free pose block (estimator.cpp:947-948)
            double de[6], xp[7];
#pragma unroll
            for (int k = 0; k < 6; ++k) de[k] = d[16 * UVS_RELO_FRAME + k];
            if (relo2) {      // its share of g . d and d^T D d: g_r (d_r + M^-1 R d_f) = -g_r . M^-1 g_r (the landmark terms below restore their part the same way)
                const double* W = c.ws + h.w_relo2;
#pragma unroll
                for (int k = 0; k < 6; ++k) { gd -= W[R2_G + k] * W[R2_MG + k]; dd2 += W[R2_DD + k] * de[k] * de[k]; }
            }
            pose_plus(sh + L_X + 184, de, xp);
#pragma unroll
            for (int k = 0; k < 7; ++k) { sh[L_XC + 184 + k] = xp[k]; const double e = xp[k] - sh[L_X + 184 + k]; step2 += e * e; xc2 += xp[k] * xp[k]; }
        } else for (int k = 0; k < 7; ++k) sh[L_XC + 184 + k] = sh[L_X + 184 + k];
    }
    // points: delta = -ginv - sum_s Einv[s] . delta_pose(frame(s))
    const int* pbeg = c.bi + h.i_pt_beg;
    for (int k = pk0 + tid; k < pk1; k += NT) {
        const int b0 = pbeg[k], b1 = pbeg[k + 1];
        const double* px = c.ws + h.w_pt_x + 4 * (size_t)k;
        const double* Eg = c.ws + h.w_pt_E + 6 * (size_t)(b0 + h.pt_xslots * k);
        double px0 = 0.0, px1 = 0.0, px2 = 0.0, iv0 = 0.0;
        if (STREAM) { px0 = px[0]; px1 = px[1]; px2 = px[2]; iv0 = invd[k]; }
        double t = 0.0;      // Einv . delta_pose  (the landmark's share of the frame step)
        if (b1 > b0) {
            // slots (anchor, observations) four at a time: the frame indices and the 6-vectors of a batch are independent loads, so a lane
            // pays one HBM/L2 round trip per BATCH instead of one per observation
            const int ns = b1 - b0 + 1;
            for (int s0 = 0; s0 < ns; s0 += PSB) {
                int fr[PSB]; double ev[PSB][6];
#pragma unroll
                for (int u = 0; u < PSB; ++u) {
                    const int sl = s0 + u, slc = sl < ns ? sl : 0;
                    fr[u] = slc == 0 ? c.bi[h.i_pt_fi + b0] : c.bi[h.i_pt_fj + b0 + slc - 1];
#pragma unroll
                    for (int a = 0; a < 6; ++a) ev[u][a] = Eg[6 * slc + a];
                }
#pragma unroll
                for (int u = 0; u < PSB; ++u) {
                    if (s0 + u >= ns) continue;
#pragma unroll
                    for (int a = 0; a < 6; ++a) t += ev[u][a] * d[16 * fr[u] + a];
                }
            }
            if (td_on) t += Eg[6 * (b1 - b0 + 1)] * d[UVS_TD_INDEX];
            if (ex_on) {
                const double* e = Eg + 6 * (b1 - b0 + 1 + (td_on ? 1 : 0));
#pragma unroll
                for (int a = 0; a < 6; ++a) t += e[a] * d[UVS_EX_INDEX(a)];
            }
        }
        if (!STREAM) { px0 = px[0]; px1 = px[1]; px2 = px[2]; iv0 = invd[k]; }
        const double dl = -px0 - t;
        const double v = iv0 + dl;
        invd_c[k] = v;
        // L_G holds the Schur-REDUCED frame gradient g_f - E^T h^-1 g_l; (E delta_f) h^-1 g_l = t * g_l restores the full g_f . delta_f
        gd += px1 * (dl + t); dd2 += px2 * dl * dl; step2 += dl * dl; xc2 += v * v;
    }
    // lines: delta(4) = -Hinv g - sum_s Y[s] delta_pose
    constexpr int LNB = LNBT;
    const int* lbeg = c.bi + h.i_ln_beg;
#ifndef UVS_X_NO_LINE_QSPLIT
    if ((ROLES && ltrig_c) || STREAM) {
        // 512-thread build: one lane per (line, parameter).  The four sums t[q] of a line are independent and each parameter's sin / cos (an FP64 sincos is a few
        // hundred instructions; four of them back to back were the longest chain of the phase) goes with its own lane.  t[q] takes its terms in the same order.
        for (int e = 4 * lk0 + line_lane(); e < 4 * lk1; e += NT) {
            const int k = e >> 2, q = e & 3;
            const int b0 = lbeg[k], b1 = lbeg[k + 1];
            const double* lx = c.ws + h.w_ln_x + UVS_LN_X * (size_t)k;
            double t = 0.0;
            constexpr int QB = STREAM ? 8 : 4;      // observations per batch of loads
            const double lxq = STREAM ? lx[q] : 0.0, lx4 = STREAM ? lx[4 + q] : 0.0, lx8 = STREAM ? lx[8 + q] : 0.0, ln0 = STREAM ? line[4 * k + q] : 0.0;
            for (int o = b0; o < b1; o += QB) {
                int fr[QB]; double yv[QB][6];
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const int oc = o + u < b1 ? o + u : o;
                    fr[u] = c.bi[h.i_ln_fj + oc];
                    const double* Y = c.ws + h.w_ln_Y + 24 * (size_t)oc + 6 * q;
#pragma unroll
                    for (int a = 0; a < 6; ++a) yv[u][a] = Y[a];
                }
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    if (o + u >= b1) continue;
#pragma unroll
                    for (int a = 0; a < 6; ++a) t += yv[u][a] * d[16 * fr[u] + a];
                }
            }
            const double dl = -(STREAM ? lxq : lx[q]) - t;
            const double v = (STREAM ? ln0 : line[4 * k + q]) + dl;
            line_c[4 * k + q] = v;
            gd += (STREAM ? lx4 : lx[4 + q]) * (dl + t); dd2 += (STREAM ? lx8 : lx[8 + q]) * dl * dl; step2 += dl * dl; xc2 += v * v;
            if (ltrig_c) sincos(v, ltrig_c + 8 * k + 2 * q, ltrig_c + 8 * k + 2 * q + 1);
        }
    } else
#endif
    for (int k = lk0 + line_lane(); k < lk1; k += NT) {
        const int b0 = lbeg[k], b1 = lbeg[k + 1];
        const double* lx = c.ws + h.w_ln_x + UVS_LN_X * (size_t)k;
        double t[4] = {0.0, 0.0, 0.0, 0.0};
        for (int o = b0; o < b1; o += LNB) {      // LNB observations per batch (LNB x 24 independent loads: one round trip per batch)
            int fr[LNB]; double yv[LNB][24];
#pragma unroll
            for (int u = 0; u < LNB; ++u) {
                const int oc = o + u < b1 ? o + u : o;
                fr[u] = c.bi[h.i_ln_fj + oc];
                const double* Y = c.ws + h.w_ln_Y + 24 * (size_t)oc;
#pragma unroll
                for (int q = 0; q < 24; ++q) yv[u][q] = Y[q];
            }
#pragma unroll
            for (int u = 0; u < LNB; ++u) {
                if (o + u >= b1) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int a = 0; a < 6; ++a) t[q] += yv[u][6 * q + a] * d[16 * fr[u] + a];
            }
        }
        double vn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double dl = -lx[q] - t[q];
            const double v = line[4 * k + q] + dl;
            line_c[4 * k + q] = v; vn[q] = v;
            gd += lx[4 + q] * (dl + t[q]); dd2 += lx[8 + q] * dl * dl; step2 += dl * dl; xc2 += v * v;
        }
        if (ltrig_c) line_trig(vn, ltrig_c + 8 * k);      // sin/cos of the candidate parameters, once per line instead of once per observation
    }
    double s4[4] = {gd, dd2, step2, xc2};
    double mx = 0.0;
    block_reduce(sh, s4, &mx);
    if (tid == 0) {
        sh[L_CTRL + C_MCC] = 0.5 * (s4[1] - s4[0]);     // model_cost_change = -(J d).(r + J d/2) with (H + D) d = -g
        sh[L_CTRL + C_STEP2] = s4[2];
        sh[L_CTRL + C_XC2] = s4[3];
        if (sums_out) { sums_out[0] = s4[0]; sums_out[1] = s4[1]; sums_out[2] = s4[2]; sums_out[3] = s4[3]; }
    }
    __syncthreads();
}

UVS_DEV double ambient_sqnorm(const Ctx& c, const double* x, const double* invd, const double* line) {
    const DevWin& h = *c.hdr;
    const int tid = lane_tid();
    double s = 0.0;
    if (tid < 176) s += x[tid] * x[tid];
    if (tid == 183 && h.td_on) s += x[183] * x[183];
    if (tid >= 176 && tid < 183 && h.ex_on) s += x[tid] * x[tid];
    if (tid >= 184 && tid < 191 && h.relo_on) s += x[tid] * x[tid];
    for (int k = tid; k < h.n_points; k += NT) s += invd[k] * invd[k];
    for (int k = tid; k < 4 * h.n_lines; k += NT) s += line[k] * line[k];
    double s4[4] = {s, 0, 0, 0}, mx = 0.0;
    block_reduce(c.sh, s4, &mx);
    return s4[0];
}

// ------------------------------------------------------------------ the kernel
// setup: IMU whitening matrices W = chol_lower(cov^-1)^T (imu_factor.h:64) and prior H0 = J0^T J0, once per solve.
// One wavefront per IMU block, the whole computation in REGISTERS: lane c < 30 owns column c of the 15 x 30 Gauss-Jordan tableau
// [cov | I], the multipliers of a pivot step are v_readlane broadcasts out of the pivot column's lane.  Same operation order as a
// sequential partial-pivoting Gauss-Jordan followed by a row-wise Cholesky (what the CPU oracle does), so the values agree with
// it to the last bits; only the 15 x 15 inverse crosses lanes once through LDS (columns -> rows).
UVS_DEV void imu_whiten_block(const double* cov, double* W, double* scr /* LDS, 225 doubles */, int lane) {
    double m[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) m[i] = lane < 15 ? cov[i * 15 + lane] : (i == lane - 15 ? 1.0 : 0.0);
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        // partial pivoting: first row with the largest |M[i][k]|, i >= k
        int piv = k; double best = fabs(bcast_lane(m[k], k));
#pragma unroll
        for (int i = k + 1; i < 15; ++i) { const double v = fabs(bcast_lane(m[i], k)); if (v > best) { best = v; piv = i; } }
        piv = __builtin_amdgcn_readfirstlane(piv);
#pragma unroll
        for (int i = k + 1; i < 15; ++i) if (piv == i) { const double t = m[k]; m[k] = m[i]; m[i] = t; }
        const double dinv = 1.0 / bcast_lane(m[k], k);
#pragma unroll
        for (int i = k + 1; i < 15; ++i) {
            const double f = bcast_lane(m[i], k) * dinv;
            if (lane >= k) m[i] -= f * m[k];
        }
    }
#pragma unroll
    for (int k = 14; k >= 0; --k) {
        const double dinv = 1.0 / bcast_lane(m[k], k);
        m[k] *= dinv;
#pragma unroll
        for (int i = 0; i < k; ++i) {
            const double f = bcast_lane(m[i], k);
            m[i] -= f * m[k];
        }
    }
    // inverse: lane 15 + c holds column c -> LDS -> lane r reads row r (its lower triangle)
    if (lane >= 15 && lane < 30) {
#pragma unroll
        for (int i = 0; i < 15; ++i) scr[i * 15 + (lane - 15)] = m[i];
    }
    wave_sync();
    double a[15], Lr[15];
#pragma unroll
    for (int j = 0; j < 15; ++j) { a[j] = (lane < 15) ? scr[lane * 15 + j] : 0.0; Lr[j] = 0.0; }
    // lower Cholesky of the inverse, lanes own rows
#pragma unroll
    for (int j = 0; j < 15; ++j) {
        double dsum = bcast_lane(a[j], j);
        double sdot = a[j];
#pragma unroll
        for (int k = 0; k < j; ++k) { const double ljk = bcast_lane(Lr[k], j); dsum -= ljk * ljk; sdot -= Lr[k] * ljk; }
        const double ljj = sqrt(dsum);
        Lr[j] = (lane == j) ? ljj : (lane > j ? sdot / ljj : 0.0);
    }
    if (lane < 15) {
#pragma unroll
        for (int i = 0; i < 15; ++i) W[i * 15 + lane] = (i <= lane) ? Lr[i] : 0.0;      // W = L^T: W[i][r] = L[r][i]
    }
    wave_sync();
}

UVS_DEV void setup_window(const Ctx& c, double* blob_rw, bool with_prior_image = true, int only_imu_frame = -1) {      // k_evaluate needs the IMU whitening only
    const DevWin& h = *c.hdr;
    const int tid = lane_tid(), lane = tid & 63, wv = tid >> 6;
    for (int b = wv; b < h.n_imu; b += NW) {
        if (only_imu_frame >= 0 && c.bi[h.i_imu + 2 * b] != only_imu_frame) continue;      // marginalization: the one block that touches the departing frame
        const double* blk = blob_rw + h.d_imu + (size_t)b * UVS_IMU_STRIDE;
        imu_whiten_block(blk + UVS_IMU_COV, c.ws + h.w_imu_w + (size_t)b * UVS_IMU_WS, c.sh + L_S + 256 * wv, lane);
    }
    if (h.prior_n > 0 && with_prior_image) {
        {   // The (H0 entry, S offset) table of the assembly (asm_prior): every pair of prior columns (a, b) whose S indices satisfy i >= j, in any order (each S entry takes
            // exactly one of them, so the order of the table does not reach the sums).  Generated here from the column map instead of packed and uploaded per window.
            const int n = h.prior_n, tot = h.n_cimg;
            const int* cm = c.bi + h.i_prior + 80;
            int* tab = (int*)(c.ws + h.w_cimg);
            int* cnt = (int*)(c.sh + L_RED);
            __syncthreads();
            if (tid == 0) *cnt = 0;
            __syncthreads();
            for (int t = tid; t < n * n; t += NT) {
                const int a = t / n, b = t - a * n, i = cm[a], j = cm[b];
                if (i >= 0 && j >= 0 && i >= j) { const int k = atomicAdd(cnt, 1); if (k < tot) { tab[k] = t; tab[tot + k] = sidx(i, j); } }
            }
            __syncthreads();
        }
        // The prior's quadratic form, ONCE per solve: H0 = J0^T J0 (dense n x n, both triangles), g0 = J0^T r0, c0 = r0^T r0 / 2 and diag(H0) by S index, in
        // the workspace.  J0 is staged in LDS (coalesced); the 16 x 16 tiles of H0 in the prior's own column order are a true contraction over the
        // n rows of J0: tile(ta, tb)[r][c] = sum_i J0[i][16 ta + r] J0[i][16 tb + c], ceil(n / 4) v_mfma_f64_16x16x4_f64 per tile, one tile per wave
        // at a time (A[r][k]: lane r + 16k, B[k][c]: lane c + 16k), operands five steps at a time.  Only the lower tiles are computed; a product
        // commutes bitwise and both triangles sum over i in the same order, so the mirrored entries ARE the upper triangle.
        const int n = h.prior_n;
        const double* J0 = blob_rw + h.d_prior;
        double* Jl = c.sh + L_S + 2048;                      // [n][n], beside the whitening scratch
        double* r0l = c.sh + L_S + 1024;                     // [n]
        const int* inv = c.bi + h.i_prior + 80 + UVS_MAX_PRIOR_DIM;      // S index -> prior column
        for (int tb = tid; tb < n * n; tb += 12 * NT) {      // J0 -> LDS, 12 loads per lane in flight (once per solve; the plain loop was one memory round trip per 4 * NT entries)
            double v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) { const int t = tb + u * NT; v[u] = J0[t < n * n ? t : tb]; }
#pragma unroll
            for (int u = 0; u < 12; ++u) { const int t = tb + u * NT; if (t < n * n) Jl[t] = v[u]; }
        }
        if (tid < n) r0l[tid] = J0[n * n + tid];
        __syncthreads();
        double* H0 = c.ws + h.w_prior_h0;
        const int li = lane & 15, lk = lane >> 4;
        const int T = (n + 15) >> 4;
        for (int t = wv; t < (T * (T + 1)) / 2; t += NW) {
            int ta = 0; while (((ta + 1) * (ta + 2)) / 2 <= t) ++ta;
            const int tb = t - (ta * (ta + 1)) / 2;
            const int ca = 16 * ta + li, cb = 16 * tb + li;
            const double* Ja = Jl + (ca < n ? ca : 0); const double* Jb = Jl + (cb < n ? cb : 0);
            d4_t acc = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = 0; k0 < n; k0 += 20) {
                double av[5], bv[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int i = k0 + 4 * u + lk, ic = i < n ? i : n - 1;
                    const double a = Ja[ic * n], bq = Jb[ic * n];                   // unconditional loads + selects
                    av[u] = (ca < n && i < n) ? a : 0.0; bv[u] = (cb < n && i < n) ? bq : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 5; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {      // C layout: row = lk + 4q (tile ta), col = li (tile tb)
                const int ra = 16 * ta + lk + 4 * q;
                if (ra < n && cb < n) { H0[ra * n + cb] = acc[q]; if (ta != tb) H0[cb * n + ra] = acc[q]; }
            }
        }
        if (tid < n) {      // g0 = J0^T r0
            double p4[4] = {0, 0, 0, 0};
            for (int i = 0; i < n; ++i) p4[i & 3] += Jl[i * n + tid] * r0l[i];
            H0[UVS_PH_G0(n) + tid] = (p4[0] + p4[1]) + (p4[2] + p4[3]);
        }
        if (tid == NT - 1) { double s2 = 0.0; for (int i = 0; i < n; ++i) s2 += r0l[i] * r0l[i]; H0[UVS_PH_C0(n)] = 0.5 * s2; }
        if (tid < UVS_RD) {      // diag(J0^T J0) by S index (added to L_HD per linearization)
            const int a = inv[tid];
            double v = 0.0;
            if (a >= 0) for (int i = 0; i < n; ++i) v += Jl[i * n + a] * Jl[i * n + a];
            H0[UVS_PH_HD(n) + tid] = v;
        }
        __threadfence_block();      // the workspace entries are read by other lanes of this workgroup after the caller's barrier
    }
}

struct DebugOut {   // optional dump of the first linearization (uvs_debug_linearize)
    double* S;      // [176*176] dense lower (damped, Schur-reduced)
    double* g;      // [176]
    double* hd;     // [176]
    double* dd;     // [176]
    double* step;   // [176] solution of S y = -g
    double* scal;   // [UVS_DEBUG_SCAL_LEN]: cost, gmax, chol_ok, mcc, step2, -, -, -, phase cycles [8..23], sub-timers [24..31]
};

__global__ __launch_bounds__(NT) void k_solve(char* blobs, const long long* blob_off, double* ws_all, const long long* ws_off,
                                              KOpts o, uvs_report* reports, DebugOut dbg) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int tid = lane_tid(), wdx = blockIdx.x;
    char* blob = blobs + blob_off[wdx];
    Ctx c;
    c.hdr = (const DevWin*)blob; c.bd = (const double*)blob; c.bi = (const int*)blob; c.ws = ws_all + ws_off[wdx]; c.sh = sh; c.o = o;
    const DevWin& h = *c.hdr;
    uvs_report* rep = reports + wdx;
    // ---- init: frames -> LDS, landmark parameters -> workspace buffer 0
    if (tid < UVS_XDIM) sh[L_X + tid] = c.bd[h.d_frames + tid];      // pose[77] sb[99] ex[7] td relo[7]
    for (int k = tid; k < h.n_points; k += NT) c.ws[h.w_invd0 + k] = c.bd[h.d_invd + k];
    for (int k = tid; k < 4 * h.n_lines; k += NT) c.ws[h.w_line0 + k] = c.bd[h.d_line + k];
    for (int k = tid; k < h.n_lines; k += NT) line_trig(c.bd + h.d_line + 4 * k, c.ws + h.w_ltrig0 + 8 * k);
    c.ltrig_ok = 1;
#ifndef UVS_X_NO_PTAB
    stage_prior_tables(c); c.ptab_ok = 1;
#endif
    for (int i = tid; i < (int)(sizeof(uvs_report) / 4); i += NT) ((int*)rep)[i] = 0;
    if (tid < 24) sh[L_PROF + tid] = (tid == 23) ? (double)clock64() : 0.0;
    if (tid < 8) sh[L_WPROF + tid] = 0.0;
    setup_window(c, (double*)blob);
    __syncthreads();
    UVS_PROF(c, P_SETUP);
    int cur = 0;
    double* invd[2] = {c.ws + h.w_invd0, c.ws + h.w_invd1};
    double* line[2] = {c.ws + h.w_line0, c.ws + h.w_line1};
    double radius = o.r0, decr = 2.0;
    double cost = 0.0, gmax = 0.0, x_norm = 0.0;
    int it = 0, invalid = 0, nsucc = 0, term = UVS_TERM_NO_CONVERGENCE, status = UVS_OK;
    // ONE linearize() call site (the kernel is one big inlined body; a second copy doubles the instruction footprint):
    // need_lin is raised at start, after an accepted step (new point) and after a rejected / invalid step (new radius).
    bool need_lin = true, first = true;
    GAcc gacc;      // gather accumulators of the current linearization (48 registers per lane, live across the iteration: a re-damping continues from them)
    int prep_mode = 0;        // how much of lin_prep the next linearization can skip (0 nothing, 1 after an accepted step, 2 after a rejected / invalid one)
    int pending = 0;          // trace slot whose cost / gradient norm the next linearization fills in
    const long long t_wall0 = wall_clock64();
    while (true) {
        if (it >= o.max_it && !first) { term = UVS_TERM_NO_CONVERGENCE; break; }
        if (o.max_ticks > 0 && !first) {      // options.max_solver_time_in_seconds (estimator.cpp:987-991): one lane reads the clock, everybody follows it
            __syncthreads();
            if (tid == 0) sh[L_CTRL + C_TIMEUP] = (wall_clock64() - t_wall0 >= o.max_ticks) ? 1.0 : 0.0;
            __syncthreads();
            if (sh[L_CTRL + C_TIMEUP] != 0.0) { term = UVS_TERM_MAX_TIME; break; }
        }
        if (need_lin) {
#ifndef UVS_X_NO_REDAMP
            if (prep_mode == 2 && !first && h.redamp_ok && o.redamp) relinearize_damping(c, sh + L_X, radius, gacc);
            else
#endif
            linearize(c, sh + L_X, invd[cur], line[cur], first, radius, prep_mode, gacc);
            need_lin = false;
            const double lc = sh[L_CTRL + C_COST];
            gmax = sh[L_CTRL + C_GMAX];
            if (first) {
                const double xn2 = ambient_sqnorm(c, sh + L_X, invd[0], line[0]);
                x_norm = sqrt(xn2); cost = lc; first = false;
                if (tid == 0) { rep->initial_cost = cost; rep->cost[0] = cost; rep->radius[0] = radius; rep->gradient_max_norm[0] = gmax; rep->accepted[0] = 1; }
                if (!isfinite(cost)) { term = UVS_TERM_NUMERIC_FAILURE; status = UVS_ERR_NUMERIC; break; }
            } else if (pending > 0) {           // re-evaluation after an accepted step (HandleSuccessfulStep)
                cost = lc;
                if (tid == 0) { rep->cost[pending] = cost; rep->gradient_max_norm[pending] = gmax; }
            }
            pending = 0;
        }
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (it >= o.max_it) { term = UVS_TERM_NO_CONVERGENCE; break; }
        if (gmax <= o.gtol) { term = UVS_TERM_GRADIENT_TOL; break; }
        if (radius <= o.rmin) { term = UVS_TERM_MIN_RADIUS; break; }
        ++it;
        const int ti = it < UVS_MAX_ITER ? it : UVS_MAX_ITER;
        // ---- step: (S) y = -g
        if (tid < UVS_RD) sh[L_DLT + tid] = -sh[L_G + tid];
        if (o.debug && it == 1 && dbg.S) {
            __syncthreads();
            for (int t = tid; t < UVS_RD * UVS_RD; t += NT) { const int i = t / UVS_RD, j = t - i * UVS_RD; dbg.S[t] = (j <= i) ? sh[L_S + sidx(i, j)] : 0.0; }
            if (tid < UVS_RD) { dbg.g[tid] = sh[L_G + tid]; dbg.hd[tid] = sh[L_HD + tid]; dbg.dd[tid] = sh[L_DD + tid]; }
        }
        UVS_PROF(c, P_MISC);
        UVS_TLOG(c, 30);
        chol_factor(c);
        UVS_PROF(c, P_CHOL);
        UVS_TLOG(c, 31);
        chol_solve(c);
        UVS_PROF(c, P_TRSV);
        UVS_TLOG(c, 32);
        bool ok = sh[L_CTRL + C_CHOLOK] != 0.0;
        backsub_candidate(c, invd[cur], line[cur], invd[cur ^ 1], line[cur ^ 1], 0, h.n_points, 0, h.n_lines, true, nullptr);
        UVS_PROF(c, P_BACKSUB);
        UVS_TLOG(c, 33);
        const double mcc = sh[L_CTRL + C_MCC], step2 = sh[L_CTRL + C_STEP2], xc2 = sh[L_CTRL + C_XC2];
        if (o.debug && it == 1 && dbg.S) {
            if (tid < UVS_RD) dbg.step[tid] = sh[L_DLT + tid];
            if (tid == 0) { dbg.scal[0] = cost; dbg.scal[1] = gmax; dbg.scal[2] = ok ? 1.0 : 0.0; dbg.scal[3] = mcc; dbg.scal[4] = step2; }
        }
        if (!isfinite(mcc) || !isfinite(step2)) ok = false;
        if (tid == 0) rep->model_cost_change[ti] = mcc;
        if (!ok || !(mcc > 0.0)) {    // invalid step (HandleInvalidStep)
            ++invalid;
            radius = radius / decr; decr *= 2.0; need_lin = true; prep_mode = 2;
            if (tid == 0) { rep->accepted[ti] = -1; rep->cost[ti] = cost; rep->candidate_cost[ti] = cost; rep->radius[ti] = radius; rep->gradient_max_norm[ti] = gmax; }
            if (invalid >= o.max_invalid) { term = UVS_TERM_INVALID_STEPS; break; }
            continue;
        }
        invalid = 0;
        // ---- candidate cost
        __syncthreads();
        long long tc_ = clock64();
        stage_rotations(c, sh + L_XC);
        prior_dx(c, sh + L_XC);
        __syncthreads();
        if (o.debug == 1 && tid == 0) { const long long t_ = clock64(); sh[L_WPROF + 0] += (double)(t_ - tc_); tc_ = t_; }
        UVS_TLOG(c, 34);
        double cc_ = prior_quad(c, L_PRC);
        UVS_TLOG(c, 35);
        if (o.debug == 1 && tid == 0) { const long long t_ = clock64(); sh[L_WPROF + 1] += (double)(t_ - tc_); tc_ = t_; }
        cc_ += cost_pass(c, sh + L_XC, invd[cur ^ 1], line[cur ^ 1], 0, h.n_pt_obs, 0, h.n_ln_obs, true);
        UVS_TLOG(c, 36);
        double s4[4] = {cc_, 0, 0, 0}, mx = 0.0;
        block_reduce(sh, s4, &mx);
        UVS_TLOG(c, 37);
        UVS_PROF(c, P_COST);
        double cand = s4[0];
        if (!isfinite(cand)) cand = 1.7976931348623157e308;
        const double step_norm = sqrt(step2);
        const double rel = (cost - cand) / mcc;
        const bool successful = rel > o.min_rel;
        if (tid == 0) { rep->candidate_cost[ti] = cand; rep->step_norm[ti] = step_norm; rep->relative_decrease[ti] = rel; rep->cost[ti] = cost; rep->radius[ti] = radius; rep->gradient_max_norm[ti] = gmax; }
        bool stop = false;
        if (step_norm <= o.ptol * (x_norm + o.ptol)) { term = UVS_TERM_PARAMETER_TOL; stop = true; }
        else if (fabs(cost - cand) <= o.ftol * cost) { term = UVS_TERM_FUNCTION_TOL; stop = true; }
        if (stop && !(o.keep_cand && successful)) break;
        if (successful) {             // HandleSuccessfulStep: x <- x_c; the re-linearization happens at the loop top (skipped when we stop)
            __syncthreads();
            if (tid < UVS_XDIM) sh[L_X + tid] = sh[L_XC + tid];
            cur ^= 1; ++nsucc;
            x_norm = sqrt(xc2);
            { const double t3 = 2.0 * rel - 1.0; radius = radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3); }     // (a generic pow() costs hundreds of instructions)
            radius = fmin(o.rmax, radius);
            decr = 2.0;
            cost = cand;              // replaced by the linearization's own sum if another iteration follows (equal up to summation order)
            need_lin = true; pending = ti; prep_mode = 1;
            if (tid == 0) { rep->accepted[ti] = 1; rep->cost[ti] = cost; rep->radius[ti] = radius; }
            if (stop) break;
        } else {                      // HandleUnsuccessfulStep
            radius = radius / decr; decr *= 2.0; need_lin = true; prep_mode = 2;
            if (tid == 0) { rep->accepted[ti] = 0; rep->radius[ti] = radius; }
        }
    }
    __syncthreads();
    UVS_PROF(c, P_MISC);
    if (o.debug && dbg.scal && tid < P_LAST) dbg.scal[8 + tid] = sh[L_PROF + tid];
    if (o.debug && dbg.scal && tid < 8) dbg.scal[24 + tid] = sh[L_WPROF + tid];
    if (tid < UVS_XDIM) c.ws[h.w_out + tid] = sh[L_X + tid];
    // the accepted landmark parameters follow the frames, so that the host fetches ONE small contiguous block per window
    for (int k = tid; k < h.n_points; k += NT) c.ws[h.w_out + UVS_XDIM + k] = invd[cur][k];
    for (int k = tid; k < 4 * h.n_lines; k += NT) c.ws[h.w_out + UVS_XDIM + h.n_points + k] = line[cur][k];
    if (h.out_host) {      // uvs_batch_stream: the state ALSO goes where the host reads it (DevWin::out_host; posted writes over PCIe: nobody waits for them before the kernel ends)
        double* od = (double*)h.out_host;
        if (tid < UVS_XDIM) od[tid] = sh[L_X + tid];
        for (int k = tid; k < h.n_points; k += NT) od[UVS_XDIM + k] = invd[cur][k];
        for (int k = tid; k < 4 * h.n_lines; k += NT) od[UVS_XDIM + h.n_points + k] = line[cur][k];
    }
    if (tid == 0) {
        rep->status = status; rep->termination = term; rep->num_iterations = it; rep->num_successful = nsucc; rep->final_cost = cost;
        ((DevWin*)blob)->cur_sel = cur;
    }
}

#ifndef UVS_SOLVE_KERNEL_ONLY      // (uvs_solve512.hip instantiates k_solve alone)
// ------------------------------------------------------------------ marginalization on the device (MARGIN_OLD)
// The window here is the SUB-window of the factors that touch the departing frame 0 (marginalization_factor.cpp:174-297 via estimator.cpp:1002-1135: IMU block 0,
// the points anchored in frame 0, the lines that start there without their anchor observation, the prior), packed with a FREE extrinsic (the reference's prior
// keeps para_Ex_Pose whether or not the solve estimates it).  One linearization of it at the post-solve state with an infinite trust-region radius IS the
// assembly A = sum J^T J, b = sum J^T r of the reference followed by the elimination of every dropped landmark block: the kernel's reduced frame system.
// out = [S lower packed (i >= j: i (i + 1) / 2 + j, padded indices) | g[176] | {cost, number of lanes with a landmark pivot <= 1e-8, 0...}[8]].
// The host eliminates the 15 dofs of frame 0 from it and factors the rest (csrc/uvs_marg.h: marg_finish).
static constexpr int MARG_OUT = UVS_RD * (UVS_RD + 1) / 2 + UVS_RD + 8;
UVS_DEV void marg_linearize_body(char* blob, double* ws, KOpts o, double* out, double* sh) {
    const int tid = lane_tid();
    Ctx c;
    c.hdr = (const DevWin*)blob; c.bd = (const double*)blob; c.bi = (const int*)blob; c.ws = ws; c.sh = sh; c.o = o; c.o.debug = 0;
    const DevWin& h = *c.hdr;
    if (tid < UVS_XDIM) sh[L_X + tid] = c.bd[h.d_frames + tid];
    for (int k = tid; k < h.n_points; k += NT) c.ws[h.w_invd0 + k] = c.bd[h.d_invd + k];
    for (int k = tid; k < 4 * h.n_lines; k += NT) c.ws[h.w_line0 + k] = c.bd[h.d_line + k];
    for (int k = tid; k < h.n_lines; k += NT) line_trig(c.bd + h.d_line + 4 * k, c.ws + h.w_ltrig0 + 8 * k);
    c.ltrig_ok = 1;
    if (tid < 24) sh[L_PROF + tid] = 0.0;
    if (tid < 8) sh[L_WPROF + tid] = 0.0;
    setup_window(c, (double*)blob);
    __syncthreads();
    GAcc gacc;
    linearize(c, sh + L_X, c.ws + h.w_invd0, c.ws + h.w_line0, true, o.r0, 0, gacc);
    // smallest pivot of the UNDAMPED landmark blocks (the reference cuts eigenvalues of A_mm at 1e-8: a landmark block that is not safely regular
    // sends the caller to the host path, which applies that cut)
    double worst = 1e300;
    for (int k = tid; k < h.n_points; k += NT) if (c.bi[h.i_pt_beg + k + 1] > c.bi[h.i_pt_beg + k]) worst = fmin(worst, c.ws[h.w_pt_x + 4 * (size_t)k + 3]);
    for (int k = tid; k < h.n_lines; k += NT) {
        if (c.bi[h.i_ln_beg + k + 1] == c.bi[h.i_ln_beg + k]) continue;
        const double* H = c.ws + h.w_ln_x + UVS_LN_X * (size_t)k + 12;      // lower packed (0,0)(1,0)(1,1)(2,0)...
        double L[10];
        const double p0 = H[0]; L[0] = sqrt(fmax(p0, 1e-300));
        L[1] = H[1] / L[0]; const double p1 = H[2] - L[1] * L[1]; L[2] = sqrt(fmax(p1, 1e-300));
        L[3] = H[3] / L[0]; L[4] = (H[4] - L[3] * L[1]) / L[2]; const double p2 = H[5] - L[3] * L[3] - L[4] * L[4]; L[5] = sqrt(fmax(p2, 1e-300));
        L[6] = H[6] / L[0]; L[7] = (H[7] - L[6] * L[1]) / L[2]; L[8] = (H[8] - L[6] * L[3] - L[7] * L[4]) / L[5];
        const double p3 = H[9] - L[6] * L[6] - L[7] * L[7] - L[8] * L[8];
        worst = fmin(worst, fmin(fmin(p0, p1), fmin(p2, p3)));
    }
    double s4[4] = {(worst > 1e-8) ? 0.0 : 1.0, 0, 0, 0}, mx = 0.0;      // lanes that hold a landmark block with a pivot at or under the reference's eps (marginalization_factor.h:70)
    block_reduce(sh, s4, &mx);
    for (int t = tid; t < UVS_RD * (UVS_RD + 1) / 2; t += NT) {
        int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        while (i * (i + 1) / 2 > t) --i;
        const int j = t - i * (i + 1) / 2;
        out[t] = sh[L_S + sidx(i, j)];
    }
    if (tid < UVS_RD) out[UVS_RD * (UVS_RD + 1) / 2 + tid] = sh[L_G + tid];
    if (tid == 0) { double* sc = out + UVS_RD * (UVS_RD + 1) / 2 + UVS_RD; sc[0] = sh[L_CTRL + C_COST]; sc[1] = s4[0]; }
}
__global__ __launch_bounds__(NT) void k_marg_linearize(char* blob, double* ws, KOpts o, double* out) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    marg_linearize_body(blob, ws, o, out, sh);
}
// the same for a BATCH of sub-windows (uvs_marginalize_batch): one workgroup per window, blobs / workspaces through offset tables like k_solve, outputs MARG_OUT doubles apart
__global__ __launch_bounds__(NT) void k_marg_linearize_batch(char* blobs, const long long* blob_off, double* ws_all, const long long* ws_off, KOpts o, double* out_all) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    marg_linearize_body(blobs + blob_off[blockIdx.x], ws_all + ws_off[blockIdx.x], o, out_all + (size_t)MARG_OUT * blockIdx.x, sh);
}

#endif
}  // namespace uvsdev
