// uvs_large_kernel.h -- ONE large window (BASELINE configs[3]: 20 k points + 5 k lines) spread over the whole GPU
// and, through an all-reduce of the pose-block partials, over several GPUs.
//
// Landmarks are conditionally independent given the 11 frame states, so the landmark chunks that k_solve walks
// sequentially inside one workgroup become the GRID here:
//   k_large_chunks   grid = #chunks : stage + Schur prep + gather of one chunk  -> per-chunk partial pose blocks
//   k_large_reduce   sums the partials in a fixed two-level order (deterministic) -> reduced[LG_RED]  (THE all-reduce payload)
//   k_large_solve    1 workgroup    : IMU + prior + damping, Cholesky, step, frame candidate
//   k_large_backsub  grid = #chunks : landmark back-substitution + candidate cost of the chunk's observations
// The LM accept / reject logic runs on the host between launches (one small read-back per iteration).
// Multi-GPU: every rank holds the landmarks k with k % G == rank; `reduced` (pose-pose Schur blocks, reduced gradient,
// diag(J^T J), landmark cost) is summed with ONE RCCL all-reduce (46.7 KB, latency bound), every rank then solves the same
// reduced system redundantly; the back-substitution scalars need a second 5-double all-reduce.
#pragma once
#include "uvs_solve_kernel.h"

namespace uvsdev {

static constexpr int LG_ACC = UVS_NBLKX * 64;               // 5824 accumulator slots [gather block (66 pose blocks + 12 time-offset blocks + 13 extrinsic / relo_Pose blocks)][row a][8] -- canonical, independent of the per-window group balance
static constexpr int LG_RED = LG_ACC + 8;                  // + {landmark cost, max |g_l|, 6 spare}
// Relocalization blocks beside a FREE extrinsic (DevWin::relo2, round 6): the 14 gather blocks of block row 13 (relo_Pose x {11 frames, td, extrinsic, itself}) travel as a TAIL
// behind everything else -- in a partial row at [LG_RED, LG_ROW), in `reduced` at [LG_XCH, LG_XCH_ALL) -- so that no offset, loop bound or all-reduce payload of a window without
// them changes; in the LDS image of large_partial_image they follow the canonical blocks directly (block b at 64 b).  Single rank only (uvs_large_solve_fused rejects several).
static constexpr int LG_R2 = (UVS_NBLKX2 - UVS_NBLKX) * 64;   // 896
static constexpr int LG_ROW = LG_RED + LG_R2;                 // row stride of `partials`
enum { LS_X = 0, LS_XC = UVS_XDIM, LS_DLT = 2 * UVS_XDIM, LS_G = LS_DLT + UVS_RD, LS_DD = LS_G + UVS_RD, LS_SC = LS_DD + UVS_RD, LS_END = LS_SC + UVS_RD };
static constexpr int LG_STATE = 1280;                      // doubles: X[192] XC[192] DLT[176] G[176] DD[176] SC[176]   (X = pose | speedbias | ex_pose | td | relo_pose | pad)
static_assert(LS_END <= LG_STATE, "large-path state vector overflows its allocation");
enum { LO_COST = 0, LO_GMAX, LO_CHOLOK, LO_GD, LO_DD2, LO_STEP2, LO_XC2, LO_FRAMECOST, LO_N };
// ---- device-resident trust-region state of the FUSED loop (uvs_large_solve_fused: no host round trip per iteration).
// The all-reduce payload of a linearization is reduced[0 .. LG_XCH): the LG_RED sums, then LX_X2 = this rank's landmark share of
// ||x||^2 (first iteration only) and LX_GMAX + r = rank r's max |g_landmark| (zero in the other ranks' slots, so that ONE SUM
// all-reduce also delivers the MAX: every rank takes the maximum over the slots afterwards).
static constexpr int LG_MAXRANKS = 8;
// frame image written by the extra workgroup of k_large_chunks: S without the landmark blocks and without damping | gradient | diag(J^T J) | {cost of the frame terms}
static constexpr int FI_S = 0, FI_G = UVS_S_DOUBLES, FI_HD = FI_G + UVS_RD, FI_COST = FI_HD + UVS_RD, LG_FIMG = FI_COST + 8;
static constexpr int LX_X2 = LG_RED, LX_GMAX = LG_RED + 1, LG_XCH = LG_RED + 1 + LG_MAXRANKS + 7;      // 5848 doubles
static constexpr int LG_XCH_ALL = LG_XCH + LG_R2;      // allocation of `reduced` (the relo2 tail behind the exchange vector)
enum { LC_RADIUS = 0, LC_DECR, LC_COST, LC_GMAX, LC_XNORM, LC_FRAME_X2, LC_IT, LC_INVALID, LC_NSUCC, LC_PENDING, LC_TERM, LC_STATUS, LC_FIRST, LC_DONE, LC_SEL, LC_REDAMP, LC_T0, LC_N };      // LC_REDAMP: the last step was rejected / invalid => the next pass re-damps the same linearization
struct LargeCtl { const double* ctl; int rank, nranks; };
// debug timeline of k_large_chunks (KOpts::debug == 7, UVS_LARGE_PROF=<file> in uvs_large_solve_fused): per workgroup 8 stamps of the 100 MHz wall clock
// {start, state + rotations staged, first chunk done, all chunks done, parts summed, partial written, -, chunks taken}; the LAST launch wins
__device__ long long g_large_prof[1024 * 8];
#define UVS_LPROF(k) do { if (o.debug == 7 && tid == 0 && blockIdx.x < 1024) g_large_prof[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)      // ctl == nullptr: the step-wise API (host-side control, arguments as given)


// The LAST workgroup of the grid (blockIdx.x == n_chunk_wgs) carries no landmarks: it builds the FRAME image of the reduced system -- IMU
// tiles, prior, their gradient and cost (and, on the first linearization, the per-solve setup: IMU whitening, prior normal matrix) --
// while the others run the landmark chunks, and writes it to `fimg`; k_large_solve then only adds the reduced landmark blocks and factors.
// (Inside k_large_solve these 45 k cycles sat on the one-workgroup critical path of every iteration.)
// the workgroup's accumulators -> its canonical partial image in LDS ([pose block][row a][8]: 6 block entries, gradient, diag(J^T J)): the parts of a split block are
// summed in a fixed order (gacc_gather_parts), the image is zeroed and the part-0 lanes add their rows.  FIVE workgroup barriers; `gather` = this wave holds
// accumulators (every wave of the 256-thread kernel, waves 4..7 of the 512-thread one, whose evaluator waves only zero their share and keep the barriers)
UVS_DEV void large_partial_image(const Ctx& c, GAcc& A, int grp, bool gather) {
    double* sh = c.sh; const int tid = lane_tid();
    __syncthreads();
    if (gather) gacc_gather_parts(A, grp, sh + L_S); else role_barriers(2);      // (the part exchange has two barriers inside)
    __syncthreads();
    for (int i = tid; i < LG_ACC + LG_R2; i += NT) sh[L_S + i] = 0.0;
    __syncthreads();
    if (gather && grp >= 0 && ((grp >> 9) & 15) == 0) {
        const int r0 = GR * (tid % UVS_GLANES);
        const bool tdrow = ((grp >> 13) & 15) == UVS_NF;       // time-offset blocks: only row 0 is real
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            if (tdrow && (r0 + r) != 0) continue;
            double* Q = sh + L_S + (grp & 255) * 64 + (r0 + r) * 8;
#pragma unroll
            for (int q = 0; q < 6; ++q) Q[q] += A.v[6 * r + q];
            Q[6] += A.g[r]; Q[7] += A.hd[r];
        }
    }
    __syncthreads();
}
__global__ __launch_bounds__(NT) void k_large_chunks(char* blob, double* ws, KOpts o, const double* state, int sel, int first, double radius, double* partials, LargeCtl lc,
                                                     int n_chunk_wgs, double* fimg) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int tid = threadIdx.x;
    if (lc.ctl) { if (lc.ctl[LC_DONE] != 0.0) return; sel = (int)lc.ctl[LC_SEL]; first = (int)lc.ctl[LC_FIRST]; radius = lc.ctl[LC_RADIUS]; }
    Ctx c; c.hdr = (const DevWin*)blob; c.bd = (const double*)blob; c.bi = (const int*)blob; c.ws = ws; c.sh = sh; c.o = o; c.o.debug = 0;
    const DevWin& h = *c.hdr;
    UVS_LPROF(0);
    // after a rejected step (fused loop): same point, new radius -- the landmark partials are UPDATED by the change of their Schur terms
    // (redamp_chunk), the frame image (undamped) stays as it is
    const bool redamp = lc.ctl && lc.ctl[LC_REDAMP] != 0.0 && !first && h.redamp_ok && o.redamp;
    if (tid < UVS_XDIM) sh[L_X + tid] = state[LS_X + tid];
    if ((int)blockIdx.x == n_chunk_wgs) {
        if (redamp) return;
        if (first) setup_window(c, (double*)blob);
        __syncthreads();
        ImuN N; GAcc none; gacc_zero(none);
        double cost = lin_frames(c, sh + L_X, N);
        __syncthreads();
        lin_assemble(c, sh + L_X, first != 0, radius, -1, none, N, 0.0, 0.0, 1);
        for (int i = tid; i < UVS_S_DOUBLES; i += NT) fimg[FI_S + i] = sh[L_S + i];
        if (tid < UVS_RD) { fimg[FI_G + tid] = sh[L_G + tid]; fimg[FI_HD + tid] = sh[L_HD + tid]; }
        double s4[4] = {cost, 0, 0, 0}, mx = 0.0;
        block_reduce(sh, s4, &mx);
        if (tid == 0) fimg[FI_COST] = s4[0];
        UVS_LPROF(5);
        return;
    }
    __syncthreads();
    stage_rotations(c, sh + L_X);
    __syncthreads();
    const int grp = gather_group(c);
    GAcc A;
    if (!ROLES) gacc_zero(A);      // (512-thread instantiation: zeroed inside the role branches, so that the accumulators are dead in the evaluators' code)
    lacc_set(sh, 0.0, 0.0);
    UVS_LPROF(1);
    const double* invd = ws + (sel ? h.w_invd1 : h.w_invd0); const double* line = ws + (sel ? h.w_line1 : h.w_line0);
    // PERSISTENT workgroups: workgroup b takes chunks b, b + gridDim.x, ... and accumulates them into ONE partial (the host sizes the chunks so
    // that their number is a multiple of the grid: 340 LDS-filling chunks on 256 CUs were two full rounds for 1.33 rounds of work)
    int taken = 0;
    if (ROLES) {
        // 512-thread instantiation (csrc/uvs_solve512.hip): the wave roles of the persistent kernel's linearization (linearize_roles) -- waves 0..3 evaluate, waves 4..7
        // copy the lists, compute the landmarks' anchor slots and walk the lists; the same barrier ledger per chunk
        // (two loops per role, re-damping and linearization apart: with both in one loop body the walk's operand sets spilled)
        const int ch0 = blockIdx.x, chs = n_chunk_wgs;
        if (role_eval()) {
            if (redamp) {
                for (int ch = ch0; ch < h.n_chunks; ch += chs) { redamp_prep(c, chunk_desc(c, ch), radius); if (taken++ == 0) UVS_LPROF(2); }
            } else {
                ChunkDesc d = chunk_desc(c, ch0 < h.n_chunks ? ch0 : 0);
                for (int ch = ch0; ch < h.n_chunks; ch += chs) {
                    chunk_eval(c, d, sh + L_X, invd, line, first != 0, radius);
                    if (taken++ == 0) UVS_LPROF(2);
                    if (ch + chs < h.n_chunks) { d = chunk_desc(c, ch + chs); chunk_touch(c, d, invd, line); }
                }
            }
            GAcc none;
            large_partial_image(c, none, -1, false);
        } else {
            GAcc Ag; gacc_zero(Ag);
            GAcc& A = Ag;
            if (redamp) {
                for (int ch = ch0; ch < h.n_chunks; ch += chs) { const ChunkDesc d = chunk_desc(c, ch); role_barriers(redamp_barriers(d)); redamp_gather(c, d, grp, A); ++taken; }
            } else {
                ChunkDesc d = chunk_desc(c, ch0 < h.n_chunks ? ch0 : 0);
                for (int ch = ch0; ch < h.n_chunks; ch += chs) {
                    __syncthreads();      // the chunk's entry barrier: the staging area is free
                    if (LISTS_BY_GATHERERS) copy_lists_gatherers(c, d);
                    AnchorPre ap; ap.b0 = 0; ap.b1 = 0; ap.sc = 1.0;
                    if (ANCHOR_BY_GATHERERS && d.type == 0) pt_anchor_pre(c, d, first != 0, ap);
                    __syncthreads();      // pass A is done
                    if (ANCHOR_BY_GATHERERS && d.type == 0) pt_anchor_pass(c, d, first != 0, radius, ap);
                    role_barriers(chunk_eval_barriers(d) - 2);
                    chunk_gather(c, d, grp, A);
                    ++taken;
                    if (ch + chs < h.n_chunks) d = chunk_desc(c, ch + chs);
                }
            }
            large_partial_image(c, A, grp, true);
        }
        UVS_LPROF(3); UVS_LPROF(4);
    } else
    for (int ch = blockIdx.x; ch < h.n_chunks; ch += n_chunk_wgs) { if (redamp) redamp_chunk(c, ch, radius, grp, A); else lin_chunk(c, ch, sh + L_X, invd, line, first != 0, radius, grp, A); if (taken++ == 0) UVS_LPROF(2); }
    UVS_LPROF(3);
    if (o.debug == 7 && tid == 0 && blockIdx.x < 1024) g_large_prof[8 * blockIdx.x + 7] = taken;
    double cost = lacc_cost(sh), gmax = lacc_gmax(sh);
    if (!ROLES) { large_partial_image(c, A, grp, true); UVS_LPROF(4); }
    double* P = partials + (size_t)blockIdx.x * LG_ROW;
    if (redamp) {      // cost and landmark gradient norm of the partial are unchanged
        for (int i = tid; i < LG_ACC; i += NT) P[i] += sh[L_S + i];
        if (h.relo2) for (int i = tid; i < LG_R2; i += NT) P[LG_RED + i] += sh[L_S + LG_ACC + i];
        return;
    }
    for (int i = tid; i < LG_ACC; i += NT) P[i] = sh[L_S + i];
    if (h.relo2) for (int i = tid; i < LG_R2; i += NT) P[LG_RED + i] = sh[L_S + LG_ACC + i];
    double s4[4] = {cost, 0, 0, 0};
    block_reduce(sh, s4, &gmax);
    if (tid == 0) { P[LG_ACC] = s4[0]; P[LG_ACC + 1] = gmax; }
    UVS_LPROF(5);
}

// Deterministic two-level sum of the per-chunk partials: a workgroup owns 16 consecutive entries i, its 16 x 16 threads split the chunk
// range into 16 contiguous slices (summed in chunk order, loads independent of each other), the 16 slice sums are then added in slice
// order.  (One thread per entry walking all chunks serially took 137 us for 340 chunks -- more than k_large_chunks itself.)
__global__ __launch_bounds__(256) void k_large_reduce(const double* partials, int n_chunks, double* reduced, LargeCtl lc, int n_ent /* LG_RED, or LG_ROW with the relo2 tail */) {
    __shared__ double part[16][17];
    if (lc.ctl && lc.ctl[LC_DONE] != 0.0) return;
    const int il = threadIdx.x & 15, p = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;
    const bool is_max = (i == LG_ACC + 1);
    const int per = (n_chunks + 15) / 16, c0 = p * per, c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;
    double s = 0.0;
    if (i < n_ent) {      // (entries >= LG_RED: the relo2 tail)
        for (int cb = c0; cb < c1; cb += 16) {      // 16 rows per batch of loads, summed in row order (the plain loop compiled to a few loads per memory round trip)
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = partials[(size_t)(cb + u < c1 ? cb + u : cb) * LG_ROW + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (cb + u < c1) s = is_max ? fmax(s, v[u]) : s + v[u];
        }
    }
    part[p][il] = s;
    __syncthreads();
    if (p == 0 && i < n_ent) {
        double t = part[0][il];
        for (int q = 1; q < 16; ++q) t = is_max ? fmax(t, part[q][il]) : t + part[q][il];      // fixed order => deterministic
        reduced[i < LG_RED ? i : LG_XCH + (i - LG_RED)] = t;
        // fused multi-GPU exchange: the landmark gradient max-norm travels in this rank's slot of a SUM all-reduce
        if (is_max && lc.ctl) for (int r = 0; r < LG_MAXRANKS; ++r) reduced[LX_GMAX + r] = (r == lc.rank) ? t : 0.0;
    }
}

// one workgroup: frame terms + assembly + Cholesky + step.  `reduced` holds the (all-reduced) landmark partials.
__global__ __launch_bounds__(NT) void k_large_solve(char* blob, double* ws, KOpts o, double* state, const double* reduced, int first, double radius, double* out, LargeCtl lc,
                                                    const double* fimg) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int tid = threadIdx.x;
    if (lc.ctl && lc.ctl[LC_DONE] != 0.0) return;
    double gmax_lm = reduced[LG_ACC + 1];
    if (lc.ctl) {
        first = (int)lc.ctl[LC_FIRST]; radius = lc.ctl[LC_RADIUS];
        gmax_lm = 0.0;
        for (int r = 0; r < LG_MAXRANKS; ++r) gmax_lm = fmax(gmax_lm, reduced[LX_GMAX + r]);
    }
    Ctx c; c.hdr = (const DevWin*)blob; c.bd = (const double*)blob; c.bi = (const int*)blob; c.ws = ws; c.sh = sh; c.o = o; c.o.debug = 0;
    if (tid < UVS_XDIM) sh[L_X + tid] = state[LS_X + tid];
    if (tid < UVS_RD && !first) sh[L_SC + tid] = state[LS_SC + tid];
    // the frame image (k_large_chunks' extra workgroup): 147 KB, 16-byte loads, everything in flight before the first store
    {
        constexpr int N2 = UVS_S_DOUBLES / 2, PER = (N2 + NT - 1) / NT;
        d2_t v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) { const int i = tid + u * NT; v[u] = *(const d2_t*)(fimg + FI_S + 2 * (i < N2 ? i : 0)); }
#pragma unroll
        for (int u = 0; u < PER; ++u) { const int i = tid + u * NT; if (i < N2) *(d2_t*)(sh + L_S + 2 * i) = v[u]; }
        if (tid < UVS_RD) { sh[L_G + tid] = fimg[FI_G + tid]; sh[L_HD + tid] = fimg[FI_HD + tid]; }
    }
    const int grp = gather_group(c);
    GAcc A;
    {
        const int r0 = GR * (tid % UVS_GLANES);
        const bool ld = grp >= 0 && !((grp >> 9) & 15);       // part 0 carries the whole (already summed) block
#pragma unroll
        for (int r = 0; r < GR; ++r) {
            const int gb = grp & 255;
            const double* Q = reduced + (gb < UVS_NBLKX ? gb * 64 : LG_XCH + (gb - UVS_NBLKX) * 64) + (r0 + r) * 8;
#pragma unroll
            for (int q = 0; q < 6; ++q) A.v[6 * r + q] = ld ? Q[q] : 0.0;
            A.g[r] = ld ? Q[6] : 0.0; A.hd[r] = ld ? Q[7] : 0.0;
        }
    }
    ImuN N;      // unused in mode 2
    if (c.hdr->relo2) for (int t = tid; t < R2_SC; t += NT) c.ws[c.hdr->w_relo2 + t] = 0.0;      // side buffer of the second-level relo_Pose (asm_zero does this for k_solve; lin_assemble's first barrier precedes the rows' stores)
    const double cost = tid == 0 ? fimg[FI_COST] + reduced[LG_ACC] : 0.0;
    lin_assemble(c, sh + L_X, first != 0, radius, grp, A, N, cost, gmax_lm, 2);
    if (tid < UVS_RD) sh[L_DLT + tid] = -sh[L_G + tid];
    chol_factor(c);
    chol_solve(c);
    backsub_candidate(c, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, true, out + LO_GD);      // frames only
    if (tid < UVS_RD) { state[LS_DLT + tid] = sh[L_DLT + tid]; state[LS_G + tid] = sh[L_G + tid]; state[LS_DD + tid] = sh[L_DD + tid]; state[LS_SC + tid] = sh[L_SC + tid]; }
    if (tid < UVS_XDIM) state[LS_XC + tid] = sh[L_XC + tid];
    if (tid == 0) { out[LO_COST] = sh[L_CTRL + C_COST]; out[LO_GMAX] = sh[L_CTRL + C_GMAX]; out[LO_CHOLOK] = sh[L_CTRL + C_CHOLOK]; }
    // (the frame part of the candidate cost -- prior + IMU at x_c -- is k_large_backsub's extra workgroup)
}

#ifndef UVS_TU_512      // (csrc/uvs_solve512.hip instantiates k_solve, k_large_chunks and k_large_solve; the kernels below exist with 256 threads only)
// per chunk: landmark back-substitution (candidate parameters into the other buffer) + candidate cost of the chunk's observations.
// Round 4: the kernel needs no staging area (only the small LDS arrays + the frame workgroup's scratch: LDS_BYTES_BACKSUB), so several workgroups share a
// compute unit; the register budget is halved for that (UVS_LARGE_OCC waves per SIMD) and the streaming loops keep fewer loads in flight per lane -- the
// other resident waves cover the latency that one wave per SIMD had to cover with its own batches.
#ifndef UVS_LARGE_PSB
#define UVS_LARGE_PSB 8      // Schur slots of a point per batch of loads in k_large_backsub (a typical track has 5 - 7: one round trip)
#endif
#ifndef UVS_LARGE_OCC
#define UVS_LARGE_OCC 2
#endif
static constexpr size_t LDS_BYTES_BACKSUB = (size_t)(L_S + 2048) * 8;      // prior_quad's partials [0, 512), the IMU residual scratch of cost_pass at 1024
__global__ __launch_bounds__(NT, UVS_LARGE_OCC) void k_large_backsub(char* blob, double* ws, KOpts o, const double* state, int sel, double* bsums, LargeCtl lc, int n_chunk_wgs, double* out) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int tid = threadIdx.x;
    if (lc.ctl) { if (lc.ctl[LC_DONE] != 0.0) return; sel = (int)lc.ctl[LC_SEL]; }
    Ctx c; c.hdr = (const DevWin*)blob; c.bd = (const double*)blob; c.bi = (const int*)blob; c.ws = ws; c.sh = sh; c.o = o; c.o.debug = 0;
    const DevWin& h = *c.hdr;
    if (tid < UVS_XDIM) sh[L_XC + tid] = state[LS_XC + tid];
    if (tid < UVS_RD) sh[L_DLT + tid] = state[LS_DLT + tid];
    __syncthreads();
    stage_rotations(c, sh + L_XC);
    if ((int)blockIdx.x == n_chunk_wgs) {      // the LAST workgroup: frame part of the candidate cost (prior + IMU at x_c), beside the landmark chunks
        prior_dx(c, sh + L_XC);
        __syncthreads();
        double cc = prior_quad(c) + cost_pass<1, 1>(c, sh + L_XC, nullptr, nullptr, 0, 0, 0, 0, true);
        double s4[4] = {cc, 0, 0, 0}, mx = 0.0;
        block_reduce(sh, s4, &mx);
        if (tid == 0) out[LO_FRAMECOST] = s4[0];
        return;
    }
    double* invd = ws + (sel ? h.w_invd1 : h.w_invd0); double* line = ws + (sel ? h.w_line1 : h.w_line0);
    double* invd_c = ws + (sel ? h.w_invd0 : h.w_invd1); double* line_c = ws + (sel ? h.w_line0 : h.w_line1);
    const int* pbeg = c.bi + h.i_pt_beg; const int* lbeg = c.bi + h.i_ln_beg;
    for (int ch = blockIdx.x; ch < h.n_chunks; ch += n_chunk_wgs) {      // persistent workgroups, as in k_large_chunks; the sums stay per chunk (8 doubles)
        const int* chunk = c.bi + h.i_chunks + UVS_CHUNK_INTS * ch;
        const int type = chunk[0], k0 = chunk[1], k1 = chunk[2];
        double* bo = bsums + 8 * (size_t)ch;
#ifndef UVS_X_NO_LARGE_TOUCH
        // the first loads of the cost pass below (index words and measurements of this lane's observations) do not depend on the back substitution: requested here,
        // their memory round trip (HBM latency in a window of this size) runs beside the back substitution's own chain; dropped after it, the lines stay in L1 / L2
        double tacc = 0.0; int tiacc = 0;
        {
            const int obA = chunk[6], obB = obA + chunk[7];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int o = obA + tid + u * NT;
                if (o < obB) {
                    if (type == 0) { tiacc += c.bi[h.i_pt_lm + o] + c.bi[h.i_pt_fi + o] + c.bi[h.i_pt_fj + o]; const double* m = c.bd + h.d_ptmeas + o; for (int q = 0; q < 6; ++q) tacc += m[q * h.pt_stride]; }
                    else { tiacc += c.bi[h.i_ln_lm + o] + c.bi[h.i_ln_fj + o] + c.bi[h.i_ln_vp + o]; const double* m = c.bd + h.d_lnmeas + o; for (int q = 0; q < 9; ++q) tacc += m[q * h.ln_stride]; }
                }
            }
        }
#endif
        backsub_candidate<2, UVS_LARGE_PSB, true>(c, invd, line, invd_c, line_c, type == 0 ? k0 : 0, type == 0 ? k1 : 0, type == 1 ? k0 : 0, type == 1 ? k1 : 0, false, bo);
#ifndef UVS_X_NO_LARGE_TOUCH
        asm volatile("" :: "v"(tacc), "v"(tiacc));
#endif
        __threadfence_block();
        __syncthreads();
        const int ob0 = chunk[6], ob1 = ob0 + chunk[7];      // the chunk's observation range (descriptor: no dependent loads from the CSR arrays)
        const int po0 = type == 0 ? ob0 : 0, po1 = type == 0 ? ob1 : 0, lo0 = type == 1 ? ob0 : 0, lo1 = type == 1 ? ob1 : 0;
        double cc = cost_pass<2, 1>(c, sh + L_XC, invd_c, line_c, po0, po1, lo0, lo1, false);
        double s4[4] = {cc, 0, 0, 0}, mx = 0.0;
        block_reduce(sh, s4, &mx);
        if (tid == 0) bo[4] = s4[0];
    }
}
// out5[5] (fused loop with a communicator): THIS rank's vote "options.max_solver_time_in_seconds is used up" -- the SUM all-reduce of the 8 doubles turns the votes
// into one number that is the same on every rank, and k_large_decide ends the solve when it is non-zero: a rank that tested its own clock could stop an
// iteration before its peers and leave stale sums in the collectives they have already enqueued (round-3 advisor finding).
__global__ __launch_bounds__(256) void k_large_sum_bsums(const double* bsums, int n_chunks, double* out5, LargeCtl lc, long long max_ticks) {
    if (lc.ctl && lc.ctl[LC_DONE] != 0.0) return;
    if (threadIdx.x == 255) { out5[5] = (lc.ctl && max_ticks > 0 && lc.ctl[LC_IT] > 0.0 && (double)wall_clock64() - lc.ctl[LC_T0] >= (double)max_ticks) ? 1.0 : 0.0; out5[6] = 0.0; out5[7] = 0.0; }
    // 5 scalars x n_chunks: 32 contiguous chunk slices per scalar (8 lanes idle per slice row), slice sums added in slice order
    __shared__ double part[32][8];
    const int i = threadIdx.x & 7, p = threadIdx.x >> 3;
    const int per = (n_chunks + 31) / 32, c0 = p * per, c1 = (c0 + per < n_chunks) ? c0 + per : n_chunks;
    double s = 0.0;
    if (i < 5) for (int cb = c0; cb < c1; cb += 16) {      // (16 rows per batch of loads, as in k_large_decide)
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = bsums[8 * (size_t)(cb + u < c1 ? cb + u : cb) + i];
#pragma unroll
        for (int u = 0; u < 16; ++u) if (cb + u < c1) s += v[u];
    }
    part[p][i] = s;
    __syncthreads();
    if (p == 0 && i < 5) { double t = part[0][i]; for (int q = 1; q < 32; ++q) t += part[q][i]; out5[i] = t; }
}

// The trust-region bookkeeping of one iteration ON THE DEVICE (same order of tests as k_solve / uvs_large_decide / SURVEY.md Appendix B):
// reads the frame part (out[LO_*], identical on every rank) and the all-reduced landmark scalars sc5 = {g.delta, delta D delta, |delta|^2,
// |x_c|^2, candidate cost}, updates ctl / the report and, on acceptance, x <- x_c.  One workgroup; every rank runs it on identical inputs.
// `bsums` != nullptr (fused loop without a communicator): the 5 landmark scalars are summed here, in k_large_sum_bsums' order, instead of by a launch of their own.
__global__ __launch_bounds__(256) void k_large_decide(double* ctl, double* state, const double* out, const double* sc5_in, const double* reduced, KOpts o, uvs_report* rep,
                                                      const double* bsums, int n_rows) {
    __shared__ int accept_sh;
    __shared__ double part[32][8];
    __shared__ double sc5_sh[8];
    const int tid = threadIdx.x;
    if (ctl[LC_DONE] != 0.0) return;
    if (bsums) {
        const int i = tid & 7, p = tid >> 3;
        const int per = (n_rows + 31) / 32, c0 = p * per, c1 = (c0 + per < n_rows) ? c0 + per : n_rows;
        double sl = 0.0;
        if (i < 5) for (int cb = c0; cb < c1; cb += 16) {      // 16 rows per batch of loads (the plain loop paid a memory round trip per row: 16 of them for the 510 rows of configs[3]); same order of the sum
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = bsums[8 * (size_t)(cb + u < c1 ? cb + u : cb) + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) if (cb + u < c1) sl += v[u];
        }
        part[p][i] = sl;
        __syncthreads();
        if (p == 0 && i < 5) { double t = part[0][i]; for (int q = 1; q < 32; ++q) t += part[q][i]; sc5_sh[i] = t; }
    } else if (tid < 6) sc5_sh[tid] = sc5_in[tid];
    __syncthreads();
    const double* sc5 = sc5_sh;
    if (tid == 0) {
        accept_sh = 0;
        double radius = ctl[LC_RADIUS], decr = ctl[LC_DECR], cost = ctl[LC_COST], gmax = ctl[LC_GMAX], x_norm = ctl[LC_XNORM];
        int it = (int)ctl[LC_IT], invalid = (int)ctl[LC_INVALID], nsucc = (int)ctl[LC_NSUCC], pending = (int)ctl[LC_PENDING], sel = (int)ctl[LC_SEL];
        int term = UVS_TERM_NO_CONVERGENCE, status = UVS_OK; bool done = false;
        // every global value this lane needs, requested BEFORE the first store below (the report and control words may alias them as far as the compiler knows:
        // a load placed after a store waits for a memory round trip of its own -- a dozen of them in a row on one lane were most of this launch's 4 - 6 us)
        const double lc_ = out[LO_COST], gm = out[LO_GMAX];
        const double o_gd = out[LO_GD], o_dd2 = out[LO_DD2], o_step2 = out[LO_STEP2], o_xc2 = out[LO_XC2], o_fcost = out[LO_FRAMECOST], o_cholok = out[LO_CHOLOK];
        const double c_first = ctl[LC_FIRST], c_fx2 = ctl[LC_FRAME_X2], c_t0 = ctl[LC_T0], r_x2 = reduced[LX_X2];
        if (c_first != 0.0) {
            cost = lc_; gmax = gm; ctl[LC_FIRST] = 0.0;
            x_norm = sqrt(c_fx2 + r_x2);      // frames (host) + every rank's landmarks (all-reduced)
            rep->initial_cost = lc_; rep->cost[0] = lc_; rep->radius[0] = radius; rep->gradient_max_norm[0] = gm; rep->accepted[0] = 1;
            if (!isfinite(lc_)) { term = UVS_TERM_NUMERIC_FAILURE; status = UVS_ERR_NUMERIC; done = true; }
        } else if (pending > 0) { cost = lc_; gmax = gm; rep->cost[pending] = lc_; rep->gradient_max_norm[pending] = gm; }
        pending = 0;
        if (!done) {
            if (it >= o.max_it) { term = UVS_TERM_NO_CONVERGENCE; done = true; }
            // options.max_solver_time_in_seconds: one process reads its own clock; with a communicator the decision is the all-reduced vote of the ranks (sc5[5]), identical everywhere
            else if (o.max_ticks > 0 && it > 0 && (bsums ? (double)wall_clock64() - c_t0 >= (double)o.max_ticks : sc5[5] > 0.0)) { term = UVS_TERM_MAX_TIME; done = true; }
            else if (gmax <= o.gtol) { term = UVS_TERM_GRADIENT_TOL; done = true; }
            else if (radius <= o.rmin) { term = UVS_TERM_MIN_RADIUS; done = true; }
        }
        if (!done) {
            ++it;
            const int ti = it < UVS_MAX_ITER ? it : UVS_MAX_ITER;
            const double gd = o_gd + sc5[0], dd2 = o_dd2 + sc5[1], step2 = o_step2 + sc5[2], xc2 = o_xc2 + sc5[3];
            const double mcc = 0.5 * (dd2 - gd);
            double cand = o_fcost + sc5[4];
            const bool ok = o_cholok != 0.0 && isfinite(mcc) && isfinite(step2);
            rep->model_cost_change[ti] = mcc;
            if (!ok || !(mcc > 0.0)) {
                ++invalid; radius /= decr; decr *= 2.0; ctl[LC_REDAMP] = 1.0;
                rep->accepted[ti] = -1; rep->cost[ti] = cost; rep->candidate_cost[ti] = cost; rep->radius[ti] = radius; rep->gradient_max_norm[ti] = gmax;
                if (invalid >= o.max_invalid) { term = UVS_TERM_INVALID_STEPS; done = true; }
                else if (it >= o.max_it) { term = UVS_TERM_NO_CONVERGENCE; done = true; }      // the last enqueued pass may be an invalid step: it still ends the solve (k_solve tests this at its loop top)
            } else {
                invalid = 0;
                if (!isfinite(cand)) cand = 1.7976931348623157e308;
                const double step_norm = sqrt(step2), rel = (cost - cand) / mcc;
                const bool successful = rel > o.min_rel;
                rep->candidate_cost[ti] = cand; rep->step_norm[ti] = step_norm; rep->relative_decrease[ti] = rel; rep->cost[ti] = cost; rep->radius[ti] = radius; rep->gradient_max_norm[ti] = gmax;
                bool stop = false;
                if (step_norm <= o.ptol * (x_norm + o.ptol)) { term = UVS_TERM_PARAMETER_TOL; stop = true; }
                else if (fabs(cost - cand) <= o.ftol * cost) { term = UVS_TERM_FUNCTION_TOL; stop = true; }
                if (stop && !(o.keep_cand && successful)) done = true;
                else if (successful) {
                    accept_sh = 1; ctl[LC_REDAMP] = 0.0;
                    sel ^= 1; ++nsucc; x_norm = sqrt(xc2);
                    { const double t3 = 2.0 * rel - 1.0; radius = radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3); }
                    radius = fmin(o.rmax, radius); decr = 2.0;
                    cost = cand; pending = ti;
                    rep->accepted[ti] = 1; rep->cost[ti] = cost; rep->radius[ti] = radius;
                    if (stop || it >= o.max_it) { if (!stop) term = UVS_TERM_NO_CONVERGENCE; done = true; }
                } else {
                    radius /= decr; decr *= 2.0; ctl[LC_REDAMP] = 1.0;
                    rep->accepted[ti] = 0; rep->radius[ti] = radius;
                    if (it >= o.max_it) { term = UVS_TERM_NO_CONVERGENCE; done = true; }
                }
            }
        }
        ctl[LC_RADIUS] = radius; ctl[LC_DECR] = decr; ctl[LC_COST] = cost; ctl[LC_GMAX] = gmax; ctl[LC_XNORM] = x_norm;
        ctl[LC_IT] = it; ctl[LC_INVALID] = invalid; ctl[LC_NSUCC] = nsucc; ctl[LC_PENDING] = pending; ctl[LC_SEL] = sel;
        if (done) {
            ctl[LC_DONE] = 1.0; ctl[LC_TERM] = term; ctl[LC_STATUS] = status;
            rep->status = status; rep->termination = term; rep->num_iterations = it; rep->num_successful = nsucc; rep->final_cost = cost;
        }
    }
    __syncthreads();
    if (accept_sh && tid < UVS_XDIM) state[LS_X + tid] = state[LS_XC + tid];
}

// Start of a fused solve, on the device: state <- frames of the blob, landmark buffers 0 <- blob, control words and report reset.  Replaces five
// small host-side copies / memsets (each a separate stream operation) by one launch.
__global__ __launch_bounds__(256) void k_large_init(const char* blob, double* ws, double* state, double* ctl, uvs_report* rep, double* reduced,
                                                    double radius0, double frame_x2, double local_x2) {
    const DevWin& h = *(const DevWin*)blob; const double* bd = (const double*)blob;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int i = t; i < LG_STATE; i += nt) state[i] = (i >= LS_X && i < LS_X + UVS_XDIM) ? bd[h.d_frames + (i - LS_X)] : 0.0;
    for (int i = t; i < h.n_points; i += nt) ws[h.w_invd0 + i] = bd[h.d_invd + i];
    for (int i = t; i < 4 * h.n_lines; i += nt) ws[h.w_line0 + i] = bd[h.d_line + i];
    for (int i = t; i < LG_XCH; i += nt) reduced[i] = (i == LX_X2) ? local_x2 : 0.0;
    for (int i = t; i < 64; i += nt) ctl[i] = i == LC_RADIUS ? radius0 : i == LC_DECR ? 2.0 : i == LC_FIRST ? 1.0 : i == LC_FRAME_X2 ? frame_x2 : i == LC_T0 ? (double)wall_clock64() : 0.0;
    for (int i = t; i < (int)(sizeof(uvs_report) / 4); i += nt) ((int*)rep)[i] = 0;
}
// End of a fused solve: everything the host reads back, gathered into one buffer [ctl 64 | report | frames 192 | inverse depths | line parameters]
// (the landmark buffer that holds the accepted values is only known on the device: ctl[LC_SEL])
__global__ __launch_bounds__(256) void k_large_pack(const char* blob, const double* ws, const double* state, const double* ctl, const uvs_report* rep, double* out) {
    const DevWin& h = *(const DevWin*)blob;
    constexpr int RD = (int)(sizeof(uvs_report) / 8);
    const int sel = (int)ctl[LC_SEL];
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int i = t; i < 64; i += nt) out[i] = ctl[i];
    for (int i = t; i < RD; i += nt) out[64 + i] = ((const double*)rep)[i];
    for (int i = t; i < UVS_XDIM; i += nt) out[64 + RD + i] = state[LS_X + i];
    double* o2 = out + 64 + RD + UVS_XDIM;
    for (int i = t; i < h.n_points; i += nt) o2[i] = ws[(sel ? h.w_invd1 : h.w_invd0) + i];
    for (int i = t; i < 4 * h.n_lines; i += nt) o2[h.n_points + i] = ws[(sel ? h.w_line1 : h.w_line0) + i];
}

#endif
}  // namespace uvsdev
