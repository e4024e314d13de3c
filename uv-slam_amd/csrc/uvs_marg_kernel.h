// uvs_marg_kernel.h -- the BACK HALF of a marginalization on the device, batched over windows (round 6; SURVEY.md section 7: k_marg_assemble / k_marg_schur / k_sym_eig).
//
// Replaces, for a batch of windows, what uvs_marg.h: marg_finish does on one host core per window (marginalization_factor.cpp:263-291): elimination of the dropped FRAME block
// (Pose[0] + SpeedBias[0] for MARGIN_OLD, Pose[WINDOW_SIZE - 1] for MARGIN_SECOND_NEW; <= 15 dofs) from the assembled, landmark-eliminated system, the Schur complement onto the
// kept blocks, the symmetric eigen-decomposition of the n x n result (n <= 76 in the reference) and the factor J0 = sqrt(S) V^T, r0 = sqrt(S^-1) V^T b with eigenvalues <= 1e-8
// cut (marginalization_factor.h:70).  One workgroup per window, everything in LDS.
//
// The eigen-decomposition is a PARALLEL cyclic Jacobi (round-robin ordering: n / 2 disjoint rotations per step, n - 1 steps per sweep) with the same rotation formulas and the
// same RELATIVE stopping rule |a_pq| <= 1.1e-16 sqrt(|a_pp a_qq|) as the host's host_sym_eig_jacobi (uvs_marg.h:30-58): the matrices are graded over twenty orders of magnitude
// and Jacobi resolves the small eigenvalues relative to their own scale.  For ONE window this is much slower than the host's tridiagonal QL (1.6 ms against ~0.2 ms): a rotation step
// is two workgroup barriers and there are ~1 000 of them, whether one workgroup runs or 256.  For a BATCH it is what makes the marginalization scale with the solve: 256 windows take one launch instead
// of 256 x 0.2 ms on a host core (uvs_marginalize_batch).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/uvs_solver.h"

namespace uvsmarg {

static constexpr int MF_NT = 704, MF_CT = 640;           // ten compute waves = sixteen lanes for each of the <= 40 rotations of a step (MF_NKEEP / 2), threads >= MF_CT = the control wave (rotation parameters of the next step)
static constexpr int MF_NMAX = 96;            // N = md + n the device path takes (the reference's largest: 15 + 76 = 91)
static constexpr int MF_NKEEP = 80;           // n it takes
static constexpr int MF_LDA = MF_NMAX + 1;    // odd row strides: a column walk touches every LDS bank
static constexpr int MF_LDV = MF_NKEEP + 1;
static constexpr int MF_MD = 15;
// per window: input descriptor (ints), dense input (doubles, mode 1), output (doubles)
static constexpr int MF_DESC = 128;           // {N, md, n, mode, map[N]}: mode 0 = rows gathered from a packed lower triangle + gradient (k_marg_linearize's output; map = padded index of row i),
                                              //                             mode 1 = dense A [N][N] row-major followed by b [N]
static constexpr int MF_IN = MF_NMAX * MF_NMAX + MF_NMAX;
static constexpr int MF_OUT = UVS_MAX_PRIOR_DIM * UVS_MAX_PRIOR_DIM + UVS_MAX_PRIOR_DIM + 8;      // status words at MF_OUT_S | r0 [n] at MF_OUT_R | J0 [n][n] (leading dimension n) at MF_OUT_J:
static constexpr int MF_OUT_S = 0, MF_OUT_R = 8, MF_OUT_J = 8 + UVS_MAX_PRIOR_DIM;                 // the used part of a slot is its HEAD (104 + n^2 doubles): the host copies only that much of every slot
enum { MF_OK = 0, MF_IRREGULAR = 1, MF_NONFINITE = 2, MF_UNCONVERGED = 3 };      // status[0]; status[1] = sweeps, status[2] = rotations, status[3] = eigenvalues cut
static constexpr int MF_NP = MF_NKEEP / 2 + 1;   // rotations of a step
static constexpr int MF_RS = (MF_NKEEP + 15) / 16;   // lane-strides of a row
static constexpr int MF_NBLK = ((MF_NKEEP / 2) * (MF_NKEEP / 2 + 1) / 2 + 1) & ~1;   // 2 x 2 blocks (i >= j) of a step's pairs
static_assert(16 * (MF_NKEEP / 2) <= MF_CT && MF_CT % 64 == 0 && MF_NT - MF_CT >= MF_NKEEP / 2 + 1, "sixteen lanes per pair, the control wave holds a lane per pair");
static constexpr size_t MF_LDS_DOUBLES = (size_t)MF_NMAX * MF_LDA + (size_t)MF_NKEEP * MF_LDV + (size_t)MF_MD * (MF_NKEEP + 2) + 4 * MF_NMAX + 14 * MF_NP + 16 + MF_NBLK / 2;      // (16: control words, 8 doubles used; then the block table of the A pass, one int per block)
static constexpr size_t MF_LDS_BYTES = MF_LDS_DOUBLES * 8;

// 1 / x and 1 / sqrt(x) from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-26 relative) + two Newton steps (quadratic: 2^-52 after the first, the second absorbs the seed's worst case)
__device__ __forceinline__ double mf_rcp(double x) { double y = __builtin_amdgcn_rcp(x); double e = fma(-x, y, 1.0); y = fma(y, e, y); e = fma(-x, y, 1.0); return fma(y, e, y); }
__device__ __forceinline__ double mf_rsq(double x) { double y = __builtin_amdgcn_rsq(x); double e = fma(-x * y, y, 1.0); y = fma(0.5 * y, e, y); e = fma(-x * y, y, 1.0); return fma(0.5 * y, e, y); }
__global__ __launch_bounds__(MF_NT) void k_marg_finish(const int* __restrict__ desc_all, const double* __restrict__ in_all, const double* __restrict__ lin_all, int lin_stride,
                                                       int tri_n, double* __restrict__ out_all, double eps) {
    extern __shared__ __attribute__((aligned(16))) double sh[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int* desc = desc_all + (size_t)MF_DESC * b;
    const int N = desc[0], md = desc[1], n = desc[2], mode = desc[3];
    double* out = out_all + (size_t)MF_OUT * b;
    double* A = sh;                                        // [N][MF_LDA]; later the n x n Schur complement in its top-left corner
    double* Vt = A + MF_NMAX * MF_LDA;                     // [n][MF_LDV]: row k = eigenvector k
    double* X = Vt + MF_NKEEP * MF_LDV;                    // [md][n + 1] = S_dd^-1 [A_dr | b_d]   (row stride n + 1)
    double* bv = X + MF_MD * (MF_NKEEP + 2);               // [N]
    double* br = bv + MF_NMAX;                             // [n]
    double* lam = br + MF_NMAX;                            // [n]
    double* Ld = lam + MF_NMAX;                            // scratch [MF_NMAX]: 1 / L_kk of the frame block; later the ranks (as doubles)
    double* rc = Ld + MF_NMAX;                             // two buffers of: per pair c, s, t, apq, app, aqq
    int* pq = (int*)(rc + 12 * MF_NP);                     // two buffers of: per pair p, q
    int* ictl = (int*)(rc + 14 * MF_NP);                   // [0] rotations of the sweep, [1] bad flag, [2] total rotations, [3] irregular, [4] eigenvalues cut
    int* blk = (int*)(rc + 14 * MF_NP + 16);               // block t of the A pass: i | j << 8 (i >= j), filled once per window
    if (tid < 16) ictl[tid] = 0;
    if (N < 1 || N > MF_NMAX || n < 1 || n > MF_NKEEP || md < 0 || md > MF_MD || md + n != N) {      // (the host sends such a window down its own path; never reached through the ABI)
        if (tid == 0) { out[MF_OUT_S] = (double)MF_IRREGULAR; out[MF_OUT_S + 1] = 0; out[MF_OUT_S + 2] = 0; out[MF_OUT_S + 3] = 0; }
        return;
    }
    __syncthreads();
    // ---- the system: rows / columns [0, md) the dropped frame block, [md, N) the kept ones
    int bad = 0;
    if (mode == 0) {
        const double* S = lin_all + (size_t)lin_stride * b; const double* g = S + (size_t)tri_n * (tri_n + 1) / 2;
        const int* map = desc + 4;
        for (int e = tid; e < N * N; e += MF_NT) {
            const int i = e / N, j = e - i * N, ia = map[i], ib = map[j];
            const int hi = ia >= ib ? ia : ib, lo = ia >= ib ? ib : ia;
            const double v = S[(size_t)hi * (hi + 1) / 2 + lo];
            A[i * MF_LDA + j] = v; bad |= !isfinite(v);
        }
        if (tid < N) { const double v = g[map[tid]]; bv[tid] = v; bad |= !isfinite(v); }
    } else {
        const double* D = in_all + (size_t)MF_IN * b;
        for (int e = tid; e < N * N; e += MF_NT) { const int i = e / N, j = e - i * N; const double v = D[e]; A[i * MF_LDA + j] = v; bad |= !isfinite(v); }
        if (tid < N) { const double v = D[(size_t)N * N + tid]; bv[tid] = v; bad |= !isfinite(v); }
    }
    if (bad) ictl[1] = 1;
    __syncthreads();
    if (ictl[1]) { if (tid == 0) { out[MF_OUT_S] = (double)MF_NONFINITE; out[MF_OUT_S + 1] = 0; out[MF_OUT_S + 2] = 0; out[MF_OUT_S + 3] = 0; } return; }
    // ---- frame block: S_dd = sym(A_dd) = L L^T on the first wave (lane i owns row i), a pivot at or under eps sends the window to the host (which forms the pseudo-inverse
    //      from the block's eigen-decomposition, uvs_marg.h: marg_solve_small)
    if (md > 0) {
        if (tid < 64) {
            const int i = tid;
            if (i < md) for (int j = 0; j <= i; ++j) { const double v = 0.5 * (A[i * MF_LDA + j] + A[j * MF_LDA + i]); A[i * MF_LDA + j] = v; }
            __builtin_amdgcn_wave_barrier();
            for (int k = 0; k < md; ++k) {
                // row k is final once columns < k have been applied: lane k finishes its diagonal, the others divide
                double t = A[k * MF_LDA + k];
                for (int q = 0; q < k; ++q) { const double l = A[k * MF_LDA + q]; t -= l * l; }
                if (!(t > eps)) { if (i == 0) ictl[3] = 1; t = 1.0; }
                const double lkk = sqrt(t);
                if (i == k) { Ld[k] = 1.0 / lkk; }
                if (i > k && i < md) {
                    double v = A[i * MF_LDA + k];
                    for (int q = 0; q < k; ++q) v -= A[i * MF_LDA + q] * A[k * MF_LDA + q];
                    A[i * MF_LDA + k] = v / lkk;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();
        if (ictl[3]) { if (tid == 0) { out[MF_OUT_S] = (double)MF_IRREGULAR; out[MF_OUT_S + 1] = 0; out[MF_OUT_S + 2] = 0; out[MF_OUT_S + 3] = 0; } return; }
        // X = S_dd^-1 [A_dr | b_d]: one thread per right-hand side (forward, then backward substitution with L; the diagonal of L is kept as 1 / L_kk)
        if (tid <= n) {
            double x[MF_MD];
            for (int k = 0; k < md; ++k) {
                double t = tid < n ? A[k * MF_LDA + md + tid] : bv[k];
                for (int q = 0; q < k; ++q) t -= A[k * MF_LDA + q] * x[q];
                x[k] = t * Ld[k];
            }
            for (int k = md - 1; k >= 0; --k) {
                double t = x[k];
                for (int q = k + 1; q < md; ++q) t -= A[q * MF_LDA + k] * x[q];
                x[k] = t * Ld[k];
            }
            for (int k = 0; k < md; ++k) X[k * (n + 1) + tid] = x[k];
        }
        __syncthreads();
    }
    // ---- Schur complement onto the kept blocks, lower triangle computed and mirrored (the reference's eigen-solver reads the lower triangle): into Vt as scratch, then back to A
    for (int e = tid; e < n * n; e += MF_NT) {
        const int i = e / n, j = e - i * n;
        if (j > i) continue;
        const double* ad = A + (md + i) * MF_LDA;
        double t = ad[md + j];
        for (int k = 0; k < md; ++k) t -= ad[k] * X[k * (n + 1) + j];
        Vt[i * MF_LDV + j] = t;
    }
    if (tid < n) {
        const double* ad = A + (md + tid) * MF_LDA;
        double t = bv[md + tid];
        for (int k = 0; k < md; ++k) t -= ad[k] * X[k * (n + 1) + n];
        br[tid] = t;
    }
    __syncthreads();
    const int ne = (n + 1) & ~1;      // even: index n (when n is odd) is a dummy that never rotates
    for (int e = tid; e < ne * ne; e += MF_NT) {
        const int i = e / ne, j = e - i * ne;
        A[i * MF_LDA + j] = (i < n && j < n) ? (j <= i ? Vt[i * MF_LDV + j] : Vt[j * MF_LDV + i]) : 0.0;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += MF_NT) { const int i = e / n, j = e - i * n; Vt[i * MF_LDV + j] = (i == j) ? 1.0 : 0.0; }
    __syncthreads();
    // ---- parallel cyclic Jacobi.  A step (n / 2 disjoint rotations J = prod J_i) is TWO workgroup barriers:
    //   [A pass]  A <- J^T A J in ONE sweep over the 2 x 2 blocks of pairs: block (i, j) = rows {p_i, q_i} x columns {p_j, q_j} becomes R_i^T M R_j -- one load and one store per
    //             entry of A (rows-then-columns would be two each, and the kernel is bound by LDS traffic); blocks i > j are computed and mirrored, the diagonal blocks are set in
    //             closed form.  The n / 2 (n / 2 + 1) / 2 blocks are dealt evenly over the compute threads.
    //   [V pass]  rows p_i, q_i of V^T rotate (sixteen lanes per pair), WHILE the control wave (threads >= MF_CT) computes the NEXT step's rotation parameters from the A the A pass
    //             just finished (the V pass does not touch A): the serial FP64 chain of a rotation's c, s hides behind the V traffic.  Parameters are double-buffered.
    const int np = ne / 2;
    int sweeps = 0; bool converged = false;
    long long cyc[3] = {0, 0, 0};      // shader-clock cycles of thread 0 (status words 4 .. 6): first parameters of a sweep, A pass, V pass (+ next parameters on the control wave)
    const bool ctl_wave = tid >= MF_CT;
    const int nblk = np * (np + 1) / 2;
    for (int t = tid; t < nblk; t += MF_NT) { int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5); while ((i + 1) * (i + 2) / 2 <= t) ++i; while (i * (i + 1) / 2 > t) --i; blk[t] = i | ((t - i * (i + 1) / 2) << 8); }
    const int gi = tid >> 4, l16 = tid & 15;      // (compute threads: row pair gi, lane l16 of its sixteen)
    // rotation parameters of `step` into buffer `buf` (control wave; pair i of a step by the circle method: i = 0: (ne - 1, step); i > 0: ((step + i) mod (ne - 1), (step - i) mod (ne - 1)))
    auto parameters = [&](int step, int buf) {
        const int i = tid - MF_CT;
        bool rot = false;
        if (i < np) {
            int p = i == 0 ? ne - 1 : (step + i) % (ne - 1), q = i == 0 ? step : (step + ne - 1 - i) % (ne - 1);
            if (p > q) { const int t_ = p; p = q; q = t_; }
            double c = 1.0, sn = 0.0, t = 0.0, apq = 0.0, app = 0.0, aqq = 0.0;
            if (q < n) {
                apq = A[p * MF_LDA + q]; app = A[p * MF_LDA + p]; aqq = A[q * MF_LDA + q];
                // (a pair INSIDE the subspace the eps cut discards -- both diagonal entries and the coupling a thousand times under eps, so both eigenvalues of the 2 x 2 block
                // are -- is left alone: its rotation would only mix two rows of V^T that leave as zero rows of J0; without this rule the relative criterion keeps such pairs,
                // whose entries are round-off of a matrix of norm 1e8..1e14, rotating for another two sweeps)
                const bool in_cut = fabs(app) <= 1e-3 * eps && fabs(aqq) <= 1e-3 * eps && fabs(apq) <= 1e-3 * eps;
                // (the stopping rule |a_pq| <= 1.1e-16 sqrt(|a_pp a_qq|) compared in squares; the two divisions and two square roots of the rotation from the hardware
                // reciprocal / reciprocal-square-root seeds with two Newton steps each: c and s are orthonormal to 1e-16 either way, which is all a Jacobi rotation needs)
                if (apq != 0.0 && !in_cut && !(apq * apq <= 1.21e-32 * fabs(app * aqq))) {
                    const double tau = (aqq - app) * mf_rcp(2.0 * apq);
                    const double w1 = fma(tau, tau, 1.0);
                    t = (tau >= 0.0 ? 1.0 : -1.0) * mf_rcp(fabs(tau) + w1 * mf_rsq(w1));
                    c = mf_rsq(fma(t, t, 1.0)); sn = t * c;
                    rot = true;
                } else apq = 0.0;      // (no rotation: the pair's entries stay as they are, exactly)
            }
            double* r = rc + (size_t)buf * 6 * MF_NP + 6 * i; r[0] = c; r[1] = sn; r[2] = t; r[3] = apq; r[4] = app; r[5] = aqq;
            pq[buf * 2 * MF_NP + 2 * i] = p; pq[buf * 2 * MF_NP + 2 * i + 1] = q;
        }
        const unsigned long long any = __ballot(rot);
        if (tid == MF_CT) { const int cnt = __popcll(any); ictl[6 + buf] = cnt; ictl[0] += cnt; }
    };
    for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
        const long long ts0 = clock64();
        if (tid == MF_CT) ictl[0] = 0;
        if (ctl_wave) parameters(0, 0);
        __syncthreads();
        cyc[0] += clock64() - ts0;
        for (int step = 0; step < ne - 1; ++step) {
            const int buf = step & 1;
            const double* rcb = rc + (size_t)buf * 6 * MF_NP; const int* pqb = pq + buf * 2 * MF_NP;
            const bool active = ictl[6 + buf] != 0;      // some pair of this step rotates (the same word in every thread: written before the last barrier)
            const long long tc0 = clock64();
            if (active && !ctl_wave) {
                // the blocks dealt evenly: thread t takes blocks t and t + MF_CT of the nblk <= 820 (a row pair per sixteen lanes left the lanes of the short rows idle)
#pragma unroll
                for (int u = 0; u < (MF_NBLK + MF_CT - 1) / MF_CT; ++u) {
                    const int t = tid + MF_CT * u;
                    if (t >= nblk) continue;
                    const int ij = blk[t], i = ij & 255, j = ij >> 8;
                    const double ci = rcb[6 * i], si = rcb[6 * i + 1], roti = rcb[6 * i + 3], cj = rcb[6 * j], sj = rcb[6 * j + 1], rotj = rcb[6 * j + 3];
                    if (roti == 0.0 && rotj == 0.0) continue;
                    const int pi = pqb[2 * i], qi = pqb[2 * i + 1];
                    if (i == j) {      // the pair's own 2 x 2 block in closed form
                        const double tt = rcb[6 * i + 2];
                        A[pi * MF_LDA + pi] = rcb[6 * i + 4] - tt * roti; A[qi * MF_LDA + qi] = rcb[6 * i + 5] + tt * roti; A[pi * MF_LDA + qi] = 0.0; A[qi * MF_LDA + pi] = 0.0;
                        continue;
                    }
                    const int pj = pqb[2 * j], qj = pqb[2 * j + 1];
                    const double m00 = A[pi * MF_LDA + pj], m01 = A[pi * MF_LDA + qj], m10 = A[qi * MF_LDA + pj], m11 = A[qi * MF_LDA + qj];
                    const double r0 = ci * m00 - si * m10, r1 = ci * m01 - si * m11, r2 = si * m00 + ci * m10, r3 = si * m01 + ci * m11;      // rows rotated by pair i
                    const double n00 = cj * r0 - sj * r1, n01 = sj * r0 + cj * r1, n10 = cj * r2 - sj * r3, n11 = sj * r2 + cj * r3;          // columns rotated by pair j
                    A[pi * MF_LDA + pj] = n00; A[pi * MF_LDA + qj] = n01; A[qi * MF_LDA + pj] = n10; A[qi * MF_LDA + qj] = n11;
                    A[pj * MF_LDA + pi] = n00; A[qj * MF_LDA + pi] = n01; A[pj * MF_LDA + qi] = n10; A[qj * MF_LDA + qi] = n11;
                }
            }
            __syncthreads();
            const long long tc1 = clock64();
            if (ctl_wave) { if (step + 1 < ne - 1) parameters(step + 1, buf ^ 1); }
            else if (active && gi < np && rcb[6 * gi + 3] != 0.0) {
                const double c = rcb[6 * gi], sn = rcb[6 * gi + 1];
                double* Vp = Vt + pqb[2 * gi] * MF_LDV; double* Vq = Vt + pqb[2 * gi + 1] * MF_LDV;
                double va[MF_RS], vb[MF_RS];
#pragma unroll
                for (int u = 0; u < MF_RS; ++u) { const int k = l16 + 16 * u; const bool in = k < n; va[u] = in ? Vp[k] : 0.0; vb[u] = in ? Vq[k] : 0.0; }
#pragma unroll
                for (int u = 0; u < MF_RS; ++u) { const int k = l16 + 16 * u; if (k < n) { Vp[k] = c * va[u] - sn * vb[u]; Vq[k] = sn * va[u] + c * vb[u]; } }
            }
            __syncthreads();
            { const long long tc2 = clock64(); cyc[1] += tc1 - tc0; cyc[2] += tc2 - tc1; }
        }
        ++sweeps;
        converged = ictl[0] == 0;
        if (tid == MF_CT) ictl[2] += ictl[0];
        __syncthreads();
    }
    // ---- eigenvalues in ascending order (ties by index, like the host's stable sort), rows of J0 / r0 in that order
    if (tid < n) lam[tid] = A[tid * MF_LDA + tid];
    __syncthreads();
    if (tid < n) {
        const double l = lam[tid]; int rk = 0;
        for (int j = 0; j < n; ++j) { const double lj = lam[j]; rk += (lj < l || (lj == l && j < tid)) ? 1 : 0; }
        Ld[tid] = (double)rk;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += MF_NT) {
        const int k = e / n, j = e - k * n;
        const double l = lam[k]; const bool on = l > eps;
        out[MF_OUT_J + (size_t)((int)Ld[k]) * n + j] = on ? sqrt(l) * Vt[k * MF_LDV + j] : 0.0;
    }
    if (tid < n) {
        const double l = lam[tid]; const bool on = l > eps;
        double vb = 0.0;
        for (int j = 0; j < n; ++j) vb += Vt[tid * MF_LDV + j] * br[j];
        out[MF_OUT_R + (int)Ld[tid]] = on ? sqrt(1.0 / l) * vb : 0.0;
        if (!on) atomicAdd(&ictl[4], 1);
    }
    __syncthreads();
    if (tid == 0) { out[MF_OUT_S] = (double)(converged ? MF_OK : MF_UNCONVERGED); out[MF_OUT_S + 1] = (double)sweeps; out[MF_OUT_S + 2] = (double)ictl[2]; out[MF_OUT_S + 3] = (double)ictl[4];
                    out[MF_OUT_S + 4] = (double)cyc[0]; out[MF_OUT_S + 5] = (double)cyc[1]; out[MF_OUT_S + 6] = (double)cyc[2]; }
}

}  // namespace uvsmarg
