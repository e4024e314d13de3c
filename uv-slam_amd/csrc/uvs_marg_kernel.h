// uvs_marg_kernel.h -- the BACK HALF of a marginalization on the device, batched over windows (round 6; SURVEY.md section 7: k_marg_assemble / k_marg_schur / k_sym_eig).
//
// Replaces, for a batch of windows, what uvs_marg.h: marg_finish does on one host core per window (marginalization_factor.cpp:263-291): elimination of the dropped FRAME block
// (Pose[0] + SpeedBias[0] for MARGIN_OLD, Pose[WINDOW_SIZE - 1] for MARGIN_SECOND_NEW; <= 15 dofs) from the assembled, landmark-eliminated system, the Schur complement onto the
// kept blocks, the symmetric eigen-decomposition of the n x n result (n <= 76 in the reference) and the factor J0 = sqrt(S) V^T, r0 = sqrt(S^-1) V^T b with eigenvalues <= 1e-8
// cut (marginalization_factor.h:70).  One workgroup per window, everything in LDS.
//
// The eigen-decomposition is a PARALLEL cyclic Jacobi (round-robin ordering: n / 2 disjoint rotations per step, n - 1 steps per sweep) with the same rotation formulas and the
// same RELATIVE stopping rule |a_pq| <= 1.1e-16 sqrt(|a_pp a_qq|) as the host's host_sym_eig_jacobi (uvs_marg.h:30-58): the matrices are graded over twenty orders of magnitude
// and Jacobi resolves the small eigenvalues relative to their own scale.  For ONE window this is no faster than the host's tridiagonal QL (~0.3 ms against ~0.2 ms): a rotation step
// is a workgroup barrier away from the next and there are ~700 of them.  For a BATCH it is what makes the marginalization scale with the solve: 256 windows take one launch instead
// of 256 x 0.2 ms on a host core (uvs_marginalize_batch).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/uvs_solver.h"

namespace uvsmarg {

static constexpr int MF_NT = 640;           // ten waves = sixteen lanes for each of the <= 40 rotations of a step (MF_NKEEP / 2); a step is three barriers around them
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
static_assert(16 * (MF_NKEEP / 2) <= MF_NT, "sixteen lanes per pair");
static constexpr size_t MF_LDS_DOUBLES = (size_t)MF_NMAX * MF_LDA + (size_t)MF_NKEEP * MF_LDV + (size_t)MF_MD * (MF_NKEEP + 2) + 4 * MF_NMAX + 7 * MF_NP + 16;      // (the last 16: control words, 8 doubles used)
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
    double* rc = Ld + MF_NMAX;                             // per pair: c, s, t, apq, app, aqq
    int* pq = (int*)(rc + 6 * MF_NP);                      // per pair: p, q
    int* ictl = (int*)(rc + 7 * MF_NP);                    // [0] rotations of the sweep, [1] bad flag, [2] total rotations, [3] irregular, [4] eigenvalues cut
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
    // ---- parallel cyclic Jacobi
    const int np = ne / 2;
    int sweeps = 0; bool converged = false;
    long long cyc[3] = {0, 0, 0};      // shader-clock cycles of thread 0 in the three parts of the steps (status words 4 .. 6: rotation parameters, row pass, column pass)
    for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
        if (tid == 0) { ictl[0] = 0; ictl[6] = 0; }
        __syncthreads();
        for (int step = 0; step < ne - 1; ++step) {
            const long long tc0 = clock64();
            // pair i of this step (circle method): i = 0: (ne - 1, step); i > 0: ((step + i) mod (ne - 1), (step + ne - 1 - i) mod (ne - 1))
            if (tid == 64) ictl[6 + ((step + 1) & 1)] = 0;      // the NEXT step's "some pair rotates" flag (nobody reads it before that step's first barrier; this step's was cleared a step ago)
            if (tid < np) {
                int p = tid == 0 ? ne - 1 : (step + tid) % (ne - 1), q = tid == 0 ? step : (step + ne - 1 - tid) % (ne - 1);
                if (p > q) { const int t_ = p; p = q; q = t_; }
                double c = 1.0, s = 0.0, t = 0.0, apq = 0.0, app = 0.0, aqq = 0.0;
                if (q < n) {
                    apq = A[p * MF_LDA + q]; app = A[p * MF_LDA + p]; aqq = A[q * MF_LDA + q];
                    // (a pair INSIDE the subspace the eps cut discards -- both diagonal entries and the coupling a thousand times under eps, so both eigenvalues of the 2 x 2 block
                    // are -- is left alone: its rotation would only mix two rows of V^T that leave as zero rows of J0; without this rule the relative criterion keeps such pairs,
                    // whose entries are round-off of a matrix of norm 1e8..1e14, rotating for another 6 - 8 sweeps)
                    const bool in_cut = fabs(app) <= 1e-3 * eps && fabs(aqq) <= 1e-3 * eps && fabs(apq) <= 1e-3 * eps;
                    // (this lane's arithmetic is the serial part of a step: the stopping rule is compared in squares -- no square root --, and the two divisions and two
                    // square roots of the rotation use the hardware reciprocal / reciprocal-square-root seeds with two Newton steps each instead of the IEEE sequences:
                    // c and s are orthonormal to 1e-16 either way, which is all a Jacobi rotation needs)
                    if (apq != 0.0 && !in_cut && !(apq * apq <= 1.21e-32 * fabs(app * aqq))) {
                        const double tau = (aqq - app) * mf_rcp(2.0 * apq);
                        const double w1 = fma(tau, tau, 1.0);
                        t = (tau >= 0.0 ? 1.0 : -1.0) * mf_rcp(fabs(tau) + w1 * mf_rsq(w1));
                        c = mf_rsq(fma(t, t, 1.0)); s = t * c;
                        atomicAdd(&ictl[0], 1); ictl[6 + (step & 1)] = 1;
                    } else apq = 0.0;      // (no rotation: the pair's 2 x 2 block stays as it is, exactly)
                }
                double* r = rc + 6 * tid; r[0] = c; r[1] = s; r[2] = t; r[3] = apq; r[4] = app; r[5] = aqq;
                pq[2 * tid] = p; pq[2 * tid + 1] = q;
            }
            __syncthreads();
            const long long tc1 = clock64();
            if (ictl[6 + (step & 1)] == 0) { cyc[0] += tc1 - tc0; continue; }      // nothing rotates in this step (the late sweeps): no passes, no barriers -- the same decision in every thread
            // rows p, q of A and of V^T, then columns p, q of A: SIXTEEN lanes per pair (all n / 2 <= 40 pairs of the step at once on 640 threads; a row is five
            // lane-strides long), every load of a lane in flight before its first store.  The rotation parameters are read once per lane (a broadcast within the 16 lanes).
            const int gi = tid >> 4, l16 = tid & 15;
            const bool mine = gi < np && rc[6 * (gi < np ? gi : 0) + 3] != 0.0;
            double c = 1.0, sn = 0.0; int p = 0, q = 0;
            if (mine) { c = rc[6 * gi]; sn = rc[6 * gi + 1]; p = pq[2 * gi]; q = pq[2 * gi + 1]; }
            if (mine) {
                double* Ap = A + p * MF_LDA; double* Aq = A + q * MF_LDA; double* Vp = Vt + p * MF_LDV; double* Vq = Vt + q * MF_LDV;
                double a[MF_RS], bq[MF_RS], va[MF_RS], vb[MF_RS];
#pragma unroll
                for (int u = 0; u < MF_RS; ++u) { const int k = l16 + 16 * u; const bool in = k < n; a[u] = in ? Ap[k] : 0.0; bq[u] = in ? Aq[k] : 0.0; va[u] = in ? Vp[k] : 0.0; vb[u] = in ? Vq[k] : 0.0; }
#pragma unroll
                for (int u = 0; u < MF_RS; ++u) {
                    const int k = l16 + 16 * u;
                    if (k < n) { Ap[k] = c * a[u] - sn * bq[u]; Aq[k] = sn * a[u] + c * bq[u]; Vp[k] = c * va[u] - sn * vb[u]; Vq[k] = sn * va[u] + c * vb[u]; }
                }
            }
            __syncthreads();
            const long long tc2 = clock64();
            if (mine) {      // columns p, q (rows k outside the pair), and the pair's 2 x 2 block in closed form
                const double* r = rc + 6 * gi;
                double a[MF_RS], bq[MF_RS];
#pragma unroll
                for (int u = 0; u < MF_RS; ++u) { const int k = l16 + 16 * u; const bool in = k < n; a[u] = in ? A[k * MF_LDA + p] : 0.0; bq[u] = in ? A[k * MF_LDA + q] : 0.0; }
#pragma unroll
                for (int u = 0; u < MF_RS; ++u) {
                    const int k = l16 + 16 * u;
                    if (k >= n) continue;
                    double* Mk = A + k * MF_LDA;
                    if (k == p) { Mk[p] = r[4] - r[2] * r[3]; Mk[q] = 0.0; }
                    else if (k == q) { Mk[q] = r[5] + r[2] * r[3]; Mk[p] = 0.0; }
                    else { Mk[p] = c * a[u] - sn * bq[u]; Mk[q] = sn * a[u] + c * bq[u]; }
                }
            }
            __syncthreads();
            { const long long tc3 = clock64(); cyc[0] += tc1 - tc0; cyc[1] += tc2 - tc1; cyc[2] += tc3 - tc2; }
        }
        ++sweeps;
        if (tid == 0) ictl[2] += ictl[0];
        converged = ictl[0] == 0;
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
