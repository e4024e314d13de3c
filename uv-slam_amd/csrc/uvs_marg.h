// uvs_marg.h -- marginalization prior construction.
//
// Replaces MarginalizationInfo::{addResidualBlockInfo, preMarginalize, marginalize, getParameterBlocks}
// (marginalization_factor.cpp:89-319) and its orchestration in Estimator::optimization()
// (estimator.cpp:1002-1228).
//
// Split (round 1): the data-parallel part -- evaluating every to-be-marginalised residual block with
// its loss correction at the post-solve state (ResidualBlockInfo::Evaluate, :3-69) -- runs on the GPU
// through k_evaluate; the dense (m+n)^2 assembly, the Schur complement and the n x n factorisation run on the
// host in this file, as they do in the reference (4 pthreads + Eigen).  The pseudo-inverse of A_mm is applied by
// block elimination (see run_marginalize) and the factorisation J0 = sqrt(S) V^T by Householder + implicit QL;
// UVS_MARG_PROFILE=1 prints the stage times.  Since then: round 3 moved the assembly + landmark elimination of MARGIN_OLD to the
// device (k_marg_linearize: the host path below is what a window falls back to), round 6 made MARGIN_SECOND_NEW host-only (no
// device round trip: host_prior_residual) and added the BATCHED form whose back half runs on the device as well
// (uvs_marg_kernel.h: k_marg_finish; uvs_marginalize_batch in uvs_solver.hip uses marg_assemble_host / marg_fill_blocks of this file).
// For ONE window the host finish below (marg_finish: Cholesky of the frame block + Householder / QL) stays the faster one.
//
// Block order is deterministic (the reference's depends on pointer hashes, Appendix D6):
// dropped = {Pose, SpeedBias, point landmarks, line landmarks}, kept = {Pose asc., SpeedBias asc., Ex_Pose}.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>
#include "uvs_eval_kernel.h"

namespace uvsdev {

// Cyclic Jacobi with a RELATIVE stopping rule |a_pq| <= eps sqrt(a_pp a_qq): the matrices of the marginalization are graded over ten
// orders of magnitude (pose information 1e7..1e14, inverse depths 1e0..1e2, near-null gauge directions of the prior 1e-6) and both the
// inverted eigenvalues of A_mm and the eps = 1e-8 cut of the n x n factor need RELATIVE accuracy on the small ones, which Jacobi
// delivers on the graded matrix itself and a tridiagonal QL does not (absolute error ~1e-16 ||A||).  Row-oriented: a rotation
// updates rows p, q of M and of V^T (contiguous), mirrors them into the columns, and sets the 2x2 pivot block analytically.
static void host_sym_eig_jacobi(int n, std::vector<double>& M, std::vector<double>& V, std::vector<double>& lam) {
    std::vector<double> Vt((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) Vt[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        int rotations = 0;
        for (int p = 0; p + 1 < n; ++p) {
            double* Mp = &M[(size_t)p * n]; double* Vp = &Vt[(size_t)p * n];
            for (int q = p + 1; q < n; ++q) {
                double* Mq = &M[(size_t)q * n];
                const double apq = Mp[q], app = Mp[p], aqq = Mq[q];
                if (apq == 0.0 || std::fabs(apq) <= 1.1e-16 * std::sqrt(std::fabs(app * aqq))) continue;
                ++rotations;
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
                for (int k = 0; k < n; ++k) { const double a = Mp[k], b = Mq[k]; Mp[k] = cs * a - sn * b; Mq[k] = sn * a + cs * b; }
                for (int k = 0; k < n; ++k) { M[(size_t)k * n + p] = Mp[k]; M[(size_t)k * n + q] = Mq[k]; }
                Mp[p] = app - t * apq; Mq[q] = aqq + t * apq; Mp[q] = 0.0; Mq[p] = 0.0;
                double* Vq = &Vt[(size_t)q * n];
                for (int k = 0; k < n; ++k) { const double a = Vp[k], b = Vq[k]; Vp[k] = cs * a - sn * b; Vq[k] = sn * a + cs * b; }
            }
        }
        if (rotations == 0) break;
    }
    lam.resize(n); V.resize((size_t)n * n);
    for (int i = 0; i < n; ++i) { lam[i] = M[(size_t)i * n + i]; for (int k = 0; k < n; ++k) V[(size_t)k * n + i] = Vt[(size_t)i * n + k]; }
}

// Eigen-decomposition of a dense symmetric matrix (row-major n x n; V: eigenvectors in columns, unsorted): Householder reduction to
// tridiagonal form followed by the implicit-shift QL iteration -- the textbook tred2 / tql2 pair, i.e. the same family of algorithm
// as Eigen's SelfAdjointEigenSolver that the reference calls (marginalization_factor.cpp:263,278), ~(4/3) n^3 + O(n^2) per sweep flops
// instead of the ~10 sweeps x 6 n^3 of a cyclic Jacobi (3.2 ms -> 0.25 ms for n = 69 on the box's host core).
// The working matrix is kept TRANSPOSED (T[j][i] = V(i, j)): every O(n^3) loop of the pair runs down a column of V, which is then a
// contiguous row of T -- unit stride, vectorisable (an AVX2 clone is selected at load time; contraction off, so both clones produce the
// same bits).  The n = 69 factorisation is the largest single piece of a marginalization.
#if !defined(__HIP_DEVICE_COMPILE__)
#define UVS_HOST_SIMD __attribute__((target_clones("arch=haswell", "default")))
#else
#define UVS_HOST_SIMD
#endif
#pragma clang fp contract(off)
UVS_HOST_SIMD static void eig_axpy2(int n, double* __restrict col, const double* __restrict e, const double* __restrict d, double f, double g) {      // col[k] -= f e[k] + g d[k]
    for (int k = 0; k < n; ++k) col[k] -= (f * e[k] + g * d[k]);
}
UVS_HOST_SIMD static double eig_dot_axpy(int n, const double* __restrict col, const double* __restrict d, double* __restrict e, double f, double g) {   // g += col . d ; e += f col
    {
#pragma clang fp reassociate(on)      // the dot product may be summed in vector lanes (a different rounding of a Householder inner product, nothing else)
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += col[k] * d[k];
        g += acc;
    }
    for (int k = 0; k < n; ++k) e[k] += col[k] * f;
    return g;
}
UVS_HOST_SIMD static double eig_dot(int n, const double* __restrict a, const double* __restrict b) {
#pragma clang fp reassociate(on)
    double acc = 0.0;
    for (int k = 0; k < n; ++k) acc += a[k] * b[k];
    return acc;
}
UVS_HOST_SIMD static void eig_axpy(int n, double* __restrict col, const double* __restrict d, double g) { for (int k = 0; k < n; ++k) col[k] -= g * d[k]; }
UVS_HOST_SIMD static void eig_rotate(int n, double* __restrict a, double* __restrict b, double c, double s) {      // a = column i, b = column i + 1 of V
    for (int k = 0; k < n; ++k) { const double h = b[k]; b[k] = s * a[k] + c * h; a[k] = c * a[k] - s * h; }
}
static void host_sym_eig(int n, std::vector<double>& M, std::vector<double>& V, std::vector<double>& lam) {
    lam.assign(n, 0.0);
    if (n == 0) { V.clear(); return; }
    std::vector<double> T(M), e(n, 0.0);      // M is symmetric: its transpose is itself
    double* d = lam.data();
    auto at = [&](int i, int j) -> double& { return T[(size_t)j * n + i]; };      // V(i, j)
    auto colp = [&](int j) -> double* { return &T[(size_t)j * n]; };               // column j of V, contiguous
    // ---- Householder tridiagonalisation (row n-1 of V carries the current vector)
    for (int j = 0; j < n; ++j) d[j] = at(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = at(i - 1, j); at(i, j) = 0.0; at(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1], g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g; h -= f * g; d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j]; at(j, i) = f; g = e[j] + at(j, j) * f;
                e[j] = eig_dot_axpy(i - 1 - j, colp(j) + j + 1, d + j + 1, e.data() + j + 1, f, g);      // k = j + 1 .. i - 1
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                eig_axpy2(i - j, colp(j) + j, e.data() + j, d + j, f, g);                                  // k = j .. i - 1
                d[j] = at(i - 1, j); at(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    // ---- accumulate the reflectors
    for (int i = 0; i < n - 1; ++i) {
        at(n - 1, i) = at(i, i); at(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            const double* ci = colp(i + 1);
            for (int k = 0; k <= i; ++k) d[k] = ci[k] / h;
            for (int j = 0; j <= i; ++j) {
                const double g = eig_dot(i + 1, ci, colp(j));
                eig_axpy(i + 1, colp(j), d, g);
            }
        }
        for (int k = 0; k <= i; ++k) at(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) { d[j] = at(n - 1, j); at(n - 1, j) = 0.0; }
    at(n - 1, n - 1) = 1.0; e[0] = 0.0;
    // ---- implicit-shift QL on (d, e), rotations applied to V
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n - 1) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }      // e[n-1] = 0 ends the scan; the bound keeps a NaN matrix inside the arrays
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 120) break;      // never seen; the caller's eps cut tolerates an unconverged tiny eigenvalue
                double g = d[l], p = (d[l + 1] - g) / (2.0 * e[l]), r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r); d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i]; h = c * p; r = std::sqrt(p * p + e[i] * e[i]);      // (entries are <= 1e15 in magnitude: no overflow guard needed)
                    e[i + 1] = s * r; s = e[i] / r; c = p / r; p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    eig_rotate(n, colp(i), colp(i + 1), c, s);
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p; d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] += f; e[l] = 0.0;
    }
    V.assign((size_t)n * n, 0.0);      // back to the callers' convention: row-major V, eigenvectors in columns
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = T[(size_t)j * n + i];
}
UVS_HOST_SIMD static void row_axpy(int n, double* __restrict y, const double* __restrict x, double a) { for (int k = 0; k < n; ++k) y[k] += a * x[k]; }      // y[0..n) += a x
#pragma clang fp contract(on)

// X <- B^+ X for a small symmetric block B (sz <= 15) and sz x nr right-hand sides (row stride ldx).  Regular case: Cholesky solve
// (an eigen-based inverse of the frame block, whose eigenvalues span 1e5..1e14, costs four digits in J0^T r0); a pivot at or under
// eps means the reference's cut would bite, and only then the pseudo-inverse is formed from the block's eigen-decomposition.
static void marg_solve_small(int sz, const double* Bm, double* X, int nr, int ldx, double eps) {
        double Lc[225]; bool regular = true;
        for (int i = 0; i < sz && regular; ++i) for (int j = 0; j <= i; ++j) {
            double t = Bm[i * sz + j];
            for (int k = 0; k < j; ++k) t -= Lc[i * sz + k] * Lc[j * sz + k];
            if (i == j) { if (!(t > eps)) { regular = false; break; } Lc[i * sz + i] = std::sqrt(t); } else Lc[i * sz + j] = t / Lc[j * sz + j];
        }
        if (regular) {
            for (int c2 = 0; c2 < nr; ++c2) {
                for (int i = 0; i < sz; ++i) { double t = X[(size_t)i * ldx + c2]; for (int k = 0; k < i; ++k) t -= Lc[i * sz + k] * X[(size_t)k * ldx + c2]; X[(size_t)i * ldx + c2] = t / Lc[i * sz + i]; }
                for (int i = sz - 1; i >= 0; --i) { double t = X[(size_t)i * ldx + c2]; for (int k = i + 1; k < sz; ++k) t -= Lc[k * sz + i] * X[(size_t)k * ldx + c2]; X[(size_t)i * ldx + c2] = t / Lc[i * sz + i]; }
            }
            return;
        }
        std::vector<double> M2(Bm, Bm + (size_t)sz * sz), Vs, ls, Binv((size_t)sz * sz, 0.0), col(sz);
        host_sym_eig_jacobi(sz, M2, Vs, ls);
        for (int k = 0; k < sz; ++k) { if (!(ls[k] > eps)) continue; const double il = 1.0 / ls[k]; for (int i = 0; i < sz; ++i) for (int j = 0; j < sz; ++j) Binv[(size_t)i * sz + j] += Vs[(size_t)i * sz + k] * il * Vs[(size_t)j * sz + k]; }
        for (int c2 = 0; c2 < nr; ++c2) {
            for (int i = 0; i < sz; ++i) { double t = 0.0; for (int k = 0; k < sz; ++k) t += Binv[(size_t)i * sz + k] * X[(size_t)k * ldx + c2]; col[i] = t; }
            for (int i = 0; i < sz; ++i) X[(size_t)i * ldx + c2] = col[i];
        }
    }

// The block table of a new prior: the kept blocks in id order with their local sizes, column offsets and the linearization point = the window's current values, frames shifted as the
// window will be (addr_shift, estimator.cpp:1139-1152 for MARGIN_OLD / :1196-1219 for MARGIN_SECOND_NEW).  out->n must be set; m = first kept row of the ordering `pos`.
static void marg_fill_blocks(uvs_prior* out, const std::vector<int>& pos, const std::vector<int>& keep_ids, int m, const uvs_window* w, int flag) {
    const int NFR = UVS_NF;
    auto gsize = [&](int id) { return id < NFR ? 7 : id < 2 * NFR ? 9 : id == 22 ? 7 : 1; };
    out->n_blocks = (int)keep_ids.size();
    int xo = 0;
    for (int b = 0; b < out->n_blocks; ++b) {
        const int id = keep_ids[b];
        int kind, frame = 0; const double* data;
        if (id < NFR) { kind = UVS_BLOCK_POSE; frame = id; data = w->pose[frame]; }
        else if (id < 2 * NFR) { kind = UVS_BLOCK_SPEEDBIAS; frame = id - NFR; data = w->speedbias[frame]; }
        else if (id == 23) { kind = UVS_BLOCK_TD; data = &w->td; }
        else { kind = UVS_BLOCK_EX_POSE; data = w->ex_pose; }
        int nf = frame;
        if (kind == UVS_BLOCK_POSE || kind == UVS_BLOCK_SPEEDBIAS) nf = (flag == 0) ? frame - 1 : (frame == UVS_WINDOW_SIZE ? frame - 1 : frame);
        out->block_kind[b] = kind; out->block_frame[b] = nf; out->block_size[b] = gsize(id); out->block_idx[b] = pos[id] - m; out->x0_off[b] = xo;
        for (int q = 0; q < gsize(id); ++q) out->x0[xo + q] = data[q];
        xo += gsize(id);
    }
}

// The tail of a marginalization, shared by the host path (A assembled and its landmark blocks eliminated on the host) and the device path (A = the reduced
// frame system a linearization kernel delivered): A is N x N with the dropped FRAME dofs in rows / columns [0, md) and the kept ones in [m, N); rows
// [md, m) (eliminated landmarks) are not read.  Eliminates the dropped frame block, factors the kept system J0 = sqrt(S) V^T, r0 = sqrt(S^-1) V^T b
// (marginalization_factor.cpp:263-291) and fills the block table with the shifted frames (estimator.cpp:1139-1152 / :1196-1219).
// Returns UVS_ERR_NUMERIC (and leaves *out untouched) when the system is not finite: a prior with NaNs would poison every later window.
static int marg_finish(int N, int m, int md, int n, std::vector<double>& A, std::vector<double>& bv, const std::vector<int>& pos, const std::vector<int>& keep_ids,
                        const uvs_window* w, int flag, uvs_prior* out, EvalScratch& sc, bool prof, const double* us_pre) {
    {
        double chk = 0.0;
        for (int i = 0; i < N; ++i) { if (i >= md && i < m) continue; chk += bv[i]; const double* ai = &A[(size_t)i * N]; for (int j = 0; j < md; ++j) chk += ai[j]; for (int j = m; j < N; ++j) chk += ai[j]; }
        if (!std::isfinite(chk)) return UVS_ERR_NUMERIC;
    }
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t3 = tnow();
    const double eps = 1e-8;                           // marginalization_factor.h:70
    const int NFR = UVS_NF;
    auto gsize = [&](int id) { return id < NFR ? 7 : id < 2 * NFR ? 9 : id == 22 ? 7 : id == 23 ? 1 : 1; };
    std::vector<double> Sd((size_t)md * md), Xd((size_t)md * (n + 1)), br(n);
    std::vector<double>& Ar = sc.work[6]; Ar.assign((size_t)n * n, 0.0);
    for (int i = 0; i < md; ++i) for (int j = 0; j < md; ++j) Sd[(size_t)i * md + j] = 0.5 * (A[(size_t)i * N + j] + A[(size_t)j * N + i]);
    for (int i = 0; i < md; ++i) { for (int j = 0; j < n; ++j) Xd[(size_t)i * (n + 1) + j] = A[(size_t)i * N + m + j]; Xd[(size_t)i * (n + 1) + n] = bv[i]; }
    marg_solve_small(md, Sd.data(), Xd.data(), n + 1, n + 1, eps);                   // X = S^+ [A_dr | b_d]
    for (int i = 0; i < n; ++i) {
        const double* ad = &A[(size_t)(m + i) * N];
        double sacc = bv[m + i]; for (int k = 0; k < md; ++k) sacc -= ad[k] * Xd[(size_t)k * (n + 1) + n]; br[i] = sacc;
        for (int j = 0; j < n; ++j) { double t = ad[m + j]; for (int k = 0; k < md; ++k) t -= ad[k] * Xd[(size_t)k * (n + 1) + j]; Ar[(size_t)i * n + j] = t; }
    }
    // ---- second eigen-decomposition -> J0 = sqrt(S) V^T, r0 = sqrt(S^-1) V^T b   (:278-291); lower triangle is read, like Eigen
    std::vector<double>&As = sc.work[7], &V2 = sc.work[8], &lam2 = sc.work[9];
    As.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) As[(size_t)i * n + j] = (j <= i) ? Ar[(size_t)i * n + j] : Ar[(size_t)j * n + i];
    auto t4 = tnow();
    host_sym_eig(n, As, V2, lam2);
    auto t5 = tnow();
    if (prof) {
        auto us = [](auto a, auto b) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() * 1e-3; };
        std::fprintf(stderr, "[uvs_marginalize] m %d n %d: evaluate / device linearization %.0f us, assemble %.0f us, landmark blocks %.0f us, frame block + schur %.0f us, eig(n) %.0f us\n", m, n, us_pre[0], us_pre[1], us_pre[2], us(t3, t4), us(t4, t5));
    }
    std::vector<int> ord(n); for (int i = 0; i < n; ++i) ord[i] = i;
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b2) { return lam2[a] < lam2[b2]; });
    std::memset(out, 0, sizeof(*out));
    out->n = n; out->n_blocks = (int)keep_ids.size();
    for (int row = 0; row < n; ++row) {
        const int k = ord[row];
        const bool on = lam2[k] > eps;
        const double ss = on ? std::sqrt(lam2[k]) : 0.0, si = on ? std::sqrt(1.0 / lam2[k]) : 0.0;
        double vb = 0.0;
        for (int j = 0; j < n; ++j) { out->linearized_jacobians[(size_t)row * n + j] = ss * V2[(size_t)j * n + k]; vb += V2[(size_t)j * n + k] * br[j]; }
        out->linearized_residuals[row] = si * vb;
    }
    marg_fill_blocks(out, pos, keep_ids, m, w, flag);
    return UVS_OK;
}

// The prior's residual r = r0 + J0 dx at the window's state, on the HOST (marginalization_factor.cpp:340-365: Euclidean difference per kept block, for a pose block
// [p - p0 ; 2 sign(w) vec(q0^-1 (x) q)]): what MARGIN_SECOND_NEW needs and all it needs -- that marginalization reads the old prior and nothing else (estimator.cpp:1159-1176),
// so it takes no device round trip at all (round 6; before, the whole window was packed, uploaded and evaluated on the device for this one vector).
static void host_prior_residual(const uvs_prior& p, const uvs_window* w, std::vector<double>& r) {
    const int n = p.n;
    double dx[UVS_MAX_PRIOR_DIM];
    for (int k = 0; k < n; ++k) dx[k] = 0.0;
    for (int b = 0; b < p.n_blocks; ++b) {
        const int kind = p.block_kind[b], size = p.block_size[b];
        const double* x = kind == UVS_BLOCK_POSE ? w->pose[p.block_frame[b]] : kind == UVS_BLOCK_SPEEDBIAS ? w->speedbias[p.block_frame[b]] : kind == UVS_BLOCK_TD ? &w->td : w->ex_pose;
        const double* x0 = p.x0 + p.x0_off[b];
        double* d = dx + p.block_idx[b];
        if (size != 7) { for (int k = 0; k < size; ++k) d[k] = x[k] - x0[k]; continue; }
        for (int k = 0; k < 3; ++k) d[k] = x[k] - x0[k];
        // e = q0^-1 (x) q, quaternions stored (x, y, z, w); Eigen's inverse() divides the conjugate by the squared norm
        const double ax = -x0[3], ay = -x0[4], az = -x0[5], aw = x0[6], nn = x0[3] * x0[3] + x0[4] * x0[4] + x0[5] * x0[5] + x0[6] * x0[6];
        const double bx = x[3], by = x[4], bz = x[5], bw = x[6];
        const double ex = (aw * bx + ax * bw + ay * bz - az * by) / nn, ey = (aw * by - ax * bz + ay * bw + az * bx) / nn, ez = (aw * bz + ax * by - ay * bx + az * bw) / nn;
        const double ew = (aw * bw - ax * bx - ay * by - az * bz) / nn;
        const double sg = ew >= 0.0 ? 2.0 : -2.0;
        d[3] = sg * ex; d[4] = sg * ey; d[5] = sg * ez;
    }
    r.assign(n, 0.0);
    for (int i = 0; i < n; ++i) {
        const double* Ji = &p.linearized_jacobians[(size_t)i * n];
        double acc = p.linearized_residuals[i];
        for (int k = 0; k < n; ++k) acc += Ji[k] * dx[k];
        r[i] = acc;
    }
}

struct MFactor { int rows; int nb; int id[5]; int sz[5]; const double* r; const double* J; int ld; int coff[5]; const double* Jx; int xcol; };   // J row stride ld, column offset per block; block with coff < 0 reads its single column from Jx[row stride 1... 2 entries]   // J row stride ld, column offset per block

// what the assembly leaves for the tail: the ordering (dropped blocks first) and the sizes; A = sc.work[0] (N x N), b = sc.work[1]
struct MargSystem { int N = 0, m = 0, md = 0, n = 0; std::vector<int> pos, keep_ids; double us_pre[3] = {0, 0, 0}; bool prof = false; };
// Everything of a marginalization up to the elimination of the dropped landmark blocks.  `done` = *out is final already (no prior / no factors / a prior that does not touch the
// dropped pose): the caller returns the status as it is.
static int marg_assemble_host(int device, hipStream_t stream, char* d_blob, double* d_ws, const DevWin& h, const uvs_window* w, const KOpts& ko,
                              int flag, uvs_prior* out, std::string& err, EvalScratch& sc, MargSystem& ms, bool& done) {
    done = true;
    const bool td_on = h.td_on != 0;
    const double eps = 1e-8;                           // marginalization_factor.h:70
    const int NFR = UVS_NF;
    // ---- GPU: evaluate all blocks with loss correction at the window's (post-solve) state
    uvs_eval ev; std::memset(&ev, 0, sizeof(ev));      // views into the handle's pinned staging buffer (run_evaluate, view mode)
    const bool prof = std::getenv("UVS_MARG_PROFILE") != nullptr;
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t0 = tnow();
    // MARGIN_OLD reads only the factors that touch frame 0 (a fifth of the window): the kernel skips the rest (mode bit 1); MARGIN_SECOND_NEW
    // reads the prior residual only, which the same subset mode delivers without evaluating a single observation of frame 0... it does evaluate
    // those, a few microseconds, to keep one code path
    std::vector<double>& prior_r_host = sc.work[10];
    if (flag == 1) {      // MARGIN_SECOND_NEW reads the prior only: its residual at the window's state is an n x n mat-vec on the host (host_prior_residual), no kernel, no copies
        if (w->prior && w->prior->n > 0) host_prior_residual(*w->prior, w, prior_r_host);
    } else {
        const int rc = run_evaluate(device, stream, d_blob, d_ws, h, ko, 1 | 2, &ev, err, sc, true);
        if (rc != UVS_OK) return rc;
    }
    const double *pt_r = ev.pt_r, *pt_J = ev.pt_J, *ln_r = ev.ln_r, *ln_J = ev.ln_J, *vp_r = ev.vp_r, *vp_J = ev.vp_J, *imu_r = ev.imu_r, *imu_J = ev.imu_J, *prior_r = flag == 1 ? prior_r_host.data() : ev.prior_r, *pt_Jtd = ev.pt_Jtd;
    auto t1 = tnow();
    // ---- host: block bookkeeping.  ids: pose f -> f ; speedbias f -> 11+f ; ex -> 22 ; td -> 23 ; point k -> 24+k ; line l -> 24+Np+l
    const int Np = w->n_points, Nl = w->n_lines, PT0 = 24, NID = PT0 + Np + Nl;
    auto lsize = [&](int id) { return id < NFR ? 6 : id < 2 * NFR ? 9 : id == 22 ? 6 : id == 23 ? 1 : id < PT0 + Np ? 1 : 4; };
    auto gsize = [&](int id) { return id < NFR ? 7 : id < 2 * NFR ? 9 : id == 22 ? 7 : id == 23 ? 1 : id < PT0 + Np ? 1 : 4; };
    std::vector<char> used(NID, 0), drop(NID, 0);
    std::vector<MFactor> fs;
    const bool have_prior = w->prior && w->prior->n > 0;
    // column map of the (single) prior factor
    std::vector<int> p_id, p_src;     // per local column: block id, source column in J0
    if (have_prior && (flag == 0 || flag == 1)) {
        const uvs_prior& p = *w->prior;
        bool use = true;
        if (flag == 1) {     // MARGIN_SECOND_NEW only if the prior touches Pose[WINDOW_SIZE-1] (estimator.cpp:1162-1163)
            use = false;
            for (int b = 0; b < p.n_blocks; ++b) if (p.block_kind[b] == UVS_BLOCK_POSE && p.block_frame[b] == UVS_WINDOW_SIZE - 1) use = true;
            if (!use) { *out = p; return UVS_OK; }
        }
        for (int b = 0; b < p.n_blocks; ++b) {
            const int id = p.block_kind[b] == UVS_BLOCK_POSE ? p.block_frame[b] : p.block_kind[b] == UVS_BLOCK_SPEEDBIAS ? NFR + p.block_frame[b] : p.block_kind[b] == UVS_BLOCK_TD ? 23 : 22;
            used[id] = 1;
            if (flag == 0 && (id == 0 || id == NFR)) drop[id] = 1;                        // drop Pose[0], SpeedBias[0]  (:1008-1015)
            if (flag == 1 && id == UVS_WINDOW_SIZE - 1) drop[id] = 1;                     // drop Pose[9]               (:1170-1176)
            for (int q = 0; q < lsize(id); ++q) { p_id.push_back(id); p_src.push_back(p.block_idx[b] + q); }
        }
    } else if (flag == 1) { std::memset(out, 0, sizeof(*out)); return UVS_OK; }
    if (flag == 0) {
        for (int b = 0; b < w->n_imu; ++b) {                                                // :1026-1035
            if (w->imu[b].frame_i != 0 || !(w->imu[b].sum_dt < 10.0)) continue;
            MFactor f; f.Jx = nullptr; f.xcol = -1; f.rows = 15; f.nb = 4; f.id[0] = 0; f.id[1] = NFR; f.id[2] = 1; f.id[3] = NFR + 1; f.sz[0] = 6; f.sz[1] = 9; f.sz[2] = 6; f.sz[3] = 9;
            f.coff[0] = 0; f.coff[1] = 6; f.coff[2] = 15; f.coff[3] = 21; f.r = &imu_r[15 * (size_t)b]; f.J = &imu_J[450 * (size_t)b]; f.ld = 30;
            for (int q = 0; q < 4; ++q) used[f.id[q]] = 1;
            drop[0] = 1; drop[NFR] = 1;
            fs.push_back(f);
        }
        for (int k = 0; k < w->n_point_obs; ++k) {                                          // :1037-1080
            if (w->pt_fi[k] != 0) continue;
            MFactor f; f.Jx = nullptr; f.xcol = -1; f.rows = 2; f.nb = 4; f.id[0] = 0; f.id[1] = w->pt_fj[k]; f.id[2] = 22; f.id[3] = PT0 + w->pt_lm[k];
            f.sz[0] = 6; f.sz[1] = 6; f.sz[2] = 6; f.sz[3] = 1; f.coff[0] = 0; f.coff[1] = 6; f.coff[2] = 12; f.coff[3] = 18;
            f.r = &pt_r[2 * (size_t)k]; f.J = &pt_J[38 * (size_t)k]; f.ld = 19;
            if (td_on) { f.nb = 5; f.id[4] = 23; f.sz[4] = 1; f.coff[4] = -1; f.Jx = &pt_Jtd[2 * (size_t)k]; }      // ProjectionTdFactor: fifth block td (estimator.cpp:1062-1070), kept
            for (int q = 0; q < f.nb; ++q) used[f.id[q]] = 1;
            drop[0] = 1; drop[f.id[3]] = 1;
            fs.push_back(f);
        }
        std::vector<int> start(Nl, -1);
        for (int k = 0; k < w->n_line_obs; ++k) if (start[w->ln_lm[k]] < 0) start[w->ln_lm[k]] = w->ln_fj[k];
        for (int k = 0; k < w->n_line_obs; ++k) {                                           // :1082-1129
            const int lm = w->ln_lm[k], fj = w->ln_fj[k];
            if (start[lm] != 0 || fj == 0) continue;
            MFactor f; f.Jx = nullptr; f.xcol = -1; f.rows = 2; f.nb = 2; f.id[0] = fj; f.id[1] = PT0 + Np + lm; f.sz[0] = 6; f.sz[1] = 4; f.coff[0] = 0; f.coff[1] = 6;
            f.r = &ln_r[2 * (size_t)k]; f.J = &ln_J[20 * (size_t)k]; f.ld = 10;
            used[f.id[0]] = 1; used[f.id[1]] = 1; drop[f.id[1]] = 1;
            fs.push_back(f);
            if (w->ln_has_vp[k]) { MFactor g = f; g.rows = 1; g.r = &vp_r[(size_t)k]; g.J = &vp_J[10 * (size_t)k]; fs.push_back(g); }
        }
    }
    if (fs.empty() && p_id.empty()) { std::memset(out, 0, sizeof(*out)); return UVS_OK; }
    // ---- ordering: dropped first
    std::vector<int> pos(NID, -1), keep_ids;
    int m = 0;
    for (int id = 0; id < NID; ++id) if (used[id] && drop[id]) { pos[id] = m; m += lsize(id); }
    int N = m;
    for (int id = 0; id < NID; ++id) if (used[id] && !drop[id]) { pos[id] = N; N += lsize(id); keep_ids.push_back(id); }
    const int n = N - m;
    if (n > UVS_MAX_PRIOR_DIM || (int)keep_ids.size() > UVS_MAX_PRIOR_BLOCKS) { err = "prior capacity"; return UVS_ERR_CAPACITY; }
    // ---- A = sum J^T J, b = sum J^T r   (ThreadsConstructA, :141-172)
    std::vector<double>&A = sc.work[0], &bv = sc.work[1];      // (kept in the handle's scratch: see EvalScratch::work)
    A.assign((size_t)N * N, 0.0); bv.assign(N, 0.0);
    if (!p_id.empty()) {
        const uvs_prior& p = *w->prior; const int pn = p.n, nc = (int)p_id.size();
        std::vector<int> gc(nc);
        { int run = 0, last = -1; for (int c2 = 0; c2 < nc; ++c2) { if (p_id[c2] != last) { last = p_id[c2]; run = 0; } gc[c2] = pos[p_id[c2]] + run++; } }
        // J0^T J0 is the single largest piece of the assembly (n^3 / 1 multiply-adds): only the half c2 <= a is accumulated, into a dense
        // nc x nc scratch with the gathered row contiguous, and mirrored when scattered (same products, same summation order over i, so
        // the values are the ones the full loop produced)
        std::vector<double>&P = sc.work[2], &rowv = sc.work[3];
        P.assign((size_t)nc * nc, 0.0); rowv.assign(nc, 0.0);
        for (int i = 0; i < pn; ++i) {
            const double* Ji = &p.linearized_jacobians[(size_t)i * pn];
            for (int a = 0; a < nc; ++a) rowv[a] = Ji[p_src[a]];
            for (int a = 0; a < nc; ++a) {
                const double ja = rowv[a];
                if (ja == 0.0) continue;
                bv[gc[a]] += ja * prior_r[i];
                row_axpy(a + 1, &P[(size_t)a * nc], rowv.data(), ja);      // P[a][c2] += ja * rowv[c2], c2 <= a (same products, same order over i)
            }
        }
        for (int a = 0; a < nc; ++a) for (int c2 = 0; c2 <= a; ++c2) {
            const double v = P[(size_t)a * nc + c2];
            A[(size_t)gc[a] * N + gc[c2]] += v;
            if (c2 != a) A[(size_t)gc[c2] * N + gc[a]] += v;
        }
    }
    for (const MFactor& f : fs) {
        int gc[32], lc[32], nc = 0;
        for (int q = 0; q < f.nb; ++q) for (int k = 0; k < f.sz[q]; ++k) { gc[nc] = pos[f.id[q]] + k; lc[nc] = f.coff[q] < 0 ? -1 : f.coff[q] + k; ++nc; }
        for (int i = 0; i < f.rows; ++i) {
            const double* Ji = f.J + (size_t)i * f.ld;
            double row[32];
            for (int a = 0; a < nc; ++a) row[a] = lc[a] < 0 ? f.Jx[i] : Ji[lc[a]];      // the td column lives in its own array (uvs_eval.pt_Jtd)
            for (int a = 0; a < nc; ++a) {
                const double ja = row[a];
                if (ja == 0.0) continue;
                bv[gc[a]] += ja * f.r[i];
                for (int c2 = 0; c2 < nc; ++c2) A[(size_t)gc[a] * N + gc[c2]] += ja * row[c2];
            }
        }
    }
    auto t2 = tnow();
    // ---- Schur complement onto the kept blocks (:263-276).  The reference forms the pseudo-inverse of the whole A_mm from one dense
    // eigen-decomposition (cut at eps).  A_mm is a small dense frame part (Pose[0] + SpeedBias[0], or Pose[9]) bordered by mutually
    // uncoupled landmark blocks (1x1 inverse depths, 4x4 lines), so the same inverse is applied here by block elimination: landmark
    // blocks first, then the <= 15 frame dofs (solve_small: Cholesky when regular, eigen-decomposition with the eps cut otherwise).
    // Identical in exact arithmetic whenever no eigenvalue of A_mm falls under eps; in floating point it agrees with an extended-precision
    // Schur complement to 1e-14 in b (tests/test_marginalization.py), and it is O(m) instead of O(m^3).
    int md = 0;
    for (int id = 0; id < PT0; ++id) if (used[id] && drop[id]) md += lsize(id);
    std::vector<int> coupled; coupled.reserve(N);
    std::vector<double>&Xk = sc.work[4], &Uk = sc.work[5];
    for (int id = PT0; id < NID; ++id) {
        if (!(used[id] && drop[id])) continue;
        const int o = pos[id], sz = lsize(id);
        double Bm[16];
        for (int i = 0; i < sz; ++i) for (int j = 0; j < sz; ++j) Bm[i * sz + j] = 0.5 * (A[(size_t)(o + i) * N + o + j] + A[(size_t)(o + j) * N + o + i]);
        coupled.clear();
        for (int i = 0; i < N; ++i) { if (i >= md && i < m) continue; bool nz = false; for (int q = 0; q < sz; ++q) nz |= A[(size_t)i * N + o + q] != 0.0; if (nz) coupled.push_back(i); }
        const int nc = (int)coupled.size(), ldx = nc + 1;
        Xk.assign((size_t)sz * ldx, 0.0);                                  // X = B^+ [A_{block, coupled} | b_block]
        for (int q = 0; q < sz; ++q) { for (int cj = 0; cj < nc; ++cj) Xk[(size_t)q * ldx + cj] = A[(size_t)(o + q) * N + coupled[cj]]; Xk[(size_t)q * ldx + nc] = bv[o + q]; }
        marg_solve_small(sz, Bm, Xk.data(), ldx, ldx, eps);
        // U = A_{coupled, block} X  (nc x (nc + 1), contiguous rows: the inner loops vectorise), then scattered; per entry the sum over q runs in
        // the same order as the scalar loop it replaces
        Uk.assign((size_t)nc * ldx, 0.0);
        for (int ci = 0; ci < nc; ++ci) {
            const double* ai = &A[(size_t)coupled[ci] * N + o];
            double* u = &Uk[(size_t)ci * ldx];
            for (int q = 0; q < sz; ++q) row_axpy(ldx, u, &Xk[(size_t)q * ldx], ai[q]);
        }
        for (int ci = 0; ci < nc; ++ci) {
            const int i = coupled[ci];
            const double* u = &Uk[(size_t)ci * ldx];
            double* Ai = &A[(size_t)i * N];
            bv[i] -= u[nc];
            for (int cj = 0; cj < nc; ++cj) Ai[coupled[cj]] -= u[cj];
        }
    }
    auto t3 = tnow();
    if (prof) { auto us = [](auto a_, auto b_) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b_ - a_).count() * 1e-3; }; ms.us_pre[0] = us(t0, t1); ms.us_pre[1] = us(t1, t2); ms.us_pre[2] = us(t2, t3); }
    ms.N = N; ms.m = m; ms.md = md; ms.n = n; ms.pos = pos; ms.keep_ids = keep_ids; ms.prof = prof;
    done = false;
    return UVS_OK;
}
static int run_marginalize(int device, hipStream_t stream, char* d_blob, double* d_ws, const DevWin& h, const uvs_window* w, const KOpts& ko,
                           int flag, uvs_prior* out, std::string& err, EvalScratch& sc) {
    MargSystem ms; bool done = false;
    const int rc = marg_assemble_host(device, stream, d_blob, d_ws, h, w, ko, flag, out, err, sc, ms, done);
    if (rc != UVS_OK || done) return rc;
    return marg_finish(ms.N, ms.m, ms.md, ms.n, sc.work[0], sc.work[1], ms.pos, ms.keep_ids, w, flag, out, sc, ms.prof, ms.us_pre);
}

}  // namespace uvsdev
