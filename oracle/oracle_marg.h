// oracle_marg.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the marginalization prior construction:
//   ResidualBlockInfo::Evaluate            marginalization_factor.cpp:3-69
//   MarginalizationInfo::preMarginalize    :110-129
//   MarginalizationInfo::marginalize       :174-297
//   MarginalizationInfo::getParameterBlocks:299-319
//   orchestration in Estimator::optimization  estimator.cpp:1002-1228
// Block ordering is made deterministic (the reference iterates an
// unordered_map keyed by pointer value): dropped blocks
// first in the order {Pose, SpeedBias, points, lines}, kept blocks in the order
// {Pose ascending, SpeedBias ascending, Ex_Pose}.  J0/r0 are therefore only
// comparable through J0^T J0 and J0^T r0 with the reference, but are directly
// comparable with the HIP/host implementation, which uses the same order.
#pragma once
#include <vector>
#include <algorithm>
#include <cmath>
#include "oracle_factors.h"

namespace orc {

// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (row-major).
// On return A is destroyed, V holds eigenvectors in COLUMNS, ev the eigenvalues.
inline void sym_eig_jacobi(int n, std::vector<double>& A, std::vector<double>& V, std::vector<double>& ev) {
    V.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dsum = 0.0;
        for (int i = 0; i < n; ++i) { dsum += A[(size_t)i * n + i] * A[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += A[(size_t)i * n + j] * A[(size_t)i * n + j]; }
        if (off == 0.0) break;
        // RELATIVE stopping rule |a_pq| <= eps sqrt(a_pp a_qq) (not off <= eps ||diag||): A_mm is graded over ten orders of magnitude and
        // its SMALL eigenvalues are the ones that get inverted -- an absolute rule leaves them with 1e-5 relative error
        int rotations = 0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q];
                if (std::fabs(apq) <= 1.1e-16 * std::sqrt(std::fabs(app * aqq))) continue;
                ++rotations;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
                    A[(size_t)k * n + p] = c * akp - s * akq; A[(size_t)k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
                    A[(size_t)p * n + k] = c * apk - s * aqk; A[(size_t)q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = c * vkp - s * vkq; V[(size_t)k * n + q] = s * vkp + c * vkq;
                }
            }
        }
        if (rotations == 0) break;
    }
    ev.resize(n);
    for (int i = 0; i < n; ++i) ev[i] = A[(size_t)i * n + i];
}

struct MargFactor {           // one ResidualBlockInfo after Evaluate()
    int rows;
    std::vector<int> blk;     // parameter block ids
    std::vector<int> bsz;     // local sizes
    std::vector<double> r;
    std::vector<double> J;    // rows x sum(bsz), row-major, columns concatenated in blk order
};

// block id space for marginalization
struct MargIds {
    int Np, Nl;
    int pose(int f) const { return f; }
    int sb(int f) const { return UVS_NUM_FRAMES + f; }
    int ex() const { return 2 * UVS_NUM_FRAMES; }
    int td() const { return 2 * UVS_NUM_FRAMES + 1; }
    int pt(int k) const { return 2 * UVS_NUM_FRAMES + 2 + k; }
    int ln(int l) const { return 2 * UVS_NUM_FRAMES + 2 + Np + l; }
    int count() const { return 2 * UVS_NUM_FRAMES + 2 + Np + Nl; }
};

// Core of MarginalizationInfo::marginalize(). pos_of[id] = column offset (dropped first), returns m, n, J0 (n x n), r0 (n)
inline bool marginalize_core(const std::vector<MargFactor>& factors, const std::vector<int>& order_drop, const std::vector<int>& order_keep,
                             const std::vector<int>& local_size, std::vector<int>& pos_of, int* m_out, int* n_out,
                             std::vector<double>& J0, std::vector<double>& r0) {
    const double eps = 1e-8;                                            // marginalization_factor.h:70
    int pos = 0;
    for (int id : order_drop) { pos_of[id] = pos; pos += local_size[id]; }
    const int m = pos;
    for (int id : order_keep) { pos_of[id] = pos; pos += local_size[id]; }
    const int n = pos - m, N = pos;
    std::vector<double> A((size_t)N * N, 0.0), b(N, 0.0);
    for (const MargFactor& f : factors) {                               // ThreadsConstructA :141-172 (summed in one thread)
        int tot = 0; for (int s : f.bsz) tot += s;
        std::vector<int> gcol(tot);
        { int c = 0; for (size_t q = 0; q < f.blk.size(); ++q) for (int k = 0; k < f.bsz[q]; ++k) gcol[c++] = pos_of[f.blk[q]] + k; }
        for (int i = 0; i < f.rows; ++i) {
            const double* Ji = &f.J[(size_t)i * tot];
            for (int a = 0; a < tot; ++a) {
                if (Ji[a] == 0.0) continue;
                b[gcol[a]] += Ji[a] * f.r[i];
                for (int c = 0; c < tot; ++c) A[(size_t)gcol[a] * N + gcol[c]] += Ji[a] * Ji[c];
            }
        }
    }
    // Amm = 0.5*(Amm + Amm^T) ; pseudo-inverse through eigen-decomposition (:263-268)
    std::vector<double> Amm((size_t)m * m), V, ev;
    for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) Amm[(size_t)i * m + j] = 0.5 * (A[(size_t)i * N + j] + A[(size_t)j * N + i]);
    sym_eig_jacobi(m, Amm, V, ev);
    std::vector<double> Ainv((size_t)m * m, 0.0);
    for (int k = 0; k < m; ++k) {
        if (!(ev[k] > eps)) continue;
        const double inv = 1.0 / ev[k];
        for (int i = 0; i < m; ++i) { const double vi = V[(size_t)i * m + k] * inv; if (vi == 0.0) continue; for (int j = 0; j < m; ++j) Ainv[(size_t)i * m + j] += vi * V[(size_t)j * m + k]; }
    }
    // Schur: A <- Arr - Arm Amm^-1 Amr ; b <- brr - Arm Amm^-1 bmm  (:270-276)
    std::vector<double> T((size_t)n * m, 0.0);     // Arm * Ainv
    for (int i = 0; i < n; ++i) for (int k = 0; k < m; ++k) { const double a = A[(size_t)(m + i) * N + k]; if (a == 0.0) continue; for (int j = 0; j < m; ++j) T[(size_t)i * m + j] += a * Ainv[(size_t)k * m + j]; }
    std::vector<double> Ar((size_t)n * n), br(n);
    for (int i = 0; i < n; ++i) {
        double s = b[m + i]; for (int k = 0; k < m; ++k) s -= T[(size_t)i * m + k] * b[k]; br[i] = s;
        for (int j = 0; j < n; ++j) { double t = A[(size_t)(m + i) * N + (m + j)]; for (int k = 0; k < m; ++k) t -= T[(size_t)i * m + k] * A[(size_t)k * N + (m + j)]; Ar[(size_t)i * n + j] = t; }
    }
    // second eigen-decomposition (:278-291).  The reference feeds A un-symmetrised to
    // SelfAdjointEigenSolver, which reads the lower triangle only; we do the same.
    std::vector<double> As((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) As[(size_t)i * n + j] = (j <= i) ? Ar[(size_t)i * n + j] : Ar[(size_t)j * n + i];
    std::vector<double> V2, ev2;
    sym_eig_jacobi(n, As, V2, ev2);
    // sort ascending like Eigen (row order of J0 follows eigenvalue order)
    std::vector<int> idx(n); for (int i = 0; i < n; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int c) { return ev2[a] < ev2[c]; });
    J0.assign((size_t)n * n, 0.0); r0.assign(n, 0.0);
    for (int row = 0; row < n; ++row) {
        const int k = idx[row];
        const double S = ev2[k] > eps ? ev2[k] : 0.0;
        const double Sinv = ev2[k] > eps ? 1.0 / ev2[k] : 0.0;
        const double ss = std::sqrt(S), si = std::sqrt(Sinv);
        double vb = 0.0;
        for (int j = 0; j < n; ++j) { J0[(size_t)row * n + j] = ss * V2[(size_t)j * n + k]; vb += V2[(size_t)j * n + k] * br[j]; }
        r0[row] = si * vb;
    }
    *m_out = m; *n_out = n;
    return true;
}

}  // namespace orc
