// uvs_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement (FP64, single thread) of the solve inside UV-SLAM's
// Estimator::optimization() (reference vins_estimator/src/estimator.cpp:761-997):
// residual blocks (oracle_factors.h) + the Ceres trust-region Levenberg-Marquardt
// loop with SPARSE_SCHUR semantics, restated from SURVEY.md Appendix B.
//
// PARITY UNPINNED: Ceres is an un-vendored, un-pinned dependency of the
// reference (vins_estimator/CMakeLists.txt:22) and is not available here; the
// reference has no tests / golden vectors.  This file is pinned by the
// known-answer, finite-difference and torch-autograd tests under tests/.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// liboracle.so.  The product (uv-slam_amd/) never links or calls it.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "oracle_factors.h"
#include "oracle_marg.h"

namespace orc {

struct Block {             // one residual block after loss correction, LOCAL columns
    int rows = 0;
    std::vector<int> col;      // global local-parameter index of each column
    std::vector<double> r;     // rows
    std::vector<double> J;     // rows x col.size(), row-major
    int lm_off = -1;           // first landmark column index (global), -1 if none
    int lm_dim = 0;
};

struct State {
    double pose[UVS_NUM_FRAMES][7];
    double sb[UVS_NUM_FRAMES][9];
    double ex[7];
    double td;
    double relo[7];            // relo_Pose (estimator.cpp:947-948), only a parameter block when the window carries relocalization blocks
    std::vector<double> invd, line;
};

struct Problem {
    const uvs_options* opt;
    const uvs_window* w;
    int F, Np, Nl, P;
    bool ex_free, td_free, relo_on;
    std::vector<double> W;    // n_imu x 225 sqrt_info
    int off_pose(int f) const { return 15 * f; }
    int off_sb(int f) const { return 15 * f + 6; }
    int off_ex() const { return 15 * UVS_NUM_FRAMES; }
    int off_td() const { return 15 * UVS_NUM_FRAMES + (ex_free ? 6 : 0); }      // para_Td, 1 dof (estimator.cpp:790-797)
    int off_relo() const { return 15 * UVS_NUM_FRAMES + (ex_free ? 6 : 0) + (td_free ? 1 : 0); }      // relo_Pose, 6 local dofs
    int off_pt(int k) const { return F + k; }
    int off_ln(int l) const { return F + Np + 4 * l; }
};

static void init_problem(Problem& pb, const uvs_options* opt, const uvs_window* w) {
    pb.opt = opt; pb.w = w;
    pb.ex_free = opt->estimate_extrinsic != 0;
    pb.td_free = opt->estimate_td != 0;
    pb.relo_on = w->n_relo_obs > 0;
    pb.F = 15 * UVS_NUM_FRAMES + (pb.ex_free ? 6 : 0) + (pb.td_free ? 1 : 0) + (pb.relo_on ? 6 : 0);
    pb.Np = w->n_points; pb.Nl = w->n_lines;
    pb.P = pb.F + pb.Np + 4 * pb.Nl;
    pb.W.assign((size_t)std::max(w->n_imu, 0) * 225, 0.0);
    for (int b = 0; b < w->n_imu; ++b) imu_sqrt_info(w->imu[b].covariance, &pb.W[(size_t)b * 225]);
}

static void init_state(State& x, const uvs_window* w) {
    std::memcpy(x.pose, w->pose, sizeof(x.pose));
    std::memcpy(x.sb, w->speedbias, sizeof(x.sb));
    std::memcpy(x.ex, w->ex_pose, sizeof(x.ex));
    x.td = w->td;
    std::memcpy(x.relo, w->relo_pose, sizeof(x.relo));
    x.invd.assign(w->inv_depth, w->inv_depth + w->n_points);
    x.line.assign(w->line_orth, w->line_orth + 4 * (size_t)w->n_lines);
}

// Evaluate every residual block (problem order of estimator.cpp: prior, IMU, points, lines(+VP)).
// If blocks != nullptr Jacobians are produced.  robust applies the Cauchy corrector.  Returns cost.
static double evaluate(const Problem& pb, const State& x, bool robust, std::vector<Block>* blocks, uvs_eval* dump) {
    const uvs_window* w = pb.w; const uvs_options* o = pb.opt;
    double cost = 0.0;
    if (blocks) blocks->clear();
    // ---- prior (estimator.cpp:803-809), no loss
    if (w->prior && w->prior->n > 0) {
        const uvs_prior& p = *w->prior;
        const int n = p.n;
        std::vector<double> dx(n), r(n);
        auto get = [&](int kind, int frame) -> const double* {
            switch (kind) { case UVS_BLOCK_POSE: return x.pose[frame]; case UVS_BLOCK_SPEEDBIAS: return x.sb[frame];
                            case UVS_BLOCK_EX_POSE: return x.ex; default: return &x.td; } };
        prior_eval(p, get, dx.data(), r.data());
        double s = 0.0; for (int i = 0; i < n; ++i) s += r[i] * r[i];
        cost += 0.5 * s;
        if (dump && dump->prior_r) for (int i = 0; i < n; ++i) dump->prior_r[i] = r[i];
        if (blocks) {
            Block B; B.rows = n; B.r = r;
            std::vector<int> src;   // source column in J0 for each kept column
            for (int b = 0; b < p.n_blocks; ++b) {
                int base = -1, local = p.block_size[b] == 7 ? 6 : p.block_size[b];
                if (p.block_kind[b] == UVS_BLOCK_POSE) base = pb.off_pose(p.block_frame[b]);
                else if (p.block_kind[b] == UVS_BLOCK_SPEEDBIAS) base = pb.off_sb(p.block_frame[b]);
                else if (p.block_kind[b] == UVS_BLOCK_EX_POSE) base = pb.ex_free ? pb.off_ex() : -1;   // constant block dropped
                else if (p.block_kind[b] == UVS_BLOCK_TD) base = pb.td_free ? pb.off_td() : -1;
                if (base < 0) continue;
                for (int k = 0; k < local; ++k) { B.col.push_back(base + k); src.push_back(p.block_idx[b] + k); }   // marginalization_factor.cpp:368-378
            }
            const int nc = (int)B.col.size();
            B.J.resize((size_t)n * nc);
            for (int i = 0; i < n; ++i) for (int c = 0; c < nc; ++c) B.J[(size_t)i * nc + c] = p.linearized_jacobians[(size_t)i * n + src[c]];
            blocks->push_back(std::move(B));
        }
    }
    // ---- IMU (estimator.cpp:811-818), no loss
    for (int b = 0; b < w->n_imu; ++b) {
        const uvs_imu_block& ib = w->imu[b];
        if (ib.skip) { if (dump && dump->imu_r) { std::memset(dump->imu_r + 15 * b, 0, 15 * 8); if (dump->imu_J) std::memset(dump->imu_J + 450 * b, 0, 450 * 8); } continue; }
        const int i = ib.frame_i, j = i + 1;
        double r[15], J[450];
        imu_eval(ib, &pb.W[(size_t)b * 225], o->gravity, x.pose[i], x.sb[i], x.pose[j], x.sb[j], r, (blocks || dump) ? J : nullptr);
        double s = 0.0; for (int k = 0; k < 15; ++k) s += r[k] * r[k];
        cost += 0.5 * s;
        if (dump && dump->imu_r) { std::memcpy(dump->imu_r + 15 * b, r, sizeof(r)); if (dump->imu_J) std::memcpy(dump->imu_J + 450 * b, J, sizeof(J)); }
        if (blocks) {
            Block B; B.rows = 15; B.r.assign(r, r + 15); B.J.assign(J, J + 450);
            for (int k = 0; k < 6; ++k) B.col.push_back(pb.off_pose(i) + k);
            for (int k = 0; k < 9; ++k) B.col.push_back(pb.off_sb(i) + k);
            for (int k = 0; k < 6; ++k) B.col.push_back(pb.off_pose(j) + k);
            for (int k = 0; k < 9; ++k) B.col.push_back(pb.off_sb(j) + k);
            blocks->push_back(std::move(B));
        }
    }
    // ---- points (estimator.cpp:823-866), CauchyLoss(1.0)
    int relo_next = 0;
    for (int k = 0; k < w->n_point_obs; ++k) {
        const int lm = w->pt_lm[k], fi = w->pt_fi[k], fj = w->pt_fj[k];
        double r[2], J[40];
        const bool wantJ = blocks || dump;
        const int ld = pb.td_free ? 20 : 19;        // ProjectionTdFactor (estimator.cpp:853-858) has a fifth block, td
        if (pb.td_free) point_td_eval(x.pose[fi], x.pose[fj], x.ex, x.invd[lm], w->pt_pi + 3 * k, w->pt_pj + 3 * k, w->pt_vel_i + 2 * k, w->pt_vel_j + 2 * k,
                                      w->pt_td_i[k], w->pt_td_j[k], x.td, o->point_sqrt_info, r, wantJ ? J : nullptr);
        else point_eval(x.pose[fi], x.pose[fj], x.ex, x.invd[lm], w->pt_pi + 3 * k, w->pt_pj + 3 * k, o->point_sqrt_info, r, wantJ ? J : nullptr);
        if (robust) cost += 0.5 * cauchy_correct(o->loss_point, 2, ld, r, wantJ ? J : nullptr);
        else cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
        if (dump && dump->pt_r) {
            dump->pt_r[2 * k] = r[0]; dump->pt_r[2 * k + 1] = r[1];
            if (dump->pt_J) for (int i = 0; i < 2; ++i) for (int c = 0; c < 19; ++c) dump->pt_J[38 * k + 19 * i + c] = J[i * ld + c];
            if (dump->pt_Jtd && pb.td_free) { dump->pt_Jtd[2 * k] = J[19]; dump->pt_Jtd[2 * k + 1] = J[39]; }
        }
        if (blocks) {
            Block B; B.rows = 2; B.r.assign(r, r + 2);
            std::vector<int> src;
            for (int c = 0; c < 6; ++c) { B.col.push_back(pb.off_pose(fi) + c); src.push_back(c); }
            for (int c = 0; c < 6; ++c) { B.col.push_back(pb.off_pose(fj) + c); src.push_back(6 + c); }
            if (pb.ex_free) for (int c = 0; c < 6; ++c) { B.col.push_back(pb.off_ex() + c); src.push_back(12 + c); }
            if (pb.td_free) { B.col.push_back(pb.off_td()); src.push_back(19); }
            B.lm_off = pb.off_pt(lm); B.lm_dim = 1;
            B.col.push_back(B.lm_off); src.push_back(18);
            const int nc = (int)B.col.size();
            B.J.resize(2 * nc);
            for (int i = 0; i < 2; ++i) for (int c = 0; c < nc; ++c) B.J[i * nc + c] = J[i * ld + src[c]];
            blocks->push_back(std::move(B));
        }
        // ---- relocalization block of this landmark (estimator.cpp:944-975): the plain ProjectionFactor between the landmark's start
        // frame and relo_Pose, CauchyLoss(1.0).  The reference appends these after the line blocks; here each one follows its landmark's
        // last ordinary observation so that the Schur elimination below sees a landmark's blocks together (only the summation order
        // of the cost differs).
        if (pb.relo_on && (k + 1 == w->n_point_obs || w->pt_lm[k + 1] != lm)) {
            while (relo_next < w->n_relo_obs && w->relo_lm[relo_next] < lm) ++relo_next;
            if (relo_next < w->n_relo_obs && w->relo_lm[relo_next] == lm) {
                const int q = relo_next++;
                double rr[2], Jr[40];
                point_eval(x.pose[fi], x.relo, x.ex, x.invd[lm], w->relo_pi + 3 * q, w->relo_pj + 3 * q, o->point_sqrt_info, rr, blocks ? Jr : nullptr);
                if (robust) cost += 0.5 * cauchy_correct(o->loss_point, 2, 19, rr, blocks ? Jr : nullptr);
                else cost += 0.5 * (rr[0] * rr[0] + rr[1] * rr[1]);
                if (blocks) {
                    Block B; B.rows = 2; B.r.assign(rr, rr + 2);
                    std::vector<int> src;
                    for (int c = 0; c < 6; ++c) { B.col.push_back(pb.off_pose(fi) + c); src.push_back(c); }
                    for (int c = 0; c < 6; ++c) { B.col.push_back(pb.off_relo() + c); src.push_back(6 + c); }
                    if (pb.ex_free) for (int c = 0; c < 6; ++c) { B.col.push_back(pb.off_ex() + c); src.push_back(12 + c); }
                    B.lm_off = pb.off_pt(lm); B.lm_dim = 1;
                    B.col.push_back(B.lm_off); src.push_back(18);
                    const int nc = (int)B.col.size();
                    B.J.resize(2 * nc);
                    for (int i = 0; i < 2; ++i) for (int c = 0; c < nc; ++c) B.J[i * nc + c] = Jr[i * 19 + src[c]];
                    blocks->push_back(std::move(B));
                }
            }
        }
    }
    // ---- lines + VP (estimator.cpp:868-927), CauchyLoss(0.1) / CauchyLoss(1.0)
    for (int k = 0; k < w->n_line_obs; ++k) {
        const int lm = w->ln_lm[k], fj = w->ln_fj[k];
        const double* lp = &x.line[4 * (size_t)lm];
        {
            double r[2], J[20];
            line_eval(x.pose[fj], lp, x.ex, w->ln_sp + 3 * k, w->ln_ep + 3 * k, o->line_factor, r, (blocks || dump) ? J : nullptr);
            if (robust) cost += 0.5 * cauchy_correct(o->loss_line, 2, 10, r, (blocks || dump) ? J : nullptr);
            else cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
            if (dump && dump->ln_r) { dump->ln_r[2 * k] = r[0]; dump->ln_r[2 * k + 1] = r[1]; if (dump->ln_J) std::memcpy(dump->ln_J + 20 * k, J, sizeof(J)); }
            if (blocks) {
                Block B; B.rows = 2; B.r.assign(r, r + 2); B.J.assign(J, J + 20);
                for (int c = 0; c < 6; ++c) B.col.push_back(pb.off_pose(fj) + c);
                B.lm_off = pb.off_ln(lm); B.lm_dim = 4;
                for (int c = 0; c < 4; ++c) B.col.push_back(B.lm_off + c);
                blocks->push_back(std::move(B));
            }
        }
        if (w->ln_has_vp[k]) {
            double r[1], J[10];
            vp_eval(x.pose[fj], lp, x.ex, w->ln_vp + 3 * k, o->vp_factor, r, (blocks || dump) ? J : nullptr);
            if (robust) cost += 0.5 * cauchy_correct(o->loss_vp, 1, 10, r, (blocks || dump) ? J : nullptr);
            else cost += 0.5 * r[0] * r[0];
            if (dump && dump->vp_r) { dump->vp_r[k] = r[0]; if (dump->vp_J) std::memcpy(dump->vp_J + 10 * k, J, sizeof(J)); }
            if (blocks) {
                Block B; B.rows = 1; B.r.assign(r, r + 1); B.J.assign(J, J + 10);
                for (int c = 0; c < 6; ++c) B.col.push_back(pb.off_pose(fj) + c);
                B.lm_off = pb.off_ln(lm); B.lm_dim = 4;
                for (int c = 0; c < 4; ++c) B.col.push_back(B.lm_off + c);
                blocks->push_back(std::move(B));
            }
        } else if (dump && dump->vp_r) { dump->vp_r[k] = 0.0; if (dump->vp_J) std::memset(dump->vp_J + 10 * k, 0, 80); }
    }
    return cost;
}

// x_plus_delta = Plus(x, delta) over every block (pose blocks on the manifold, the rest additive;
// the 4-vector line block has NO local parameterization, Appendix D2 / B.7).
static void plus(const Problem& pb, const State& x, const double* d, State& out) {
    out = x;
    for (int f = 0; f < UVS_NUM_FRAMES; ++f) {
        pose_plus(x.pose[f], d + pb.off_pose(f), out.pose[f]);
        for (int k = 0; k < 9; ++k) out.sb[f][k] = x.sb[f][k] + d[pb.off_sb(f) + k];
    }
    if (pb.ex_free) pose_plus(x.ex, d + pb.off_ex(), out.ex);
    if (pb.td_free) out.td = x.td + d[pb.off_td()];
    if (pb.relo_on) pose_plus(x.relo, d + pb.off_relo(), out.relo);
    for (int k = 0; k < pb.Np; ++k) out.invd[k] = x.invd[k] + d[pb.off_pt(k)];
    for (int k = 0; k < 4 * pb.Nl; ++k) out.line[k] = x.line[k] + d[pb.F + pb.Np + k];
}

// ambient-coordinate norms used by Ceres' parameter tolerance (Appendix B.4)
static double ambient_sqnorm(const Problem& pb, const State& x, const State* y) {
    double s = 0.0;
    auto acc = [&](const double* a, const double* b, int n) { for (int i = 0; i < n; ++i) { double v = b ? a[i] - b[i] : a[i]; s += v * v; } };
    for (int f = 0; f < UVS_NUM_FRAMES; ++f) { acc(x.pose[f], y ? y->pose[f] : nullptr, 7); acc(x.sb[f], y ? y->sb[f] : nullptr, 9); }
    if (pb.ex_free) acc(x.ex, y ? y->ex : nullptr, 7);
    if (pb.td_free) acc(&x.td, y ? &y->td : nullptr, 1);
    if (pb.relo_on) acc(x.relo, y ? y->relo : nullptr, 7);
    acc(x.invd.data(), y ? y->invd.data() : nullptr, pb.Np);
    acc(x.line.data(), y ? y->line.data() : nullptr, 4 * pb.Nl);
    return s;
}
static double ambient_maxdiff(const Problem& pb, const State& x, const State& y) {
    double m = 0.0;
    auto acc = [&](const double* a, const double* b, int n) { for (int i = 0; i < n; ++i) m = std::fmax(m, std::fabs(a[i] - b[i])); };
    for (int f = 0; f < UVS_NUM_FRAMES; ++f) { acc(x.pose[f], y.pose[f], 7); acc(x.sb[f], y.sb[f], 9); }
    if (pb.ex_free) acc(x.ex, y.ex, 7);
    if (pb.td_free) acc(&x.td, &y.td, 1);
    if (pb.relo_on) acc(x.relo, y.relo, 7);
    acc(x.invd.data(), y.invd.data(), pb.Np);
    acc(x.line.data(), y.line.data(), 4 * pb.Nl);
    return m;
}

// dense in-place lower Cholesky + solve; A is n x n row-major (lower used). returns false if not PD
static bool chol_solve_inplace(int n, std::vector<double>& A, std::vector<double>& b) {
    for (int j = 0; j < n; ++j) {
        double* Aj = &A[(size_t)j * n];
        double d = Aj[j];
        for (int k = 0; k < j; ++k) d -= Aj[k] * Aj[k];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        const double ljj = std::sqrt(d);
        Aj[j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double* Ai = &A[(size_t)i * n];
            double s = Ai[j];
            for (int k = 0; k < j; ++k) s -= Ai[k] * Aj[k];
            Ai[j] = s / ljj;
        }
    }
    for (int i = 0; i < n; ++i) { double s = b[i]; const double* Ai = &A[(size_t)i * n]; for (int k = 0; k < i; ++k) s -= Ai[k] * b[k]; b[i] = s / Ai[i]; }
    for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
    return true;
}

// Solve (Js^T Js + D^2) y = -Js^T r  exactly, Js = J*diag(scale).
// mode 0: landmark Schur elimination onto the frame block (SPARSE_SCHUR semantics), dense Cholesky of the reduced system.
// mode 1: dense Cholesky of the full system (cross-check; small windows only).
static bool linear_solve(const Problem& pb, const std::vector<Block>& blocks, const std::vector<double>& scale,
                         const std::vector<double>& D2, int mode, std::vector<double>& y) {
    const int P = pb.P, F = pb.F;
    y.assign(P, 0.0);
    if (mode == 1) {
        std::vector<double> H((size_t)P * P, 0.0), g(P, 0.0);
        for (const Block& B : blocks) {
            const int nc = (int)B.col.size();
            for (int i = 0; i < B.rows; ++i) {
                const double* Ji = &B.J[(size_t)i * nc];
                for (int a = 0; a < nc; ++a) {
                    const double ja = Ji[a] * scale[B.col[a]];
                    if (ja == 0.0) continue;
                    g[B.col[a]] += ja * B.r[i];
                    for (int c = 0; c < nc; ++c) { if (B.col[c] <= B.col[a]) H[(size_t)B.col[a] * P + B.col[c]] += ja * Ji[c] * scale[B.col[c]]; }
                }
            }
        }
        for (int k = 0; k < P; ++k) { H[(size_t)k * P + k] += D2[k]; g[k] = -g[k]; }
        if (!chol_solve_inplace(P, H, g)) return false;
        y = g;
        return true;
    }
    // ---- Schur
    std::vector<double> S((size_t)F * F, 0.0), gf(F, 0.0);
    struct LM { int off, dim; double Minv[16]; double gl[4]; std::vector<int> rows; std::vector<double> Wt; /* rows.size() x dim */ };
    std::vector<LM> lms;
    size_t bi = 0;
    std::vector<double> Wd((size_t)F * 4, 0.0);
    std::vector<char> touched(F, 0);
    while (bi < blocks.size()) {
        const Block& B0 = blocks[bi];
        if (B0.lm_off < 0) {     // frame-only block: straight into S
            const int nc = (int)B0.col.size();
            for (int i = 0; i < B0.rows; ++i) {
                const double* Ji = &B0.J[(size_t)i * nc];
                for (int a = 0; a < nc; ++a) {
                    const double ja = Ji[a] * scale[B0.col[a]];
                    if (ja == 0.0) continue;
                    gf[B0.col[a]] += ja * B0.r[i];
                    for (int c = 0; c < nc; ++c) S[(size_t)B0.col[a] * F + B0.col[c]] += ja * Ji[c] * scale[B0.col[c]];
                }
            }
            ++bi; continue;
        }
        // group of consecutive blocks sharing the same landmark
        size_t be = bi;
        while (be < blocks.size() && blocks[be].lm_off == B0.lm_off) ++be;
        LM L; L.off = B0.lm_off; L.dim = B0.lm_dim;
        const int d = L.dim;
        double Hll[16] = {0}, gl[4] = {0};
        for (size_t q = bi; q < be; ++q) {
            const Block& B = blocks[q];
            const int nc = (int)B.col.size(), nf = nc - d;
            for (int i = 0; i < B.rows; ++i) {
                const double* Ji = &B.J[(size_t)i * nc];
                double jl[4];
                for (int c = 0; c < d; ++c) jl[c] = Ji[nf + c] * scale[B.col[nf + c]];
                for (int a = 0; a < d; ++a) { gl[a] += jl[a] * B.r[i]; for (int c = 0; c < d; ++c) Hll[a * d + c] += jl[a] * jl[c]; }
                for (int a = 0; a < nf; ++a) {
                    const int ga = B.col[a];
                    const double ja = Ji[a] * scale[ga];
                    gf[ga] += ja * B.r[i];
                    for (int c = 0; c < nf; ++c) S[(size_t)ga * F + B.col[c]] += ja * Ji[c] * scale[B.col[c]];
                    if (!touched[ga]) { touched[ga] = 1; L.rows.push_back(ga); }
                    for (int c = 0; c < d; ++c) Wd[(size_t)ga * 4 + c] += ja * jl[c];
                }
            }
        }
        for (int a = 0; a < d; ++a) Hll[a * d + a] += D2[L.off + a];
        // Minv = (Hll + D^2)^-1 via Cholesky
        double Lc[16];
        if (!chol_lower(d, Hll, Lc)) return false;
        for (int c = 0; c < d; ++c) {   // solve for each unit vector
            double e[4] = {0, 0, 0, 0}; e[c] = 1.0;
            for (int i = 0; i < d; ++i) { double s = e[i]; for (int k = 0; k < i; ++k) s -= Lc[i * d + k] * e[k]; e[i] = s / Lc[i * d + i]; }
            for (int i = d - 1; i >= 0; --i) { double s = e[i]; for (int k = i + 1; k < d; ++k) s -= Lc[k * d + i] * e[k]; e[i] = s / Lc[i * d + i]; }
            for (int i = 0; i < d; ++i) L.Minv[i * d + c] = e[i];
        }
        std::sort(L.rows.begin(), L.rows.end());
        const int nr = (int)L.rows.size();
        L.Wt.resize((size_t)nr * d);
        std::vector<double> WM((size_t)nr * d);
        for (int a = 0; a < nr; ++a) {
            for (int c = 0; c < d; ++c) L.Wt[(size_t)a * d + c] = Wd[(size_t)L.rows[a] * 4 + c];
            for (int c = 0; c < d; ++c) { double s = 0.0; for (int k = 0; k < d; ++k) s += L.Wt[(size_t)a * d + k] * L.Minv[k * d + c]; WM[(size_t)a * d + c] = s; }
        }
        for (int a = 0; a < nr; ++a) {
            double s = 0.0; for (int k = 0; k < d; ++k) s += WM[(size_t)a * d + k] * gl[k];
            gf[L.rows[a]] -= s;
            for (int b = 0; b < nr; ++b) { double t = 0.0; for (int k = 0; k < d; ++k) t += WM[(size_t)a * d + k] * L.Wt[(size_t)b * d + k]; S[(size_t)L.rows[a] * F + L.rows[b]] -= t; }
        }
        for (int c = 0; c < d; ++c) L.gl[c] = gl[c];
        for (int a = 0; a < nr; ++a) { touched[L.rows[a]] = 0; for (int c = 0; c < 4; ++c) Wd[(size_t)L.rows[a] * 4 + c] = 0.0; }
        lms.push_back(std::move(L));
        bi = be;
    }
    for (int k = 0; k < F; ++k) { S[(size_t)k * F + k] += D2[k]; gf[k] = -gf[k]; }
    if (!chol_solve_inplace(F, S, gf)) return false;
    for (int k = 0; k < F; ++k) y[k] = gf[k];
    // landmarks that never appeared keep y = 0 (their gradient is 0)
    for (const LM& L : lms) {
        const int d = L.dim, nr = (int)L.rows.size();
        double rhs[4];
        for (int c = 0; c < d; ++c) { double s = -L.gl[c]; for (int a = 0; a < nr; ++a) s -= L.Wt[(size_t)a * d + c] * y[L.rows[a]]; rhs[c] = s; }
        for (int i = 0; i < d; ++i) { double s = 0.0; for (int c = 0; c < d; ++c) s += L.Minv[i * d + c] * rhs[c]; y[L.off + i] = s; }
    }
    return true;
}

static void copy_state_out(const Problem& pb, const State& x, uvs_state* out) {
    std::memcpy(out->pose, x.pose, sizeof(x.pose));
    std::memcpy(out->speedbias, x.sb, sizeof(x.sb));
    std::memcpy(out->ex_pose, x.ex, sizeof(x.ex));
    out->td = x.td;
    std::memcpy(out->relo_pose, x.relo, sizeof(x.relo));
    if (out->inv_depth) std::memcpy(out->inv_depth, x.invd.data(), sizeof(double) * pb.Np);
    if (out->line_orth) std::memcpy(out->line_orth, x.line.data(), sizeof(double) * 4 * pb.Nl);
}

// Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy, restated.
static int solve(const uvs_options* opt, const uvs_window* w, int linear_mode, uvs_state* out, uvs_report* rep) {
    Problem pb; init_problem(pb, opt, w);
    State x; init_state(x, w);
    State xc = x;
    const int P = pb.P;
    std::memset(rep, 0, sizeof(*rep));
    std::vector<Block> blocks;
    std::vector<double> g(P), scale(P, 1.0), diag(P), D2(P), y, delta(P), negg(P);

    auto gradient_and_norms = [&](double* gmax) {
        std::fill(g.begin(), g.end(), 0.0);
        for (const Block& B : blocks) { const int nc = (int)B.col.size(); for (int i = 0; i < B.rows; ++i) for (int c = 0; c < nc; ++c) g[B.col[c]] += B.J[(size_t)i * nc + c] * B.r[i]; }
        for (int k = 0; k < P; ++k) negg[k] = -g[k];
        State xg; plus(pb, x, negg.data(), xg);
        *gmax = ambient_maxdiff(pb, x, xg);            // ||x - Plus(x,-g)||_inf
    };

    double cost = evaluate(pb, x, true, &blocks, nullptr);
    double gmax; gradient_and_norms(&gmax);
    if (opt->jacobi_scaling) {                          // computed ONCE from the initial Jacobian (B.2)
        std::vector<double> cn(P, 0.0);
        for (const Block& B : blocks) { const int nc = (int)B.col.size(); for (int i = 0; i < B.rows; ++i) for (int c = 0; c < nc; ++c) { double v = B.J[(size_t)i * nc + c]; cn[B.col[c]] += v * v; } }
        for (int k = 0; k < P; ++k) scale[k] = 1.0 / (1.0 + std::sqrt(cn[k]));
    }
    double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int invalid = 0, it = 0, nsucc = 0;
    double x_norm = std::sqrt(ambient_sqnorm(pb, x, nullptr));
    rep->initial_cost = cost; rep->cost[0] = cost; rep->radius[0] = radius; rep->gradient_max_norm[0] = gmax; rep->accepted[0] = 1;
    int term = UVS_TERM_NO_CONVERGENCE;
    int status = UVS_OK;
    if (!std::isfinite(cost)) { term = UVS_TERM_NUMERIC_FAILURE; status = UVS_ERR_NUMERIC; }
    else if (gmax <= opt->gradient_tolerance) term = UVS_TERM_GRADIENT_TOL;
    else while (true) {
        // FinalizeIterationAndCheckIfMinimizerCanContinue
        if (it >= opt->max_num_iterations) { term = UVS_TERM_NO_CONVERGENCE; break; }
        if (gmax <= opt->gradient_tolerance) { term = UVS_TERM_GRADIENT_TOL; break; }
        if (radius <= opt->min_trust_region_radius) { term = UVS_TERM_MIN_RADIUS; break; }
        ++it;
        const int ti = std::min(it, UVS_MAX_ITER);
        // LevenbergMarquardtStrategy::ComputeStep
        if (!reuse_diagonal) {
            std::fill(diag.begin(), diag.end(), 0.0);
            for (const Block& B : blocks) { const int nc = (int)B.col.size(); for (int i = 0; i < B.rows; ++i) for (int c = 0; c < nc; ++c) { double v = B.J[(size_t)i * nc + c] * scale[B.col[c]]; diag[B.col[c]] += v * v; } }
            for (int k = 0; k < P; ++k) diag[k] = std::fmin(std::fmax(diag[k], opt->min_lm_diagonal), opt->max_lm_diagonal);
        }
        for (int k = 0; k < P; ++k) D2[k] = diag[k] / radius;
        reuse_diagonal = true;
        bool ok = linear_solve(pb, blocks, scale, D2, linear_mode, y);
        double model_cost_change = 0.0;
        if (ok) {
            for (int k = 0; k < P; ++k) if (!std::isfinite(y[k])) { ok = false; break; }
        }
        if (ok) {   // model_cost_change = -m.(r + m/2), m = Js*y   (B.3)
            for (const Block& B : blocks) {
                const int nc = (int)B.col.size();
                for (int i = 0; i < B.rows; ++i) { double m = 0.0; for (int c = 0; c < nc; ++c) m += B.J[(size_t)i * nc + c] * scale[B.col[c]] * y[B.col[c]]; model_cost_change -= m * (B.r[i] + m / 2.0); }
            }
        }
        rep->model_cost_change[ti] = model_cost_change;
        if (!ok || !(model_cost_change > 0.0)) {         // invalid step (HandleInvalidStep)
            ++invalid;
            rep->accepted[ti] = -1; rep->cost[ti] = cost; rep->candidate_cost[ti] = cost;
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            rep->radius[ti] = radius; rep->gradient_max_norm[ti] = gmax;
            if (invalid >= opt->max_consecutive_invalid_steps) { term = UVS_TERM_INVALID_STEPS; break; }
            continue;
        }
        invalid = 0;
        for (int k = 0; k < P; ++k) delta[k] = y[k] * scale[k];
        plus(pb, x, delta.data(), xc);
        double cand = evaluate(pb, xc, true, nullptr, nullptr);
        if (!std::isfinite(cand)) cand = 1.7976931348623157e308;
        rep->candidate_cost[ti] = cand;
        const double step_norm = std::sqrt(ambient_sqnorm(pb, x, &xc));
        rep->step_norm[ti] = step_norm;
        rep->cost[ti] = cost; rep->radius[ti] = radius; rep->gradient_max_norm[ti] = gmax;
        const double rel = (cost - cand) / model_cost_change;
        rep->relative_decrease[ti] = rel;
        const bool successful = rel > opt->min_relative_decrease;
        if (opt->function_tol_keeps_candidate && successful) {
            // variant: accept first, then test tolerances (recorded option, Appendix B.4)
        }
        if (step_norm <= opt->parameter_tolerance * (x_norm + opt->parameter_tolerance)) {
            if (opt->function_tol_keeps_candidate && successful) { x = xc; cost = cand; rep->cost[ti] = cost; rep->accepted[ti] = 1; ++nsucc; }
            term = UVS_TERM_PARAMETER_TOL; break;
        }
        if (std::fabs(cost - cand) <= opt->function_tolerance * cost) {
            if (opt->function_tol_keeps_candidate && successful) { x = xc; cost = cand; rep->cost[ti] = cost; rep->accepted[ti] = 1; ++nsucc; }
            term = UVS_TERM_FUNCTION_TOL; break;
        }
        if (successful) {                                  // HandleSuccessfulStep
            x = xc; ++nsucc;
            x_norm = std::sqrt(ambient_sqnorm(pb, x, nullptr));
            cost = evaluate(pb, x, true, &blocks, nullptr);
            gradient_and_norms(&gmax);
            radius = radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3));
            radius = std::fmin(opt->max_trust_region_radius, radius);
            decrease_factor = 2.0; reuse_diagonal = false;
            rep->accepted[ti] = 1;
        } else {                                           // HandleUnsuccessfulStep
            radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            rep->accepted[ti] = 0;
        }
        rep->cost[ti] = cost; rep->radius[ti] = radius; rep->gradient_max_norm[ti] = gmax;
    }
    rep->status = status; rep->termination = term; rep->num_iterations = it; rep->num_successful = nsucc; rep->final_cost = cost;
    copy_state_out(pb, x, out);
    return status;
}


// ---------------------------------------------------------------- marginalization
// Estimator::optimization(), estimator.cpp:1002-1228 restated on block indices.
static int marginalize(const uvs_options* opt, const uvs_window* w, int flag, uvs_prior* out) {
    Problem pb; init_problem(pb, opt, w);
    State x; init_state(x, w);
    MargIds ids{pb.Np, pb.Nl};
    const uvs_options* o = opt;
    std::vector<int> local_size(ids.count(), 0), global_size(ids.count(), 0);
    for (int f = 0; f < UVS_NUM_FRAMES; ++f) { local_size[ids.pose(f)] = 6; global_size[ids.pose(f)] = 7; local_size[ids.sb(f)] = 9; global_size[ids.sb(f)] = 9; }
    local_size[ids.ex()] = 6; global_size[ids.ex()] = 7;
    local_size[ids.td()] = 1; global_size[ids.td()] = 1;
    for (int k = 0; k < pb.Np; ++k) { local_size[ids.pt(k)] = 1; global_size[ids.pt(k)] = 1; }
    for (int l = 0; l < pb.Nl; ++l) { local_size[ids.ln(l)] = 4; global_size[ids.ln(l)] = 4; }
    std::vector<MargFactor> factors;
    std::vector<char> used(ids.count(), 0), dropped(ids.count(), 0);
    const bool have_prior = w->prior && w->prior->n > 0;

    auto add_prior_factor = [&](auto is_dropped) {                        // MarginalizationFactor as a ResidualBlockInfo, no loss
        const uvs_prior& p = *w->prior;
        const int n = p.n;
        MargFactor f; f.rows = n; f.r.resize(n);
        std::vector<double> dx(n);
        auto get = [&](int kind, int frame) -> const double* {
            switch (kind) { case UVS_BLOCK_POSE: return x.pose[frame]; case UVS_BLOCK_SPEEDBIAS: return x.sb[frame];
                            case UVS_BLOCK_EX_POSE: return x.ex; default: return &x.td; } };
        prior_eval(p, get, dx.data(), f.r.data());
        std::vector<int> src;
        for (int b = 0; b < p.n_blocks; ++b) {
            int id = p.block_kind[b] == UVS_BLOCK_POSE ? ids.pose(p.block_frame[b]) : p.block_kind[b] == UVS_BLOCK_SPEEDBIAS ? ids.sb(p.block_frame[b]) : p.block_kind[b] == UVS_BLOCK_TD ? ids.td() : ids.ex();
            f.blk.push_back(id); f.bsz.push_back(local_size[id]);
            used[id] = 1; if (is_dropped(p.block_kind[b], p.block_frame[b])) dropped[id] = 1;
            for (int k = 0; k < local_size[id]; ++k) src.push_back(p.block_idx[b] + k);
        }
        const int nc = (int)src.size();
        f.J.resize((size_t)n * nc);
        for (int i = 0; i < n; ++i) for (int c = 0; c < nc; ++c) f.J[(size_t)i * nc + c] = p.linearized_jacobians[(size_t)i * n + src[c]];
        factors.push_back(std::move(f));
    };

    if (flag == 0) {   // MARGIN_OLD :1003-1159
        if (have_prior) add_prior_factor([](int kind, int frame) { return (kind == UVS_BLOCK_POSE || kind == UVS_BLOCK_SPEEDBIAS) && frame == 0; });   // :1008-1024
        for (int b = 0; b < w->n_imu; ++b) {                                                                        // :1026-1035
            const uvs_imu_block& ib = w->imu[b];
            if (ib.frame_i != 0 || !(ib.sum_dt < 10.0)) continue;
            MargFactor f; f.rows = 15; f.r.resize(15); f.J.resize(450);
            imu_eval(ib, &pb.W[(size_t)b * 225], o->gravity, x.pose[0], x.sb[0], x.pose[1], x.sb[1], f.r.data(), f.J.data());
            f.blk = {ids.pose(0), ids.sb(0), ids.pose(1), ids.sb(1)}; f.bsz = {6, 9, 6, 9};
            for (int id : f.blk) used[id] = 1;
            dropped[ids.pose(0)] = 1; dropped[ids.sb(0)] = 1;
            factors.push_back(std::move(f));
        }
        for (int k = 0; k < w->n_point_obs; ++k) {                                                                   // :1037-1080
            if (w->pt_fi[k] != 0) continue;
            const int lm = w->pt_lm[k], fj = w->pt_fj[k];
            MargFactor f; f.rows = 2; f.r.resize(2);
            if (pb.td_free) {      // ProjectionTdFactor with its fifth block td, which is kept (estimator.cpp:1062-1070: drop_set {0, 3})
                f.J.resize(40);
                point_td_eval(x.pose[0], x.pose[fj], x.ex, x.invd[lm], w->pt_pi + 3 * k, w->pt_pj + 3 * k, w->pt_vel_i + 2 * k, w->pt_vel_j + 2 * k,
                              w->pt_td_i[k], w->pt_td_j[k], x.td, o->point_sqrt_info, f.r.data(), f.J.data());
                cauchy_correct(o->loss_point, 2, 20, f.r.data(), f.J.data());
                f.blk = {ids.pose(0), ids.pose(fj), ids.ex(), ids.pt(lm), ids.td()}; f.bsz = {6, 6, 6, 1, 1};
            } else {
                f.J.resize(38);
                point_eval(x.pose[0], x.pose[fj], x.ex, x.invd[lm], w->pt_pi + 3 * k, w->pt_pj + 3 * k, o->point_sqrt_info, f.r.data(), f.J.data());
                cauchy_correct(o->loss_point, 2, 19, f.r.data(), f.J.data());
                f.blk = {ids.pose(0), ids.pose(fj), ids.ex(), ids.pt(lm)}; f.bsz = {6, 6, 6, 1};
            }
            for (int id : f.blk) used[id] = 1;
            dropped[ids.pose(0)] = 1; dropped[ids.pt(lm)] = 1;                                                     // drop_set {0,3}
            factors.push_back(std::move(f));
        }
        {                                                                                                            // :1082-1129
            std::vector<int> start(pb.Nl, -1);
            for (int k = 0; k < w->n_line_obs; ++k) if (start[w->ln_lm[k]] < 0) start[w->ln_lm[k]] = w->ln_fj[k];
            for (int k = 0; k < w->n_line_obs; ++k) {
                const int lm = w->ln_lm[k], fj = w->ln_fj[k];
                if (start[lm] != 0 || fj == 0) continue;                                                            // :1092-1103
                const double* lp = &x.line[4 * (size_t)lm];
                MargFactor f; f.rows = 2; f.r.resize(2); f.J.resize(20);
                line_eval(x.pose[fj], lp, x.ex, w->ln_sp + 3 * k, w->ln_ep + 3 * k, o->line_factor, f.r.data(), f.J.data());
                cauchy_correct(o->loss_line, 2, 10, f.r.data(), f.J.data());
                f.blk = {ids.pose(fj), ids.ln(lm)}; f.bsz = {6, 4};
                used[ids.pose(fj)] = 1; used[ids.ln(lm)] = 1; dropped[ids.ln(lm)] = 1;                              // drop_set {1}
                factors.push_back(std::move(f));
                if (w->ln_has_vp[k]) {
                    MargFactor g; g.rows = 1; g.r.resize(1); g.J.resize(10);
                    vp_eval(x.pose[fj], lp, x.ex, w->ln_vp + 3 * k, o->vp_factor, g.r.data(), g.J.data());
                    cauchy_correct(o->loss_vp, 1, 10, g.r.data(), g.J.data());
                    g.blk = {ids.pose(fj), ids.ln(lm)}; g.bsz = {6, 4};
                    factors.push_back(std::move(g));
                }
            }
        }
    } else {           // MARGIN_SECOND_NEW :1160-1228
        bool touches = false;
        if (have_prior) for (int b = 0; b < w->prior->n_blocks; ++b) if (w->prior->block_kind[b] == UVS_BLOCK_POSE && w->prior->block_frame[b] == UVS_WINDOW_SIZE - 1) touches = true;
        if (!touches) { if (have_prior) *out = *w->prior; else std::memset(out, 0, sizeof(*out)); return UVS_OK; }
        add_prior_factor([](int kind, int frame) { return kind == UVS_BLOCK_POSE && frame == UVS_WINDOW_SIZE - 1; });
    }
    if (factors.empty()) { std::memset(out, 0, sizeof(*out)); return UVS_OK; }
    std::vector<int> order_drop, order_keep;
    for (int id = 0; id < ids.count(); ++id) if (used[id] && dropped[id]) order_drop.push_back(id);
    // kept order: Pose ascending, SpeedBias ascending, Ex
    for (int id = 0; id < ids.count(); ++id) if (used[id] && !dropped[id]) order_keep.push_back(id);
    std::vector<int> pos_of(ids.count(), -1);
    int m = 0, n = 0;
    std::vector<double> J0, r0;
    marginalize_core(factors, order_drop, order_keep, local_size, pos_of, &m, &n, J0, r0);
    if (n > UVS_MAX_PRIOR_DIM || (int)order_keep.size() > UVS_MAX_PRIOR_BLOCKS) return UVS_ERR_CAPACITY;
    std::memset(out, 0, sizeof(*out));
    out->n = n; out->n_blocks = (int)order_keep.size();
    int xo = 0;
    for (int b = 0; b < out->n_blocks; ++b) {
        const int id = order_keep[b];
        int kind, frame = 0; const double* data;
        if (id < UVS_NUM_FRAMES) { kind = UVS_BLOCK_POSE; frame = id; data = x.pose[frame]; }
        else if (id < 2 * UVS_NUM_FRAMES) { kind = UVS_BLOCK_SPEEDBIAS; frame = id - UVS_NUM_FRAMES; data = x.sb[frame]; }
        else if (id == ids.td()) { kind = UVS_BLOCK_TD; data = &x.td; }
        else { kind = UVS_BLOCK_EX_POSE; data = x.ex; }
        // addr_shift (estimator.cpp:1139-1152 / :1196-1219)
        int nf = frame;
        if (kind == UVS_BLOCK_POSE || kind == UVS_BLOCK_SPEEDBIAS) { if (flag == 0) nf = frame - 1; else nf = (frame == UVS_WINDOW_SIZE) ? frame - 1 : frame; }
        out->block_kind[b] = kind; out->block_frame[b] = nf; out->block_size[b] = global_size[id];
        out->block_idx[b] = pos_of[id] - m; out->x0_off[b] = xo;
        for (int k = 0; k < global_size[id]; ++k) out->x0[xo + k] = data[k];
        xo += global_size[id];
    }
    for (int i = 0; i < n; ++i) { out->linearized_residuals[i] = r0[i]; for (int j = 0; j < n; ++j) out->linearized_jacobians[(size_t)i * n + j] = J0[(size_t)i * n + j]; }
    return UVS_OK;
}

}  // namespace orc

extern "C" {

int oracle_solve(const uvs_options* opt, const uvs_window* w, int linear_mode, uvs_state* out, uvs_report* rep) {
    if (!opt || !w || !out || !rep) return UVS_ERR_INVALID_ARG;
    if (opt->estimate_td && w->n_point_obs > 0 && (!w->pt_vel_i || !w->pt_vel_j || !w->pt_td_i || !w->pt_td_j)) return UVS_ERR_INVALID_ARG;
    return orc::solve(opt, w, linear_mode, out, rep);
}

int oracle_evaluate(const uvs_options* opt, const uvs_window* w, int robust, uvs_eval* out) {
    if (!opt || !w || !out) return UVS_ERR_INVALID_ARG;
    orc::Problem pb; orc::init_problem(pb, opt, w);
    pb.relo_on = false;      // like uvs_evaluate / the marginalization (estimator.cpp:1002-1228): relocalization blocks are solve-only
    orc::State x; orc::init_state(x, w);
    out->cost = orc::evaluate(pb, x, robust != 0, nullptr, out);
    return UVS_OK;
}

int oracle_marginalize(const uvs_options* opt, const uvs_window* w, int flag, uvs_prior* out) {
    if (!opt || !w || !out) return UVS_ERR_INVALID_ARG;
    return orc::marginalize(opt, w, flag, out);
}

// symmetric eigen-decomposition (Jacobi), for tests: A n x n row-major in, V columns out, ev out
void oracle_sym_eig(int n, const double* A, double* V, double* ev) {
    std::vector<double> a(A, A + (size_t)n * n), v, e;
    orc::sym_eig_jacobi(n, a, v, e);
    std::memcpy(V, v.data(), sizeof(double) * n * n); std::memcpy(ev, e.data(), sizeof(double) * n);
}

// IMU whitening matrix (15x15 row-major upper) for tests
int oracle_imu_sqrt_info(const double* cov, double* W) { return orc::imu_sqrt_info(cov, W) ? UVS_OK : UVS_ERR_NUMERIC; }

// Plus on a pose block, for tests
void oracle_pose_plus(const double* x, const double* d, double* out) { orc::pose_plus(x, d, out); }

}  // extern "C"
