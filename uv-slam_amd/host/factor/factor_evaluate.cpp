// factor_evaluate.cpp -- the per-block Evaluate() / operator() surface of the reference's factor classes (SURVEY.md 8b "Factor API
// surface to keep"), routed through uvs_evaluate(): every call packs ONE residual block into a uvs_window whose frames 0 and 1 hold the
// two pose / speed-bias blocks, runs the HIP evaluation kernel without the loss (a cost function's Evaluate() knows no loss) and
// scatters the local-size Jacobians into the global-size row-major buffers Ceres expects (7th column of a pose block = 0:
// projection_factor.cpp:122, imu_factor.h:106, marginalization_factor.cpp:376).
#include <cstring>
#include <vector>
#include "factors.h"
#include "../parameters.h"

namespace {
uvs_solver* g_eval_solver = nullptr;

void empty_window(uvs_window& w) {
    std::memset(&w, 0, sizeof(w));
    for (int f = 0; f < UVS_NUM_FRAMES; ++f) w.pose[f][6] = 1.0;
    w.ex_pose[6] = 1.0; w.relo_pose[6] = 1.0;
}
void put_extrinsic(uvs_window& w, const Eigen::Matrix3d& ric, const Eigen::Vector3d& tic) {
    const Eigen::Quaterniond q{ric};
    const double e[7] = {tic.x(), tic.y(), tic.z(), q.x(), q.y(), q.z(), q.w()};
    std::memcpy(w.ex_pose, e, sizeof(e));
}
// rows x local -> rows x global (one zero column appended when global = local + 1)
void widen(const double* J, int rows, int ld, int col0, int local, int global, double* out) {
    for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < local; ++c) out[r * global + c] = J[r * ld + col0 + c];
        for (int c = local; c < global; ++c) out[r * global + c] = 0.0;
    }
}
// line / VP functors share everything but the output row
bool line_block(const Eigen::Matrix3d& ric, const Eigen::Vector3d& tic, const Eigen::Vector3d& sp, const Eigen::Vector3d& ep, const Eigen::Vector3d* vp,
                const double* pose, const double* line, double* residuals, double* J_pose, double* J_line) {
    uvs_solver* s = uvs::evaluation_solver();
    if (!s) return false;
    uvs_window w; empty_window(w);
    std::memcpy(w.pose[0], pose, 7 * sizeof(double));
    put_extrinsic(w, ric, tic);
    double orth[4]; std::memcpy(orth, line, sizeof(orth));
    int32_t lm = 0, fj = 0, has_vp = vp ? 1 : 0;
    const double spv[3] = {sp.x(), sp.y(), sp.z()}, epv[3] = {ep.x(), ep.y(), ep.z()}, vpv[3] = {vp ? vp->x() : 0.0, vp ? vp->y() : 0.0, vp ? vp->z() : 0.0};
    w.n_lines = 1; w.n_line_obs = 1; w.line_orth = orth; w.ln_lm = &lm; w.ln_fj = &fj; w.ln_has_vp = &has_vp; w.ln_sp = spv; w.ln_ep = epv; w.ln_vp = vpv;
    double ln_r[2], ln_J[20], vp_r[1], vp_J[10];
    uvs_eval ev; std::memset(&ev, 0, sizeof(ev));
    ev.ln_r = ln_r; ev.ln_J = ln_J; ev.vp_r = vp_r; ev.vp_J = vp_J;
    if (uvs_evaluate(s, &w, 0, &ev) != UVS_OK) return false;
    const int rows = vp ? 1 : 2;
    const double* r = vp ? vp_r : ln_r; const double* J = vp ? vp_J : ln_J;      // [rows][6 pose | 4 line]
    for (int k = 0; k < rows; ++k) residuals[k] = r[k];
    if (J_pose) widen(J, rows, 10, 0, 6, 7, J_pose);
    if (J_line) widen(J, rows, 10, 6, 4, 4, J_line);
    return true;
}
}  // namespace

void uvs::set_evaluation_solver(uvs_solver* s) { g_eval_solver = s; }
uvs_solver* uvs::evaluation_solver() { return g_eval_solver; }

bool ProjectionFactor::Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    uvs_solver* s = uvs::evaluation_solver();
    if (!s) return false;
    uvs_window w; empty_window(w);
    std::memcpy(w.pose[0], parameters[0], 7 * sizeof(double));
    std::memcpy(w.pose[1], parameters[1], 7 * sizeof(double));
    std::memcpy(w.ex_pose, parameters[2], 7 * sizeof(double));
    double inv_depth = parameters[3][0];
    int32_t lm = 0, fi = 0, fj = 1;
    const double pi[3] = {pts_i.x(), pts_i.y(), pts_i.z()}, pj[3] = {pts_j.x(), pts_j.y(), pts_j.z()};
    w.n_points = 1; w.n_point_obs = 1; w.inv_depth = &inv_depth; w.pt_lm = &lm; w.pt_fi = &fi; w.pt_fj = &fj; w.pt_pi = pi; w.pt_pj = pj;
    double r[2], J[38];      // [2][pose_i 6 | pose_j 6 | ex 6 | lambda]
    uvs_eval ev; std::memset(&ev, 0, sizeof(ev)); ev.pt_r = r; ev.pt_J = J;
    if (uvs_evaluate(s, &w, 0, &ev) != UVS_OK) return false;
    residuals[0] = r[0]; residuals[1] = r[1];
    if (jacobians) {
        for (int b = 0; b < 3; ++b) if (jacobians[b]) widen(J, 2, 19, 6 * b, 6, 7, jacobians[b]);
        if (jacobians[3]) widen(J, 2, 19, 18, 1, 1, jacobians[3]);
    }
    return true;
}

// the forward-difference self test shared by ProjectionFactor::check and ProjectionTdFactor::check (projection_factor.cpp:178-283,
// projection_td_factor.cpp:147-...): `n_scalar` trailing 1-dof blocks after the three pose blocks
namespace {
template <class Factor> double projection_check(const Factor& f, double** parameters, int n_scalar) {
    double r0[2], Jp[3][14], Js[2][2]; double* jac[5] = {Jp[0], Jp[1], Jp[2], Js[0], Js[1]};
    if (!f.Evaluate(parameters, r0, jac)) return -1.0;
    const double eps = 1e-6;
    double worst = 0.0;
    for (int k = 0; k < 18 + n_scalar; ++k) {      // 6 + 6 + 6 tangent directions, then the scalars
        double P[3][7], sc[2] = {parameters[3][0], n_scalar > 1 ? parameters[4][0] : 0.0};
        for (int b = 0; b < 3; ++b) std::memcpy(P[b], parameters[b], sizeof(P[b]));
        if (k < 18) {
            const int b = k / 6, a = k % 6;
            if (a < 3) P[b][a] += eps;
            else {
                Eigen::Vector3d d; d(a - 3) = eps;
                const Eigen::Quaterniond q = (Eigen::Quaterniond(P[b][6], P[b][3], P[b][4], P[b][5]) * Utility::deltaQ(d)).normalized();
                P[b][3] = q.x(); P[b][4] = q.y(); P[b][5] = q.z(); P[b][6] = q.w();
            }
        } else sc[k - 18] += eps;
        const double* pp[5] = {P[0], P[1], P[2], &sc[0], &sc[1]};
        double r1[2];
        if (!f.Evaluate(pp, r1, nullptr)) return -1.0;
        for (int row = 0; row < 2; ++row) {
            const double analytic = k < 18 ? Jp[k / 6][row * 7 + k % 6] : Js[k - 18][row];
            worst = std::max(worst, std::fabs((r1[row] - r0[row]) / eps - analytic));
        }
    }
    return worst;
}
}  // namespace

double ProjectionFactor::check(double** parameters) const { return projection_check(*this, parameters, 1); }

bool ProjectionTdFactor::Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    uvs_solver* s = uvs::evaluation_solver();
    if (!s) return false;
    uvs_window w; empty_window(w);
    std::memcpy(w.pose[0], parameters[0], 7 * sizeof(double));
    std::memcpy(w.pose[1], parameters[1], 7 * sizeof(double));
    std::memcpy(w.ex_pose, parameters[2], 7 * sizeof(double));
    double inv_depth = parameters[3][0];
    w.td = parameters[4][0];
    int32_t lm = 0, fi = 0, fj = 1;
    const double pi[3] = {pts_i.x(), pts_i.y(), pts_i.z()}, pj[3] = {pts_j.x(), pts_j.y(), pts_j.z()};
    const double vi[2] = {velocity_i.x(), velocity_i.y()}, vj[2] = {velocity_j.x(), velocity_j.y()};
    const double tdi = td_i - TR / ROW * row_i, tdj = td_j - TR / ROW * row_j;      // the rolling-shutter term folded into the capture offset (include/uvs_solver.h)
    w.n_points = 1; w.n_point_obs = 1; w.inv_depth = &inv_depth; w.pt_lm = &lm; w.pt_fi = &fi; w.pt_fj = &fj; w.pt_pi = pi; w.pt_pj = pj;
    w.pt_vel_i = vi; w.pt_vel_j = vj; w.pt_td_i = &tdi; w.pt_td_j = &tdj;
    double r[2], J[38], Jtd[2];      // [2][pose_i 6 | pose_j 6 | ex 6 | lambda], d r / d td
    uvs_eval ev; std::memset(&ev, 0, sizeof(ev)); ev.pt_r = r; ev.pt_J = J; ev.pt_Jtd = Jtd;
    if (uvs_evaluate(s, &w, 0, &ev) != UVS_OK) return false;      // (a handle without estimate_td rejects the time-offset arrays)
    residuals[0] = r[0]; residuals[1] = r[1];
    if (jacobians) {
        for (int b = 0; b < 3; ++b) if (jacobians[b]) widen(J, 2, 19, 6 * b, 6, 7, jacobians[b]);
        if (jacobians[3]) widen(J, 2, 19, 18, 1, 1, jacobians[3]);
        if (jacobians[4]) { jacobians[4][0] = Jtd[0]; jacobians[4][1] = Jtd[1]; }
    }
    return true;
}
double ProjectionTdFactor::check(double** parameters) const { return projection_check(*this, parameters, 2); }

bool IMUFactor::Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    uvs_solver* s = uvs::evaluation_solver();
    if (!s || !pre_integration) return false;
    uvs_window w; empty_window(w);
    std::memcpy(w.pose[0], parameters[0], 7 * sizeof(double)); std::memcpy(w.speedbias[0], parameters[1], 9 * sizeof(double));
    std::memcpy(w.pose[1], parameters[2], 7 * sizeof(double)); std::memcpy(w.speedbias[1], parameters[3], 9 * sizeof(double));
    const IntegrationBase* p = pre_integration;
    uvs_imu_block b; std::memset(&b, 0, sizeof(b));
    b.sum_dt = p->sum_dt;
    for (int k = 0; k < 3; ++k) { b.delta_p[k] = p->delta_p(k); b.delta_v[k] = p->delta_v(k); b.linearized_ba[k] = p->linearized_ba(k); b.linearized_bg[k] = p->linearized_bg(k); }
    b.delta_q[0] = p->delta_q.x(); b.delta_q[1] = p->delta_q.y(); b.delta_q[2] = p->delta_q.z(); b.delta_q[3] = p->delta_q.w();
    std::memcpy(b.jacobian, p->jacobian, sizeof(b.jacobian)); std::memcpy(b.covariance, p->covariance, sizeof(b.covariance));
    b.frame_i = 0; b.skip = 0;
    w.n_imu = 1; w.imu = &b;
    double r[15]; std::vector<double> J(15 * 30);      // [15][pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9]
    uvs_eval ev; std::memset(&ev, 0, sizeof(ev)); ev.imu_r = r; ev.imu_J = J.data();
    if (uvs_evaluate(s, &w, 0, &ev) != UVS_OK) return false;
    std::memcpy(residuals, r, sizeof(r));
    if (jacobians) {
        if (jacobians[0]) widen(J.data(), 15, 30, 0, 6, 7, jacobians[0]);
        if (jacobians[1]) widen(J.data(), 15, 30, 6, 9, 9, jacobians[1]);
        if (jacobians[2]) widen(J.data(), 15, 30, 15, 6, 7, jacobians[2]);
        if (jacobians[3]) widen(J.data(), 15, 30, 21, 9, 9, jacobians[3]);
    }
    return true;
}

bool LineProjectionFactor::EvaluateWithJacobians(const double* pose, const double* line, double* residuals, double* J_pose, double* J_line) const {
    return line_block(ric, tic, sp, ep, nullptr, pose, line, residuals, J_pose, J_line);
}
bool VPProjectionFactor::EvaluateWithJacobians(const double* pose, const double* line, double* residuals, double* J_pose, double* J_line) const {
    return line_block(ric, tic, sp, ep, &vp, pose, line, residuals, J_pose, J_line);
}

bool MarginalizationFactor::Evaluate(double const* const* parameters, double* residuals, double** jacobians) const {
    uvs_solver* s = uvs::evaluation_solver();
    if (!s || !marginalization_info || marginalization_info->prior.n <= 0) return false;
    const uvs_prior& p = marginalization_info->prior;
    uvs_window w; empty_window(w);
    for (int b = 0; b < p.n_blocks; ++b) {      // the kept blocks land where their (kind, frame) says
        const double* x = parameters[b];
        switch (p.block_kind[b]) {
            case UVS_BLOCK_POSE: std::memcpy(w.pose[p.block_frame[b]], x, 7 * sizeof(double)); break;
            case UVS_BLOCK_SPEEDBIAS: std::memcpy(w.speedbias[p.block_frame[b]], x, 9 * sizeof(double)); break;
            case UVS_BLOCK_EX_POSE: std::memcpy(w.ex_pose, x, 7 * sizeof(double)); break;
            default: w.td = x[0]; break;
        }
    }
    w.prior = &p;
    std::vector<double> r(UVS_MAX_PRIOR_DIM);
    uvs_eval ev; std::memset(&ev, 0, sizeof(ev)); ev.prior_r = r.data();
    if (uvs_evaluate(s, &w, 0, &ev) != UVS_OK) return false;
    const int n = p.n;
    for (int i = 0; i < n; ++i) residuals[i] = r[i];
    if (jacobians)
        for (int b = 0; b < p.n_blocks; ++b) {
            if (!jacobians[b]) continue;
            const int size = p.block_size[b], local = size == 7 ? 6 : size;
            widen(p.linearized_jacobians, n, n, p.block_idx[b], local, size, jacobians[b]);
        }
    return true;
}
