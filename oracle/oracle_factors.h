// oracle_factors.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the residual blocks on the hot path of UV-SLAM's
// Estimator::optimization().  PARITY UNPINNED: the reference ships no tests or
// golden vectors and cannot be compiled here (no Eigen/Ceres/ROS); this
// restatement is pinned only by the analytic / autodiff / known-answer tests in
// tests/ (SURVEY.md section 8c).
//
// Each function cites the reference file:line it follows.  Jacobians are
// returned in LOCAL size (pose = 6 columns): Ceres multiplies the 7-wide global
// Jacobian by PoseLocalParameterization::ComputeJacobian = [I6;0]
// (pose_local_parameterization.cpp:20-27), i.e. keeps columns 0..5.
#pragma once
#include "oracle_math.h"
#include "../include/uvs_solver.h"

namespace orc {

// VP guard (documented deviation): the reference's Jet
// derivative -1/sqrt(1-c^2) is inf/NaN when the line direction is (numerically)
// parallel to the VP.  Below this sin^2 threshold both oracle and HIP path use
// residual = VP_FACTOR*acos(min(|c|,1)) and a zero Jacobian.
static const double kVpSin2Guard = 1e-14;

// ---- a10: ceres::CauchyLoss + Corrector, mirrored in ResidualBlockInfo::Evaluate
// (marginalization_factor.cpp:37-68). Returns rho[0]; scales r (rows) and J (rows x cols) in place.
inline double cauchy_correct(double a, int rows, int cols, double* r, double* J) {
    double sq_norm = 0.0;
    for (int i = 0; i < rows; ++i) sq_norm += r[i] * r[i];
    const double b = a * a, c = 1.0 / b;
    const double sum = 1.0 + sq_norm * c, inv = 1.0 / sum;
    double rho[3];
    rho[0] = b * std::log(sum);
    rho[1] = std::fmax(2.2250738585072014e-308, inv);
    rho[2] = -c * (inv * inv);
    const double sqrt_rho1 = std::sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq_norm == 0.0 || rho[2] <= 0.0) {
        residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0;
    } else {   // never taken for Cauchy (rho'' < 0), kept for fidelity
        const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(D);
        residual_scaling = sqrt_rho1 / (1.0 - alpha);
        alpha_sq_norm = alpha / sq_norm;
    }
    if (J) {
        for (int cidx = 0; cidx < cols; ++cidx) {
            double rtj = 0.0;
            for (int i = 0; i < rows; ++i) rtj += r[i] * J[i * cols + cidx];
            for (int i = 0; i < rows; ++i) J[i * cols + cidx] = sqrt_rho1 * (J[i * cols + cidx] - alpha_sq_norm * r[i] * rtj);
        }
    }
    for (int i = 0; i < rows; ++i) r[i] *= residual_scaling;
    return rho[0];
}

// ---- a3: PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-19)
inline void pose_plus(const double* x, const double* d, double* out) {
    out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; out[2] = x[2] + d[2];
    Qd q = quat_xyzw(x + 3);
    Qd dq = {1.0, d[3] / 2.0, d[4] / 2.0, d[5] / 2.0};     // Utility::deltaQ, utility.h:11-24
    Qd r = qnormalized(qmul(q, dq));
    out[3] = r.x; out[4] = r.y; out[5] = r.z; out[6] = r.w;
}

// ---- a5: ProjectionFactor::Evaluate (projection_factor.cpp:22-175)
// J layout: 2 x 19 row-major = [pose_i(6) | pose_j(6) | ex(6) | lambda(1)]
inline void point_eval(const double* pose_i, const double* pose_j, const double* ex, double inv_dep,
                       const double* pi3, const double* pj3, double sqrt_info, double* r, double* J) {
    V3d Pi = v3(pose_i), Pj = v3(pose_j), tic = v3(ex);
    Qd Qi = quat_xyzw(pose_i + 3), Qj = quat_xyzw(pose_j + 3), qic = quat_xyzw(ex + 3);
    V3d pts_i = v3(pi3), pts_j = v3(pj3);
    V3d pts_camera_i = {pts_i.x / inv_dep, pts_i.y / inv_dep, pts_i.z / inv_dep};       // :44
    V3d pts_imu_i = qrot(qic, pts_camera_i) + tic;                                      // :45
    V3d pts_w = qrot(Qi, pts_imu_i) + Pi;                                               // :46
    V3d pts_imu_j = qrot(qinv(Qj), pts_w - Pj);                                         // :47
    V3d pts_camera_j = qrot(qinv(qic), pts_imu_j - tic);                                // :48
    const double dep_j = pts_camera_j.z;                                                // :56
    r[0] = sqrt_info * (pts_camera_j.x / dep_j - pts_j.x);                              // :57,:67
    r[1] = sqrt_info * (pts_camera_j.y / dep_j - pts_j.y);
    if (!J) return;
    M3d Ri = qmat(Qi), Rj = qmat(Qj), ric = qmat(qic);                                  // :74-76
    double reduce[2][3] = {{1.0 / dep_j, 0.0, -pts_camera_j.x / (dep_j * dep_j)},       // :90-91
                           {0.0, 1.0 / dep_j, -pts_camera_j.y / (dep_j * dep_j)}};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 3; ++j) reduce[i][j] *= sqrt_info;   // :93
    M3d ricT = transpose(ric), RjT = transpose(Rj);
    M3d A = mul(ricT, RjT);                       // ric^T Rj^T
    M3d ARi = mul(A, Ri);
    double jac[3][19];
    M3d ji_r = mul(ARi, neg(skew(pts_imu_i)));                                          // :100-102
    M3d jj_l = neg(A);                                                                  // :113
    M3d jj_r = mul(ricT, skew(pts_imu_j));                                              // :114
    // :143-147 extrinsic
    M3d I3 = identity3<double>();
    M3d RjTRi = mul(RjT, Ri);
    M3d tmpm; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) tmpm.m[i][j] = RjTRi.m[i][j] - I3.m[i][j];
    M3d jex_l = mul(ricT, tmpm);
    M3d tmp_r = mul(ARi, ric);
    V3d t1 = mul(tmp_r, pts_camera_i);
    V3d inner = mul(RjT, mul(Ri, tic) + Pi - Pj) - tic;
    V3d t2 = mul(ricT, inner);
    M3d a1 = mul(tmp_r, skew(pts_camera_i)), a2 = skew(t1), a3 = skew(t2);
    // :166 feature
    V3d jf = mul(tmp_r, pts_i);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) {
            jac[i][j] = A.m[i][j]; jac[i][3 + j] = ji_r.m[i][j];
            jac[i][6 + j] = jj_l.m[i][j]; jac[i][9 + j] = jj_r.m[i][j];
            jac[i][12 + j] = jex_l.m[i][j]; jac[i][15 + j] = -a1.m[i][j] + a2.m[i][j] + a3.m[i][j];
        }
    }
    jac[0][18] = jf.x * -1.0 / (inv_dep * inv_dep);
    jac[1][18] = jf.y * -1.0 / (inv_dep * inv_dep);
    jac[2][18] = jf.z * -1.0 / (inv_dep * inv_dep);
    for (int i = 0; i < 2; ++i) for (int c = 0; c < 19; ++c)
        J[i * 19 + c] = reduce[i][0] * jac[0][c] + reduce[i][1] * jac[1][c] + reduce[i][2] * jac[2][c];
}

// ProjectionTdFactor::Evaluate, projection_td_factor.cpp:34-145: the a5 projection of the TIME-SHIFTED observations
//   pts_i_td = pts_i - (td - td_i) * (vel_i, 0),  pts_j_td = pts_j - (td - td_j) * (vel_j, 0)            (:51-52; the rolling-shutter term
// TR / ROW * row is folded into td_i / td_j by the caller, include/uvs_solver.h) plus a fifth 1-dof block td (:135-140):
//   d r / d td = reduce * ric^T Rj^T Ri ric * vel_i * (-1 / inv_dep) + sqrt_info * vel_j.xy.
// J is 2 x 20 row-major: the 19 columns of point_eval at the shifted observations, then the td column.
inline void point_td_eval(const double* pose_i, const double* pose_j, const double* ex, double inv_dep, const double* pi3, const double* pj3,
                          const double* vel_i2, const double* vel_j2, double td_i, double td_j, double td, double sqrt_info, double* r, double* J) {
    const double pi_td[3] = {pi3[0] - (td - td_i) * vel_i2[0], pi3[1] - (td - td_i) * vel_i2[1], pi3[2]};
    const double pj_td[3] = {pj3[0] - (td - td_j) * vel_j2[0], pj3[1] - (td - td_j) * vel_j2[1], pj3[2]};
    double J19[38];
    point_eval(pose_i, pose_j, ex, inv_dep, pi_td, pj_td, sqrt_info, r, J ? J19 : nullptr);
    if (!J) return;
    for (int i = 0; i < 2; ++i) for (int c = 0; c < 19; ++c) J[i * 20 + c] = J19[i * 19 + c];
    // column 18 of point_eval is reduce * tmp_r * pts_i_td * (-1 / inv_dep^2) with tmp_r = ric^T Rj^T Ri ric (:166 / td :131); the td column needs
    // reduce * tmp_r * vel_i * (-1 / inv_dep): same linear map applied to (vel_i, 0), so evaluate it with a unit-depth trick:
    //   reduce * tmp_r * v = -inv_dep^2 * (column 18 of a point_eval whose pts_i is v)   -- but reduce depends on pts_i, so compute it directly.
    Qd Qi = quat_xyzw(pose_i + 3), Qj = quat_xyzw(pose_j + 3), qic = quat_xyzw(ex + 3);
    V3d Pi = v3(pose_i), Pj = v3(pose_j), tic = v3(ex);
    V3d pts_camera_i = {pi_td[0] / inv_dep, pi_td[1] / inv_dep, pi_td[2] / inv_dep};
    V3d pts_camera_j = qrot(qinv(qic), qrot(qinv(Qj), qrot(Qi, qrot(qic, pts_camera_i) + tic) + Pi - Pj) - tic);
    const double dep_j = pts_camera_j.z;
    const double reduce[2][3] = {{sqrt_info / dep_j, 0.0, -sqrt_info * pts_camera_j.x / (dep_j * dep_j)},
                                 {0.0, sqrt_info / dep_j, -sqrt_info * pts_camera_j.y / (dep_j * dep_j)}};
    M3d tmp_r = mul(mul(mul(transpose(qmat(qic)), transpose(qmat(Qj))), qmat(Qi)), qmat(qic));
    V3d vi = {vel_i2[0], vel_i2[1], 0.0};
    V3d tv = mul(tmp_r, vi);
    for (int i = 0; i < 2; ++i)
        J[i * 20 + 19] = (reduce[i][0] * tv.x + reduce[i][1] * tv.y + reduce[i][2] * tv.z) * (-1.0 / inv_dep) + sqrt_info * vel_j2[i];
}

// ---- shared front part of a7/a8 (line_projection_factor.h:21-54, vp_projection_factor.h:24-57)
template <typename T>
inline void line_to_camera(const T* pose, const T* line, const M3d& ric_d, const V3d& tic_d, V3<T>* n_c, V3<T>* d_c) {
    const V3<T> t_wb = {pose[0], pose[1], pose[2]};
    const Quat<T> q_wb = {pose[6], pose[3], pose[4], pose[5]};
    const Quat<T> roll = qaxis<T>(line[0], 0), pitch = qaxis<T>(line[1], 1), yaw = qaxis<T>(line[2], 2);
    const T phi = line[3];
    M3<T> R_wc = mul(qmat(q_wb), castm<T>(ric_d));                 // q_wb * ric   (:28)
    V3<T> t_wc = qrot(q_wb, cast3<T>(tic_d)) + t_wb;               // (:29)
    M3<T> Rpsi = qmat(qmul(qmul(roll, pitch), yaw));               // roll*pitch*yaw (:31)
    T cphi = jcos(phi), sphi = jsin(phi);
    V3<T> n_w = {cphi * Rpsi.m[0][0], cphi * Rpsi.m[1][0], cphi * Rpsi.m[2][0]};   // (:33)
    V3<T> d_w = {sphi * Rpsi.m[0][1], sphi * Rpsi.m[1][1], sphi * Rpsi.m[2][1]};   // (:34)
    M3<T> RT = transpose(R_wc);
    V3<T> t_cw = -mul(RT, t_wc);                                   // (:41)
    M3<T> tss = skew(t_cw);                                        // (:42-45)
    M3<T> tssRT = mul(tss, RT);                                    // (:49)
    *n_c = mul(RT, n_w) + mul(tssRT, d_w);                         // l_c = T_cw * l_w (:52-53)
    *d_c = mul(RT, d_w);                                           // (:54)
}

// ---- a7: LineProjectionFactor::operator() (line_projection_factor.h:16-60)
template <typename T>
inline void line_functor(const T* pose, const T* line, const M3d& ric, const V3d& tic, const double* sp, const double* ep,
                         double line_factor, T* res) {
    V3<T> n_c, d_c;
    line_to_camera<T>(pose, line, ric, tic, &n_c, &d_c);
    V3<T> sps = {T(sp[0]), T(sp[1]), T(sp[2])}, eps = {T(ep[0]), T(ep[1]), T(ep[2])};
    res[0] = T(line_factor) * dot(sps, n_c) / jsqrt(jpow2(n_c.x) + jpow2(n_c.y));   // :56
    res[1] = T(line_factor) * dot(eps, n_c) / jsqrt(jpow2(n_c.x) + jpow2(n_c.y));   // :57
}

// ---- a8: VPProjectionFactor::operator() (vp_projection_factor.h:19-66)
template <typename T>
inline void vp_functor(const T* pose, const T* line, const M3d& ric, const V3d& tic, const double* vp,
                       double vp_factor, T* res, T* cosabs) {
    V3<T> n_c, d_c;
    line_to_camera<T>(pose, line, ric, tic, &n_c, &d_c);
    V3<T> vp3 = {T(vp[0]), T(vp[1]), T(vp[2])};
    T c = jabs(dot(d_c, vp3) / (norm(d_c) * norm(vp3)));          // :61
    *cosabs = c;
    res[0] = T(vp_factor) * jacos(c);
}

// Autodiff wrappers: J is rows x 10 row-major = [pose cols 0..5 of the 7 raw scalars | 4 line scalars]
inline void line_eval(const double* pose, const double* line, const double* ex, const double* sp, const double* ep,
                      double line_factor, double* r, double* J) {
    M3d ric = qmat(quat_xyzw(ex + 3)); V3d tic = v3(ex);
    if (!J) { line_functor<double>(pose, line, ric, tic, sp, ep, line_factor, r); return; }
    typedef Jet<11> JT;
    JT p[7], l[4], res[2];
    for (int i = 0; i < 7; ++i) p[i] = JT(pose[i], i);
    for (int i = 0; i < 4; ++i) l[i] = JT(line[i], 7 + i);
    line_functor<JT>(p, l, ric, tic, sp, ep, line_factor, res);
    for (int k = 0; k < 2; ++k) {
        r[k] = res[k].a;
        for (int c = 0; c < 6; ++c) J[k * 10 + c] = res[k].v[c];            // leftCols(6): [I6;0] local Jacobian
        for (int c = 0; c < 4; ++c) J[k * 10 + 6 + c] = res[k].v[7 + c];
    }
}
inline void vp_eval(const double* pose, const double* line, const double* ex, const double* vp,
                    double vp_factor, double* r, double* J) {
    M3d ric = qmat(quat_xyzw(ex + 3)); V3d tic = v3(ex);
    double c0;
    if (!J) {
        vp_functor<double>(pose, line, ric, tic, vp, vp_factor, r, &c0);
        if (!(c0 < 1.0)) r[0] = vp_factor * std::acos(std::fmin(c0, 1.0));
        return;
    }
    typedef Jet<11> JT;
    JT p[7], l[4], res[1], cj;
    for (int i = 0; i < 7; ++i) p[i] = JT(pose[i], i);
    for (int i = 0; i < 4; ++i) l[i] = JT(line[i], 7 + i);
    vp_functor<JT>(p, l, ric, tic, vp, vp_factor, res, &cj);
    if (1.0 - cj.a * cj.a <= kVpSin2Guard) {       // documented deviation D8
        r[0] = vp_factor * std::acos(std::fmin(cj.a, 1.0));
        for (int c = 0; c < 10; ++c) J[c] = 0.0;
        return;
    }
    r[0] = res[0].a;
    for (int c = 0; c < 6; ++c) J[c] = res[0].v[c];
    for (int c = 0; c < 4; ++c) J[6 + c] = res[0].v[7 + c];
}

// ---- small dense helpers for the IMU information matrix
// Eigen Matrix<15,15>::inverse() = PartialPivLU; LLT(...).matrixL().transpose()  (imu_factor.h:64)
inline bool inverse_lu(int n, const double* A, double* Ainv) {
    double M[15 * 30];
    if (n > 15) return false;
    for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) { M[i * 2 * n + j] = A[i * n + j]; M[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0; } }
    for (int k = 0; k < n; ++k) {
        int piv = k; double best = std::fabs(M[k * 2 * n + k]);
        for (int i = k + 1; i < n; ++i) { double v = std::fabs(M[i * 2 * n + k]); if (v > best) { best = v; piv = i; } }
        if (best == 0.0) return false;
        if (piv != k) for (int j = 0; j < 2 * n; ++j) { double t = M[k * 2 * n + j]; M[k * 2 * n + j] = M[piv * 2 * n + j]; M[piv * 2 * n + j] = t; }
        const double d = M[k * 2 * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = M[i * 2 * n + k] / d;
            if (f != 0.0) for (int j = k; j < 2 * n; ++j) M[i * 2 * n + j] -= f * M[k * 2 * n + j];
        }
    }
    for (int k = n - 1; k >= 0; --k) {
        const double d = M[k * 2 * n + k];
        for (int j = 0; j < 2 * n; ++j) M[k * 2 * n + j] /= d;
        for (int i = 0; i < k; ++i) { const double f = M[i * 2 * n + k]; if (f != 0.0) for (int j = 0; j < 2 * n; ++j) M[i * 2 * n + j] -= f * M[k * 2 * n + j]; }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ainv[i * n + j] = M[i * 2 * n + n + j];
    return true;
}
// lower Cholesky A = L L^T (row-major, reads lower triangle); returns false if not PD
inline bool chol_lower(int n, const double* A, double* L) {
    for (int i = 0; i < n * n; ++i) L[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0.0)) return false;
        const double ljj = std::sqrt(d);
        L[j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / ljj;
        }
    }
    return true;
}
// sqrt_info (15x15 upper, row-major) with sqrt_info^T sqrt_info = cov^-1
inline bool imu_sqrt_info(const double* cov, double* W) {
    double inv[225], L[225];
    if (!inverse_lu(15, cov, inv)) return false;
    if (!chol_lower(15, inv, L)) return false;
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) W[i * 15 + j] = L[j * 15 + i];
    return true;
}

inline void qleft(const Qd& q, double M[4][4]) {    // utility.h:46-54
    M[0][0] = q.w; M[0][1] = -q.x; M[0][2] = -q.y; M[0][3] = -q.z;
    M[1][0] = q.x; M[2][0] = q.y; M[3][0] = q.z;
    M3d S = skew(V3d{q.x, q.y, q.z});
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[1 + i][1 + j] = (i == j ? q.w : 0.0) + S.m[i][j];
}
inline void qright(const Qd& p, double M[4][4]) {   // utility.h:56-64
    M[0][0] = p.w; M[0][1] = -p.x; M[0][2] = -p.y; M[0][3] = -p.z;
    M[1][0] = p.x; M[2][0] = p.y; M[3][0] = p.z;
    M3d S = skew(V3d{p.x, p.y, p.z});
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[1 + i][1 + j] = (i == j ? p.w : 0.0) - S.m[i][j];
}

// ---- a4: IMUFactor::Evaluate (imu_factor.h:19-182) + IntegrationBase::evaluate (integration_base.h:160-186)
// W = sqrt_info (precomputed by imu_sqrt_info: numerically equivalent to recomputing per call, Appendix D5).
// J layout: 15 x 30 row-major = [pose_i(6) | sb_i(9) | pose_j(6) | sb_j(9)].  whiten=0 returns raw r/J.
inline void imu_eval(const uvs_imu_block& b, const double* W, const double* G3,
                     const double* pose_i, const double* sb_i, const double* pose_j, const double* sb_j,
                     double* r, double* J, int whiten = 1) {
    const int O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12;
    V3d Pi = v3(pose_i), Pj = v3(pose_j), Vi = v3(sb_i), Vj = v3(sb_j);
    V3d Bai = v3(sb_i + 3), Bgi = v3(sb_i + 6), Baj = v3(sb_j + 3), Bgj = v3(sb_j + 6);
    Qd Qi = quat_xyzw(pose_i + 3), Qj = quat_xyzw(pose_j + 3);
    V3d G = v3(G3);
    const double sum_dt = b.sum_dt;
    auto blk = [&](int r0, int c0) { M3d M; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M.m[i][j] = b.jacobian[(r0 + i) * 15 + c0 + j]; return M; };
    M3d dp_dba = blk(O_P, O_BA), dp_dbg = blk(O_P, O_BG), dq_dbg = blk(O_R, O_BG), dv_dba = blk(O_V, O_BA), dv_dbg = blk(O_V, O_BG);
    V3d lin_ba = v3(b.linearized_ba), lin_bg = v3(b.linearized_bg);
    V3d dba = Bai - lin_ba, dbg = Bgi - lin_bg;
    Qd delta_q = quat_xyzw(b.delta_q);
    V3d delta_p = v3(b.delta_p), delta_v = v3(b.delta_v);
    V3d th = mul(dq_dbg, dbg);
    Qd dq = {1.0, th.x / 2.0, th.y / 2.0, th.z / 2.0};
    Qd corrected_delta_q = qmul(delta_q, dq);                                           // integration_base.h:173
    V3d corrected_delta_v = delta_v + mul(dv_dba, dba) + mul(dv_dbg, dbg);              // :174
    V3d corrected_delta_p = delta_p + mul(dp_dba, dba) + mul(dp_dbg, dbg);              // :175
    Qd Qi_inv = qinv(Qi);
    V3d a_p = G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt;
    V3d a_v = G * sum_dt + Vj - Vi;
    V3d rp = qrot(Qi_inv, a_p) - corrected_delta_p;                                      // :177
    Qd qe = qmul(qinv(corrected_delta_q), qmul(Qi_inv, Qj));
    V3d rq = {2.0 * qe.x, 2.0 * qe.y, 2.0 * qe.z};                                       // :178
    V3d rv = qrot(Qi_inv, a_v) - corrected_delta_v;                                      // :179
    V3d rba = Baj - Bai, rbg = Bgj - Bgi;                                                // :180-181
    double raw[15] = {rp.x, rp.y, rp.z, rq.x, rq.y, rq.z, rv.x, rv.y, rv.z, rba.x, rba.y, rba.z, rbg.x, rbg.y, rbg.z};
    if (whiten) { for (int i = 0; i < 15; ++i) { double s = 0.0; for (int k = 0; k < 15; ++k) s += W[i * 15 + k] * raw[k]; r[i] = s; } }
    else for (int i = 0; i < 15; ++i) r[i] = raw[i];
    if (!J) return;
    double Jr[15][30];
    std::memset(Jr, 0, sizeof(Jr));
    auto put = [&](int r0, int c0, const M3d& M, double s) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Jr[r0 + i][c0 + j] = s * M.m[i][j]; };
    M3d RiT = qmat(Qi_inv);                                                               // Qi.inverse().toRotationMatrix()
    M3d I3 = identity3<double>();
    double Ml[4][4], Mr[4][4];
    auto br3 = [&](double A[4][4], double B[4][4]) { M3d M; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0.0; for (int k = 0; k < 4; ++k) s += A[1 + i][k] * B[k][1 + j]; M.m[i][j] = s; } return M; };
    // pose_i (cols 0..5)  imu_factor.h:94-104
    put(O_P, 0 + O_P, RiT, -1.0);
    put(O_P, 0 + O_R, skew(qrot(Qi_inv, a_p)), 1.0);
    qleft(qmul(qinv(Qj), Qi), Ml); qright(corrected_delta_q, Mr);
    put(O_R, 0 + O_R, br3(Ml, Mr), -1.0);
    put(O_V, 0 + O_R, skew(qrot(Qi_inv, a_v)), 1.0);
    // speedbias_i (cols 6..14)  :119-137 ; O_x - O_V offsets
    const int sbi = 6;
    put(O_P, sbi + 0, RiT, -sum_dt);
    put(O_P, sbi + 3, dp_dba, -1.0);
    put(O_P, sbi + 6, dp_dbg, -1.0);
    {
        qleft(qmul(qmul(qinv(Qj), Qi), delta_q), Ml);                                     // :128 uses un-corrected delta_q
        M3d L3; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) L3.m[i][j] = Ml[1 + i][1 + j];
        put(O_R, sbi + 6, mul(L3, dq_dbg), -1.0);
    }
    put(O_V, sbi + 0, RiT, -1.0);
    put(O_V, sbi + 3, dv_dba, -1.0);
    put(O_V, sbi + 6, dv_dbg, -1.0);
    put(O_BA, sbi + 3, I3, -1.0);
    put(O_BG, sbi + 6, I3, -1.0);
    // pose_j (cols 15..20)  :149-155
    const int pj = 15;
    put(O_P, pj + O_P, RiT, 1.0);
    {
        qleft(qmul(qmul(qinv(corrected_delta_q), Qi_inv), Qj), Ml);
        M3d L3; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) L3.m[i][j] = Ml[1 + i][1 + j];
        put(O_R, pj + O_R, L3, 1.0);
    }
    // speedbias_j (cols 21..29)  :168-172
    const int sbj = 21;
    put(O_V, sbj + 0, RiT, 1.0);
    put(O_BA, sbj + 3, I3, 1.0);
    put(O_BG, sbj + 6, I3, 1.0);
    if (whiten) {
        for (int i = 0; i < 15; ++i) for (int c = 0; c < 30; ++c) { double s = 0.0; for (int k = 0; k < 15; ++k) s += W[i * 15 + k] * Jr[k][c]; J[i * 30 + c] = s; }
    } else {
        for (int i = 0; i < 15; ++i) for (int c = 0; c < 30; ++c) J[i * 30 + c] = Jr[i][c];
    }
}

// ---- a9: MarginalizationFactor::Evaluate (marginalization_factor.cpp:333-381)
// dx (n) from the kept blocks; r = r0 + J0*dx.  block accessor returns the current global values.
template <typename GetBlock>
inline void prior_eval(const uvs_prior& p, GetBlock get, double* dx, double* r) {
    const int n = p.n;
    for (int i = 0; i < n; ++i) dx[i] = 0.0;
    for (int b = 0; b < p.n_blocks; ++b) {
        const int size = p.block_size[b], idx = p.block_idx[b];
        const double* x = get(p.block_kind[b], p.block_frame[b]);
        const double* x0 = p.x0 + p.x0_off[b];
        if (size != 7) { for (int k = 0; k < size; ++k) dx[idx + k] = x[k] - x0[k]; }
        else {
            for (int k = 0; k < 3; ++k) dx[idx + k] = x[k] - x0[k];
            Qd q0 = quat_xyzw(x0 + 3), q = quat_xyzw(x + 3);
            Qd e = qmul(qinv(q0), q);                                                    // :356
            double sgn = (e.w >= 0.0) ? 1.0 : -1.0;                                       // :357-360
            dx[idx + 3] = 2.0 * sgn * e.x; dx[idx + 4] = 2.0 * sgn * e.y; dx[idx + 5] = 2.0 * sgn * e.z;
        }
    }
    for (int i = 0; i < n; ++i) { double s = p.linearized_residuals[i]; for (int k = 0; k < n; ++k) s += p.linearized_jacobians[i * n + k] * dx[k]; r[i] = s; }   // :364
}

}  // namespace orc
