// uvs_factors.h -- device-side residual / Jacobian evaluation for the sliding-window solve.
//
// Hand-written FP64 HIP for gfx950.  One lane evaluates one residual block; the
// per-frame rotation matrices and the camera extrinsic are staged in LDS by the
// caller and passed in by pointer.  Every function cites the reference code it
// replaces; the line / vanishing-point Jacobians are HAND-DERIVED (the reference
// uses ceres::AutoDiffCostFunction) and reproduce the reference's convention of
// differentiating w.r.t. the raw quaternion scalars (qx,qy,qz) with qw fixed
// (SURVEY.md Appendix D1), not the tangent-space derivative.
#pragma once
#include <hip/hip_runtime.h>

namespace uvsdev {

#define UVS_DEV __device__ __forceinline__

// sin^2 guard of the VP residual (documented deviation)
static constexpr double kVpSin2Guard = 1e-14;

// ---------------------------------------------------------------- small vector helpers (row-major 3x3)
UVS_DEV void quat_to_R(const double* q /*x,y,z,w*/, double* R) {   // Eigen toRotationMatrix polynomial
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
UVS_DEV void mat_vec(const double* R, const double* v, double* o) {
    o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    o[1] = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    o[2] = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
}
UVS_DEV void matT_vec(const double* R, const double* v, double* o) {
    o[0] = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    o[1] = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    o[2] = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
}
UVS_DEV void mat_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
UVS_DEV void matT_mul(const double* A, const double* B, double* C) {   // A^T B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
UVS_DEV void mat_mulT(const double* A, const double* B, double* C) {   // A B^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[3 * j] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
UVS_DEV void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
UVS_DEV double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
// M * [v]x  (columns: M * (e_c x ... ) )  -> out = M * skew(v)
UVS_DEV void mat_skew(const double* M, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = M[3 * i], b = M[3 * i + 1], c = M[3 * i + 2];
        o[3 * i + 0] = b * v[2] - c * v[1];
        o[3 * i + 1] = c * v[0] - a * v[2];
        o[3 * i + 2] = a * v[1] - b * v[0];
    }
}
// [v]x * M
UVS_DEV void skew_mat(const double* v, const double* M, double* o) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double a = M[j], b = M[3 + j], c = M[6 + j];
        o[j] = v[1] * c - v[2] * b;
        o[3 + j] = v[2] * a - v[0] * c;
        o[6 + j] = v[0] * b - v[1] * a;
    }
}
UVS_DEV void quat_mul(const double* a, const double* b, double* o) {   // (x,y,z,w) storage, Eigen product
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3], bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
}
UVS_DEV void quat_inv(const double* q, double* o) {   // Eigen inverse(): conjugate / squaredNorm
    const double n2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    const double in2 = 1.0 / n2;
    o[0] = -q[0] * in2; o[1] = -q[1] * in2; o[2] = -q[2] * in2; o[3] = q[3] * in2;
}
UVS_DEV void quat_rot(const double* q, const double* v, double* o) {   // Eigen _transformVector
    double uv[3]; cross3(q, v, uv);
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    double c[3]; cross3(q, uv, c);
    o[0] = v[0] + q[3] * uv[0] + c[0]; o[1] = v[1] + q[3] * uv[1] + c[1]; o[2] = v[2] + q[3] * uv[2] + c[2];
}

// ---------------------------------------------------------------- a10: Cauchy loss
// Cauchy(a): rho'' < 0 always => the Ceres corrector reduces to scaling r and J by sqrt(rho')
// (marginalization_factor.cpp:47-67, first branch).  Returns rho(s); *scale = sqrt(rho'(s)).
UVS_DEV double cauchy(double a, double sq_norm, double* scale) {
    const double b = a * a, c = (a == 1.0) ? 1.0 : 1.0 / b;      // (the point and VP losses have a = 1: no division)
    const double sum = 1.0 + sq_norm * c;
    *scale = rsqrt(fmin(sum, 1.7976931348623157e308));      // sqrt(rho') = sqrt(1 / sum): one reciprocal square root instead of a division and a square root
    return b * log(sum);
}

// ---------------------------------------------------------------- a5: point reprojection
// Replaces ProjectionFactor::Evaluate (projection_factor.cpp:22-175).
// Ri,Rj: frame rotations (row-major), Pi,Pj positions, ric/tic extrinsic.  Outputs (un-robustified):
//   r[2], Ji[12] (2x6 wrt pose_i), Jj[12] (2x6 wrt pose_j), Jl[2] (wrt inverse depth), Jex[12] if WITH_EX.
template <bool WITH_J, bool WITH_EX>
UVS_DEV void point_eval(const double* Pi, const double* Ri, const double* Pj, const double* Rj, const double* ric, const double* tic,
                        double inv_dep, const double* pts_i, const double* pts_j, double sqrt_info,
                        double* r, double* Ji, double* Jj, double* Jl, double* Jex,
                        const double* vel_i = nullptr, const double* vel_j = nullptr, double* Jtd = nullptr) {
    const double dep = 1.0 / inv_dep;      // one reciprocal instead of three divisions (an FP64 division is ~30 instructions; this runs per observation and pass)
    double pc_i[3] = {pts_i[0] * dep, pts_i[1] * dep, pts_i[2] * dep};      // :44
    double p_imu_i[3]; mat_vec(ric, pc_i, p_imu_i);
    p_imu_i[0] += tic[0]; p_imu_i[1] += tic[1]; p_imu_i[2] += tic[2];                    // :45
    double pw[3]; mat_vec(Ri, p_imu_i, pw);
    pw[0] += Pi[0] - Pj[0]; pw[1] += Pi[1] - Pj[1]; pw[2] += Pi[2] - Pj[2];              // :46 (minus Pj of :47)
    double p_imu_j[3]; matT_vec(Rj, pw, p_imu_j);                                       // :47
    double d[3] = {p_imu_j[0] - tic[0], p_imu_j[1] - tic[1], p_imu_j[2] - tic[2]};
    double pc_j[3]; matT_vec(ric, d, pc_j);                                             // :48
    const double inv_z = 1.0 / pc_j[2];
    r[0] = sqrt_info * (pc_j[0] * inv_z - pts_j[0]);                                    // :57,:67
    r[1] = sqrt_info * (pc_j[1] * inv_z - pts_j[1]);
    if (!WITH_J) return;
    // reduce (2x3) = sqrt_info * [1/z 0 -x/z^2 ; 0 1/z -y/z^2]   :90-93
    const double r00 = sqrt_info * inv_z, r02 = -sqrt_info * pc_j[0] * inv_z * inv_z, r12 = -sqrt_info * pc_j[1] * inv_z * inv_z;
    double A[9];  { double t[9]; mat_mulT(ric, Rj, t); /* ric * Rj^T ... need ric^T Rj^T */
                    // ric^T * Rj^T = (Rj * ric)^T
                    double Rr[9]; mat_mul(Rj, ric, Rr);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) A[3 * i + j] = Rr[3 * j + i];
                    (void)t; }
    double ARi[9]; mat_mul(A, Ri, ARi);
    double m[9];
    // pose_i: [A | ARi * (-skew(p_imu_i))]   :100-102
    mat_skew(ARi, p_imu_i, m);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Ji[c] = r00 * A[c] + r02 * A[6 + c];           Ji[6 + c] = r00 * A[3 + c] + r12 * A[6 + c];
        Ji[3 + c] = -(r00 * m[c] + r02 * m[6 + c]);    Ji[9 + c] = -(r00 * m[3 + c] + r12 * m[6 + c]);
    }
    // pose_j: [-A | ric^T skew(p_imu_j)]   :113-114
    { double ricT[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) ricT[3 * i + j] = ric[3 * j + i];
      mat_skew(ricT, p_imu_j, m); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Jj[c] = -(r00 * A[c] + r02 * A[6 + c]);        Jj[6 + c] = -(r00 * A[3 + c] + r12 * A[6 + c]);
        Jj[3 + c] = r00 * m[c] + r02 * m[6 + c];       Jj[9 + c] = r00 * m[3 + c] + r12 * m[6 + c];
    }
    // inverse depth: reduce * (ARi*ric) * pts_i * (-1/lambda^2)   :166
    double T[9]; mat_mul(ARi, ric, T);
    { double v[3]; mat_vec(T, pts_i, v);
      const double s = -(dep * dep);
      Jl[0] = (r00 * v[0] + r02 * v[2]) * s; Jl[1] = (r00 * v[1] + r12 * v[2]) * s; }
    if (Jtd) {   // ProjectionTdFactor, projection_td_factor.cpp:135-140: reduce * tmp_r * (vel_i, 0) * (-1 / inv_dep) + sqrt_info * vel_j.xy (pts_i / pts_j are the shifted ones)
        const double v0 = T[0] * vel_i[0] + T[1] * vel_i[1], v1 = T[3] * vel_i[0] + T[4] * vel_i[1], v2 = T[6] * vel_i[0] + T[7] * vel_i[1];
        const double s = -dep;
        Jtd[0] = (r00 * v0 + r02 * v2) * s + sqrt_info * vel_j[0];
        Jtd[1] = (r00 * v1 + r12 * v2) * s + sqrt_info * vel_j[1];
    }
    if (WITH_EX) {   // :143-147
        double ricT[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) ricT[3 * i + j] = ric[3 * j + i];
        double RjTRi[9]; matT_mul(Rj, Ri, RjTRi);
        RjTRi[0] -= 1.0; RjTRi[4] -= 1.0; RjTRi[8] -= 1.0;
        double L[9]; mat_mul(ricT, RjTRi, L);
        double a1[9]; mat_skew(T, pc_i, a1);            // tmp_r * skew(pts_camera_i)
        double t1[3]; mat_vec(T, pc_i, t1);             // skew(tmp_r * pts_camera_i)
        double in[3]; { double u[3]; mat_vec(Ri, tic, u); u[0] += Pi[0] - Pj[0]; u[1] += Pi[1] - Pj[1]; u[2] += Pi[2] - Pj[2];
                        matT_vec(Rj, u, in); in[0] -= tic[0]; in[1] -= tic[1]; in[2] -= tic[2]; }
        double t2[3]; mat_vec(ricT, in, t2);
        const double s[3] = {t1[0] + t2[0], t1[1] + t2[1], t1[2] + t2[2]};
        double Rm[9] = {-a1[0], -a1[1] - s[2], -a1[2] + s[1],
                        -a1[3] + s[2], -a1[4], -a1[5] - s[0],
                        -a1[6] - s[1], -a1[7] + s[0], -a1[8]};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Jex[c] = r00 * L[c] + r02 * L[6 + c];          Jex[6 + c] = r00 * L[3 + c] + r12 * L[6 + c];
            Jex[3 + c] = r00 * Rm[c] + r02 * Rm[6 + c];    Jex[9 + c] = r00 * Rm[3 + c] + r12 * Rm[6 + c];
        }
    }
}

// ---------------------------------------------------------------- a7 + a8: Pluecker line reprojection and VP angle
// Replaces LineProjectionFactor::operator() (line_projection_factor.h:16-60) and
// VPProjectionFactor::operator() (vp_projection_factor.h:19-66) and their Ceres autodiff.
// Derivatives are w.r.t. (t[3], qx,qy,qz | qw fixed) and the 4 additive line scalars.
//   pose = (t, q) raw 7 scalars; R = R(q) passed in (polynomial form).
// Outputs (un-robustified): rl[2], Jlp[12] (2x6), Jll[8] (2x4); if has_vp: rv, Jvp[6], Jvl[4].
struct LineGeom {
    double n_c[3], d_c[3];
    double dn_dt[9], dn_du[9], dd_du[9];   // 3x3 each, row-major; dd_dt = 0
    double dn_dl[12], dd_dl[12];           // 3x4 each
};

UVS_DEV void line_trig(const double* line, double* tr) {      // (sa, ca, sb, cb, sc, cc, sp, cp) of the orthonormal line parameters
    sincos(line[0], &tr[0], &tr[1]); sincos(line[1], &tr[2], &tr[3]); sincos(line[2], &tr[4], &tr[5]); sincos(line[3], &tr[6], &tr[7]);
}
// `trig` (optional) = line_trig(line) computed once per parameter update: a line is observed from ~7 frames and evaluated ~22 times per
// solve, and four double-precision sincos are most of the arithmetic of one observation
template <bool WITH_J>
UVS_DEV void line_geom(const double* t, const double* q, const double* R, const double* ric, const double* tic, const double* line, LineGeom& g, const double* trig = nullptr) {
    // U = Rx(psi_x) Ry(psi_y) Rz(psi_z)   (:23-31)
    double sa, ca, sb, cb, sc, cc, sp, cp;
    if (trig) { sa = trig[0]; ca = trig[1]; sb = trig[2]; cb = trig[3]; sc = trig[4]; cc = trig[5]; sp = trig[6]; cp = trig[7]; }
    else { sincos(line[0], &sa, &ca); sincos(line[1], &sb, &cb); sincos(line[2], &sc, &cc); sincos(line[3], &sp, &cp); }
    const double U0[3] = {cb * cc, sa * sb * cc + ca * sc, -ca * sb * cc + sa * sc};
    const double U1[3] = {-cb * sc, -sa * sb * sc + ca * cc, ca * sb * sc + sa * cc};
    const double n_w[3] = {cp * U0[0], cp * U0[1], cp * U0[2]};        // :33
    const double d_w[3] = {sp * U1[0], sp * U1[1], sp * U1[2]};        // :34
    double t_wc[3]; mat_vec(R, tic, t_wc); t_wc[0] += t[0]; t_wc[1] += t[1]; t_wc[2] += t[2];   // :29
    double m[3], np[3], dp[3];
    matT_vec(R, t_wc, m); matT_vec(R, n_w, np); matT_vec(R, d_w, dp);
    double Am[3], An[3];
    matT_vec(ric, m, Am); matT_vec(ric, np, An); matT_vec(ric, dp, g.d_c);          // A = ric^T
    const double t_cw[3] = {-Am[0], -Am[1], -Am[2]};                                 // :41
    double x[3]; cross3(t_cw, g.d_c, x);
    g.n_c[0] = An[0] + x[0]; g.n_c[1] = An[1] + x[1]; g.n_c[2] = An[2] + x[2];       // :52-53
    if (!WITH_J) return;
    // M = A R^T = (R ric)^T
    double M[9]; { double Rr[9]; mat_mul(R, ric, Rr);
#pragma unroll
                   for (int i = 0; i < 3; ++i)
#pragma unroll
                       for (int j = 0; j < 3; ++j) M[3 * i + j] = Rr[3 * j + i]; }
    // d n_c / d t = [d_c]x M
    skew_mat(g.d_c, M, g.dn_dt);
    // raw-quaternion derivative helpers: Gm(a) = d(R^T a)/du, Gp(a) = d(R a)/du, u = (qx,qy,qz), w = qw fixed
    const double u[3] = {q[0], q[1], q[2]}; const double w = q[3];
    auto Gfun = [&](const double* a, double sgn, double* G) {   // sgn=+1 -> Gm, -1 -> Gp
        const double ua = dot3(u, a);
        const double k = 2.0 * w * sgn;
        // 2w*sgn*[a]x + 2[(u.a)I + u a^T - 2 a u^T]
        G[0] = 2.0 * (ua + u[0] * a[0] - 2.0 * a[0] * u[0]);
        G[1] = k * (-a[2]) + 2.0 * (u[0] * a[1] - 2.0 * a[0] * u[1]);
        G[2] = k * (a[1]) + 2.0 * (u[0] * a[2] - 2.0 * a[0] * u[2]);
        G[3] = k * (a[2]) + 2.0 * (u[1] * a[0] - 2.0 * a[1] * u[0]);
        G[4] = 2.0 * (ua + u[1] * a[1] - 2.0 * a[1] * u[1]);
        G[5] = k * (-a[0]) + 2.0 * (u[1] * a[2] - 2.0 * a[1] * u[2]);
        G[6] = k * (-a[1]) + 2.0 * (u[2] * a[0] - 2.0 * a[2] * u[0]);
        G[7] = k * (a[0]) + 2.0 * (u[2] * a[1] - 2.0 * a[2] * u[1]);
        G[8] = 2.0 * (ua + u[2] * a[2] - 2.0 * a[2] * u[2]);
    };
    double G1[9], G2[9], T1[9], T2[9];
    // dm/du = Gm(t_wc) + R^T Gp(tic)
    Gfun(t_wc, 1.0, G1); Gfun(tic, -1.0, G2); matT_mul(R, G2, T1);
#pragma unroll
    for (int i = 0; i < 9; ++i) G1[i] += T1[i];
    double Adm[9]; matT_mul(ric, G1, Adm);                  // A dm/du ;  dt_cw/du = -Adm
    // dd_c/du = A Gm(d_w)
    Gfun(d_w, 1.0, G2); matT_mul(ric, G2, g.dd_du);
    // dn_c/du = A Gm(n_w) + [d_c]x (A dm/du) + [t_cw]x dd_c/du
    Gfun(n_w, 1.0, G2); matT_mul(ric, G2, T1);
    skew_mat(g.d_c, Adm, T2);
    double T3[9]; skew_mat(t_cw, g.dd_du, T3);
#pragma unroll
    for (int i = 0; i < 9; ++i) g.dn_du[i] = T1[i] + T2[i] + T3[i];
    // line parameters: dn_w, dd_w columns then push through M and [t_cw]x M
    double tM[9]; skew_mat(t_cw, M, tM);
    const double ry[3] = {0.0, ca, sa};                      // Rx * e_y
    double dU0[3][3], dU1[3][3];
    // psi_x: e_x x U_col
    dU0[0][0] = 0.0; dU0[0][1] = -U0[2]; dU0[0][2] = U0[1];
    dU1[0][0] = 0.0; dU1[0][1] = -U1[2]; dU1[0][2] = U1[1];
    // psi_y: (Rx e_y) x U_col
    cross3(ry, U0, dU0[1]); cross3(ry, U1, dU1[1]);
    // psi_z: dU0 = U1, dU1 = -U0
    dU0[2][0] = U1[0]; dU0[2][1] = U1[1]; dU0[2][2] = U1[2];
    dU1[2][0] = -U0[0]; dU1[2][1] = -U0[1]; dU1[2][2] = -U0[2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        double dn[3], dd[3];
        if (p < 3) { dn[0] = cp * dU0[p][0]; dn[1] = cp * dU0[p][1]; dn[2] = cp * dU0[p][2];
                     dd[0] = sp * dU1[p][0]; dd[1] = sp * dU1[p][1]; dd[2] = sp * dU1[p][2]; }
        else { dn[0] = -sp * U0[0]; dn[1] = -sp * U0[1]; dn[2] = -sp * U0[2];
               dd[0] = cp * U1[0]; dd[1] = cp * U1[1]; dd[2] = cp * U1[2]; }
        double a[3], b[3], c[3];
        mat_vec(M, dn, a); mat_vec(tM, dd, b); mat_vec(M, dd, c);
        g.dn_dl[0 * 4 + p] = a[0] + b[0]; g.dn_dl[1 * 4 + p] = a[1] + b[1]; g.dn_dl[2 * 4 + p] = a[2] + b[2];
        g.dd_dl[0 * 4 + p] = c[0]; g.dd_dl[1 * 4 + p] = c[1]; g.dd_dl[2 * 4 + p] = c[2];
    }
}

template <bool WITH_J>
UVS_DEV void line_residual(const LineGeom& g, const double* sp, const double* ep, double line_factor, double* r, double* Jp, double* Jl) {
    const double l2 = g.n_c[0] * g.n_c[0] + g.n_c[1] * g.n_c[1];
    const double il = rsqrt(l2);      // 1 / sqrt(n_x^2 + n_y^2)
    const double es = dot3(sp, g.n_c), ee = dot3(ep, g.n_c);
    r[0] = line_factor * es * il;                                                     // :56
    r[1] = line_factor * ee * il;                                                     // :57
    if (!WITH_J) return;
    const double il3 = il * il * il;      // 1 / l^3 without a second division
    double gs[3] = {line_factor * (sp[0] * il - es * g.n_c[0] * il3), line_factor * (sp[1] * il - es * g.n_c[1] * il3), line_factor * sp[2] * il};
    double ge[3] = {line_factor * (ep[0] * il - ee * g.n_c[0] * il3), line_factor * (ep[1] * il - ee * g.n_c[1] * il3), line_factor * ep[2] * il};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Jp[c] = gs[0] * g.dn_dt[c] + gs[1] * g.dn_dt[3 + c] + gs[2] * g.dn_dt[6 + c];
        Jp[3 + c] = gs[0] * g.dn_du[c] + gs[1] * g.dn_du[3 + c] + gs[2] * g.dn_du[6 + c];
        Jp[6 + c] = ge[0] * g.dn_dt[c] + ge[1] * g.dn_dt[3 + c] + ge[2] * g.dn_dt[6 + c];
        Jp[9 + c] = ge[0] * g.dn_du[c] + ge[1] * g.dn_du[3 + c] + ge[2] * g.dn_du[6 + c];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        Jl[p] = gs[0] * g.dn_dl[p] + gs[1] * g.dn_dl[4 + p] + gs[2] * g.dn_dl[8 + p];
        Jl[4 + p] = ge[0] * g.dn_dl[p] + ge[1] * g.dn_dl[4 + p] + ge[2] * g.dn_dl[8 + p];
    }
}

template <bool WITH_J>
UVS_DEV void vp_residual(const LineGeom& g, const double* vp, double vp_factor, double* r, double* Jp, double* Jl) {
    const double dn2 = dot3(g.d_c, g.d_c), vn2 = dot3(vp, vp);
    const double dv = dot3(g.d_c, vp);
    const double i1 = rsqrt(dn2 * vn2);      // 1 / (|d| |v|)
    const double c0 = dv * i1;
    const double c = fabs(c0);                                                        // :61
    const double s2 = 1.0 - c * c;
    if (!(s2 > kVpSin2Guard)) {      // documented deviation D8 (reference yields inf/NaN here)
        r[0] = vp_factor * acos(fmin(c, 1.0));
        if (WITH_J) {
#pragma unroll
            for (int k = 0; k < 6; ++k) Jp[k] = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) Jl[k] = 0.0;
        }
        return;
    }
    r[0] = vp_factor * acos(c);
    if (!WITH_J) return;
    const double k = vp_factor * (-rsqrt(s2)) * (c0 < 0.0 ? -1.0 : 1.0);
    const double i2 = c0 / dn2;      // dv / (|d|^3 |v|)
    const double gd[3] = {k * (vp[0] * i1 - g.d_c[0] * i2), k * (vp[1] * i1 - g.d_c[1] * i2), k * (vp[2] * i1 - g.d_c[2] * i2)};
#pragma unroll
    for (int cidx = 0; cidx < 3; ++cidx) {
        Jp[cidx] = 0.0;   // d d_c / d t = 0
        Jp[3 + cidx] = gd[0] * g.dd_du[cidx] + gd[1] * g.dd_du[3 + cidx] + gd[2] * g.dd_du[6 + cidx];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) Jl[p] = gd[0] * g.dd_dl[p] + gd[1] * g.dd_dl[4 + p] + gd[2] * g.dd_dl[8 + p];
}

// ---------------------------------------------------------------- a4: IMU pre-integration factor
// Replaces IMUFactor::Evaluate (imu_factor.h:19-182) + IntegrationBase::evaluate (integration_base.h:160-186).
// blk: [0]=sum_dt, [1..3]=delta_p, [4..7]=delta_q(x,y,z,w), [8..10]=delta_v, [11..13]=lin_ba, [14..16]=lin_bg; jac: 15x15 row-major.
// Raw (un-whitened) residual r[15]; if Jraw != nullptr the raw 15x30 Jacobian is written (row-major, zero-filled first).
// Jraw (may be NULL) receives the 15 x 30 Jacobian at Jraw[row * LD + col + (col >= 15 ? GAP : 0)]; ZERO = clear the 450 entries first
// (LD = 30, GAP = 0: dense row-major; the solve kernel uses LD = 48, GAP = 1 into a pre-zeroed, frame-padded MFMA operand tile).
// PARTS = bit mask of the Jacobian groups to write (the solve kernel hands the four groups of a block to four different waves; every
// group needs the same short preamble, the compiler drops what a group does not use):
//   1 the five +-R_i^T blocks, 2 the skew blocks + the -jacobian copies + the +-1 entries, 4 blocks (3,3) and (3,12), 8 block (3,18)
template <int LD = 30, int GAP = 0, bool ZERO = true, int PARTS = 15>
UVS_DEV void imu_raw(const double* blk, const double* jac, const double* G, const double* pose_i, const double* sb_i,
                     const double* pose_j, const double* sb_j, double* r, double* Jraw) {
    const double sum_dt = blk[0];
    const double* delta_p = blk + 1; const double* delta_q = blk + 4; const double* delta_v = blk + 8;
    const double* lin_ba = blk + 11; const double* lin_bg = blk + 14;
    const double* Pi = pose_i; const double* Qi = pose_i + 3; const double* Pj = pose_j; const double* Qj = pose_j + 3;
    const double* Vi = sb_i; const double* Bai = sb_i + 3; const double* Bgi = sb_i + 6;
    const double* Vj = sb_j; const double* Baj = sb_j + 3; const double* Bgj = sb_j + 6;
    const double dba[3] = {Bai[0] - lin_ba[0], Bai[1] - lin_ba[1], Bai[2] - lin_ba[2]};
    const double dbg[3] = {Bgi[0] - lin_bg[0], Bgi[1] - lin_bg[1], Bgi[2] - lin_bg[2]};
    auto J3 = [&](int r0, int c0, const double* v, double* o) {
#pragma unroll
        for (int i = 0; i < 3; ++i) o[i] = jac[UVS_IMU_JIDX(r0, c0, i, 0)] * v[0] + jac[UVS_IMU_JIDX(r0, c0, i, 1)] * v[1] + jac[UVS_IMU_JIDX(r0, c0, i, 2)] * v[2];
    };
    double th[3]; J3(3, 12, dbg, th);                                             // dq_dbg * dbg
    const double dq[4] = {th[0] / 2.0, th[1] / 2.0, th[2] / 2.0, 1.0};            // Utility::deltaQ
    double cq[4]; quat_mul(delta_q, dq, cq);                                       // corrected_delta_q  :173
    double t1[3], t2[3];
    J3(6, 9, dba, t1); J3(6, 12, dbg, t2);
    const double cv[3] = {delta_v[0] + t1[0] + t2[0], delta_v[1] + t1[1] + t2[1], delta_v[2] + t1[2] + t2[2]};   // :174
    J3(0, 9, dba, t1); J3(0, 12, dbg, t2);
    const double cp[3] = {delta_p[0] + t1[0] + t2[0], delta_p[1] + t1[1] + t2[1], delta_p[2] + t1[2] + t2[2]};   // :175
    double Qi_inv[4]; quat_inv(Qi, Qi_inv);
    const double ap[3] = {0.5 * G[0] * sum_dt * sum_dt + Pj[0] - Pi[0] - Vi[0] * sum_dt,
                          0.5 * G[1] * sum_dt * sum_dt + Pj[1] - Pi[1] - Vi[1] * sum_dt,
                          0.5 * G[2] * sum_dt * sum_dt + Pj[2] - Pi[2] - Vi[2] * sum_dt};
    const double av[3] = {G[0] * sum_dt + Vj[0] - Vi[0], G[1] * sum_dt + Vj[1] - Vi[1], G[2] * sum_dt + Vj[2] - Vi[2]};
    double rap[3], rav[3]; quat_rot(Qi_inv, ap, rap); quat_rot(Qi_inv, av, rav);
    double qij[4]; quat_mul(Qi_inv, Qj, qij);
    double cq_inv[4]; quat_inv(cq, cq_inv);
    double qe[4]; quat_mul(cq_inv, qij, qe);
    r[0] = rap[0] - cp[0]; r[1] = rap[1] - cp[1]; r[2] = rap[2] - cp[2];           // :177
    r[3] = 2.0 * qe[0]; r[4] = 2.0 * qe[1]; r[5] = 2.0 * qe[2];                   // :178
    r[6] = rav[0] - cv[0]; r[7] = rav[1] - cv[1]; r[8] = rav[2] - cv[2];           // :179
    r[9] = Baj[0] - Bai[0]; r[10] = Baj[1] - Bai[1]; r[11] = Baj[2] - Bai[2];      // :180
    r[12] = Bgj[0] - Bgi[0]; r[13] = Bgj[1] - Bgi[1]; r[14] = Bgj[2] - Bgi[2];     // :181
    if (!Jraw) return;
    if (ZERO) for (int k = 0; k < 450; ++k) Jraw[k] = 0.0;
    double RiT[9]; quat_to_R(Qi_inv, RiT);                                         // Qi.inverse().toRotationMatrix()
    auto cm = [](int col) { return col + (col >= 15 ? GAP : 0); };
    auto put = [&](int r0, int c0, const double* M, double s) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Jraw[(r0 + i) * LD + cm(c0 + j)] = s * M[3 * i + j];
    };
    auto putskew = [&](int r0, int c0, const double* v) {
        Jraw[(r0 + 0) * LD + cm(c0 + 1)] = -v[2]; Jraw[(r0 + 0) * LD + cm(c0 + 2)] = v[1];
        Jraw[(r0 + 1) * LD + cm(c0 + 0)] = v[2];  Jraw[(r0 + 1) * LD + cm(c0 + 2)] = -v[0];
        Jraw[(r0 + 2) * LD + cm(c0 + 0)] = -v[1]; Jraw[(r0 + 2) * LD + cm(c0 + 1)] = v[0];
    };
    // Qleft(a).bottomRight3x3 = a.w I + skew(a.vec) ; Qright(b).bottomRight = b.w I - skew(b.vec)
    // (Qleft(a) Qright(b)).bottomRight3x3 [i][j] = a.v[i]*(-b.v[j]) + sum_k (a.w I + [a.v]x)[i][k] (b.w I - [b.v]x)[k][j]
    auto LRbr = [&](const double* a, const double* b, double* M) {
        double La[9] = {a[3], -a[2], a[1], a[2], a[3], -a[0], -a[1], a[0], a[3]};
        double Rb[9] = {b[3], b[2], -b[1], -b[2], b[3], b[0], b[1], -b[0], b[3]};
        mat_mul(La, Rb, M);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[3 * i + j] -= a[i] * b[j];
    };
    double M[9];
    // pose_i  (imu_factor.h:94-104)
    if (PARTS & 1) put(0, 0, RiT, -1.0);
    if (PARTS & 2) putskew(0, 3, rap);
    if (PARTS & 4) { double qji[4], Qj_inv[4]; quat_inv(Qj, Qj_inv); quat_mul(Qj_inv, Qi, qji);
      LRbr(qji, cq, M); put(3, 3, M, -1.0);
      // speedbias_i O_R/O_BG (:128): -Qleft(Qj^-1 Qi delta_q).bottomRight * dq_dbg
      double qq[4]; quat_mul(qji, delta_q, qq);
      double La[9] = {qq[3], -qq[2], qq[1], qq[2], qq[3], -qq[0], -qq[1], qq[0], qq[3]};
      double D[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) D[3 * i + j] = jac[UVS_IMU_JIDX(3, 12, i, j)];
      mat_mul(La, D, M); put(3, 6 + 6, M, -1.0); }
    if (PARTS & 2) putskew(6, 3, rav);
    // speedbias_i  (:119-137)
    if (PARTS & 1) put(0, 6 + 0, RiT, -sum_dt);
    if (PARTS & 2)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Jraw[(0 + i) * LD + cm(6 + 3 + j)] = -jac[UVS_IMU_JIDX(0, 9, i, j)];     // -dp_dba
            Jraw[(0 + i) * LD + cm(6 + 6 + j)] = -jac[UVS_IMU_JIDX(0, 12, i, j)];    // -dp_dbg
            Jraw[(6 + i) * LD + cm(6 + 3 + j)] = -jac[UVS_IMU_JIDX(6, 9, i, j)];     // -dv_dba
            Jraw[(6 + i) * LD + cm(6 + 6 + j)] = -jac[UVS_IMU_JIDX(6, 12, i, j)];    // -dv_dbg
        }
    if (PARTS & 1) put(6, 6 + 0, RiT, -1.0);
    if (PARTS & 2)
#pragma unroll
    for (int i = 0; i < 3; ++i) { Jraw[(9 + i) * LD + cm(6 + 3 + i)] = -1.0; Jraw[(12 + i) * LD + cm(6 + 6 + i)] = -1.0; }
    // pose_j  (:149-155)
    if (PARTS & 1) put(0, 15 + 0, RiT, 1.0);
    if (PARTS & 8) { double q3[4]; quat_mul(cq_inv, qij, q3);
      double La[9] = {q3[3], -q3[2], q3[1], q3[2], q3[3], -q3[0], -q3[1], q3[0], q3[3]};
      put(3, 15 + 3, La, 1.0); }
    // speedbias_j  (:168-172)
    if (PARTS & 1) put(6, 21 + 0, RiT, 1.0);
    if (PARTS & 2)
#pragma unroll
    for (int i = 0; i < 3; ++i) { Jraw[(9 + i) * LD + cm(21 + 3 + i)] = 1.0; Jraw[(12 + i) * LD + cm(21 + 6 + i)] = 1.0; }
}

// ---------------------------------------------------------------- a3: PoseLocalParameterization::Plus
UVS_DEV void pose_plus(const double* x, const double* d, double* o) {   // pose_local_parameterization.cpp:3-19
    o[0] = x[0] + d[0]; o[1] = x[1] + d[1]; o[2] = x[2] + d[2];
    const double dq[4] = {d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0};
    double qn[4]; quat_mul(x + 3, dq, qn);
    const double in = rsqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
    o[3] = qn[0] * in; o[4] = qn[1] * in; o[5] = qn[2] * in; o[6] = qn[3] * in;
}

}  // namespace uvsdev
