// eigen_lite.h -- the handful of Eigen types the mirrored Estimator API needs (Eigen itself is not in this image).
// A host project that has the real Eigen defines UVS_HAVE_EIGEN and includes <Eigen/Dense> instead; the members used
// by host/*.h (x() y() z() w(), operator(), toRotationMatrix(), normalized(), transpose(), operator*) are the Eigen ones.
#pragma once
#ifdef UVS_HAVE_EIGEN
#include <Eigen/Dense>
#else
#include <cmath>
#include <vector>
namespace Eigen {
struct Vector3d {
    double v[3];
    Vector3d() : v{0, 0, 0} {}
    Vector3d(double a, double b, double c) : v{a, b, c} {}
    double& x() { return v[0]; } double& y() { return v[1]; } double& z() { return v[2]; }
    double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
    double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; }
    Vector3d operator+(const Vector3d& o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    Vector3d operator-(const Vector3d& o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    Vector3d operator-() const { return {-v[0], -v[1], -v[2]}; }
    Vector3d operator*(double s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
    Vector3d operator/(double s) const { return {v[0] / s, v[1] / s, v[2] / s}; }
    double dot(const Vector3d& o) const { return v[0] * o.v[0] + v[1] * o.v[1] + v[2] * o.v[2]; }
    Vector3d cross(const Vector3d& o) const { return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]}; }
    double norm() const { return std::sqrt(dot(*this)); }
    void setZero() { v[0] = v[1] = v[2] = 0; }
    static Vector3d Zero() { return {}; }
};
inline Vector3d operator*(double s, const Vector3d& a) { return a * s; }
struct Vector2d { double v[2]; Vector2d() : v{0, 0} {} Vector2d(double a, double b) : v{a, b} {}
    double& x() { return v[0]; } double& y() { return v[1]; } double x() const { return v[0]; } double y() const { return v[1]; }
    double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Vector4d { double v[4]; Vector4d() : v{0, 0, 0, 0} {} Vector4d(double a, double b, double c, double d) : v{a, b, c, d} {}
    double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };
struct Matrix3d {
    double m[3][3];
    Matrix3d() { for (auto& r : m) for (double& e : r) e = 0; }
    double& operator()(int i, int j) { return m[i][j]; } double operator()(int i, int j) const { return m[i][j]; }
    void setIdentity() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = i == j; }
    static Matrix3d Identity() { Matrix3d I; I.setIdentity(); return I; }
    Matrix3d transpose() const { Matrix3d t; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) t.m[i][j] = m[j][i]; return t; }
    Matrix3d operator*(const Matrix3d& o) const { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += m[i][k] * o.m[k][j]; r.m[i][j] = s; } return r; }
    Vector3d operator*(const Vector3d& a) const { return {m[0][0] * a.v[0] + m[0][1] * a.v[1] + m[0][2] * a.v[2], m[1][0] * a.v[0] + m[1][1] * a.v[1] + m[1][2] * a.v[2], m[2][0] * a.v[0] + m[2][1] * a.v[1] + m[2][2] * a.v[2]}; }
    Matrix3d operator*(double s) const { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] * s; return r; }
    Matrix3d operator+(const Matrix3d& o) const { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] + o.m[i][j]; return r; }
    Matrix3d operator-(const Matrix3d& o) const { Matrix3d r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.m[i][j] = m[i][j] - o.m[i][j]; return r; }
    Vector3d col(int j) const { return {m[0][j], m[1][j], m[2][j]}; }
};
struct Quaterniond {
    double qw, qx, qy, qz;
    Quaterniond() : qw(1), qx(0), qy(0), qz(0) {}
    Quaterniond(double w, double x, double y, double z) : qw(w), qx(x), qy(y), qz(z) {}
    explicit Quaterniond(const Matrix3d& R) {   // Shepperd, same branch structure as Eigen's quaternion-from-matrix
        double t = R(0, 0) + R(1, 1) + R(2, 2);
        if (t > 0) { t = std::sqrt(t + 1.0); qw = 0.5 * t; t = 0.5 / t; qx = (R(2, 1) - R(1, 2)) * t; qy = (R(0, 2) - R(2, 0)) * t; qz = (R(1, 0) - R(0, 1)) * t; }
        else {
            int i = 0; if (R(1, 1) > R(0, 0)) i = 1; if (R(2, 2) > R(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0);
            double q[3]; q[i] = 0.5 * t; t = 0.5 / t; qw = (R(k, j) - R(j, k)) * t; q[j] = (R(j, i) + R(i, j)) * t; q[k] = (R(k, i) + R(i, k)) * t;
            qx = q[0]; qy = q[1]; qz = q[2];
        }
    }
    double w() const { return qw; } double x() const { return qx; } double y() const { return qy; } double z() const { return qz; }
    double& w() { return qw; } double& x() { return qx; } double& y() { return qy; } double& z() { return qz; }
    Quaterniond operator*(const Quaterniond& b) const {
        return {qw * b.qw - qx * b.qx - qy * b.qy - qz * b.qz, qw * b.qx + qx * b.qw + qy * b.qz - qz * b.qy,
                qw * b.qy + qy * b.qw + qz * b.qx - qx * b.qz, qw * b.qz + qz * b.qw + qx * b.qy - qy * b.qx};
    }
    Quaterniond normalized() const { const double n = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz); return {qw / n, qx / n, qy / n, qz / n}; }
    void normalize() { *this = normalized(); }
    Quaterniond inverse() const { const double n2 = qw * qw + qx * qx + qy * qy + qz * qz; return {qw / n2, -qx / n2, -qy / n2, -qz / n2}; }
    Matrix3d toRotationMatrix() const {
        Matrix3d R; const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz, twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy; R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
        R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy); return R;
    }
    Vector3d operator*(const Vector3d& a) const { return toRotationMatrix() * a; }
    void setIdentity() { qw = 1; qx = qy = qz = 0; }
    static Quaterniond Identity() { return {}; }
};
typedef std::vector<double> VectorXd;
// the front-end hands features over as Eigen::Matrix<double, 7, 1> / <double, 15, 1> columns (estimator.h:41-42): element access only
template <typename T, int R, int C> struct Matrix { T v[R * C]; Matrix() { for (auto& e : v) e = 0; } T& operator()(int i) { return v[i]; } T operator()(int i) const { return v[i]; } };
}  // namespace Eigen
#endif
