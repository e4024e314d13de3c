// oracle_math.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Tiny templated 3-vector / 3x3 / quaternion algebra plus a forward-mode dual
// number ("Jet") so that the oracle can differentiate the line / vanishing-point
// functors the same way the reference does (ceres::AutoDiffCostFunction over the
// RAW 7-scalar pose).  The formulas for quaternion
// product, quaternion->matrix and quaternion*vector are the polynomial forms
// Eigen 3 evaluates (no normalisation), which is what the reference's Jets see.
//
// Nothing here is shipped: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load the library built from this directory.
#pragma once
#include <cmath>
#include <cstring>

namespace orc {

// ---------------------------------------------------------------- Jet<N>
template <int N>
struct Jet {
    double a;
    double v[N];
    Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
    Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT (implicit on purpose)
    Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
};
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) { Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
    // Ceres: h = f/g ; dh = (df - h dg)/g
    Jet<N> h; const double gi = 1.0 / g.a; h.a = f.a * gi;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - h.a * g.v[i]) * gi;
    return h; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return Jet<N>(s) * f; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) { return f * Jet<N>(s); }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { return Jet<N>(s) + f; }
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { return f + Jet<N>(s); }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { return Jet<N>(s) - f; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { return f - Jet<N>(s); }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { return f / Jet<N>(s); }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& f) { return Jet<N>(s) / f; }
template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> inline Jet<N>& operator-=(Jet<N>& f, const Jet<N>& g) { f = f - g; return f; }

template <int N> inline Jet<N> jcos(const Jet<N>& f) { Jet<N> h; h.a = std::cos(f.a); const double d = -std::sin(f.a); for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
template <int N> inline Jet<N> jsin(const Jet<N>& f) { Jet<N> h; h.a = std::sin(f.a); const double d = std::cos(f.a); for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
template <int N> inline Jet<N> jsqrt(const Jet<N>& f) { Jet<N> h; h.a = std::sqrt(f.a); const double d = 1.0 / (2.0 * h.a); for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
template <int N> inline Jet<N> jpow2(const Jet<N>& f) { Jet<N> h; h.a = std::pow(f.a, 2.0); const double d = 2.0 * f.a; for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
template <int N> inline Jet<N> jabs(const Jet<N>& f) { return f.a < 0.0 ? -f : f; }
template <int N> inline Jet<N> jacos(const Jet<N>& f) { Jet<N> h; h.a = std::acos(f.a); const double d = -1.0 / std::sqrt(1.0 - f.a * f.a); for (int i = 0; i < N; ++i) h.v[i] = d * f.v[i]; return h; }
inline double jcos(double x) { return std::cos(x); }
inline double jsin(double x) { return std::sin(x); }
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jpow2(double x) { return std::pow(x, 2.0); }
inline double jabs(double x) { return std::fabs(x); }
inline double jacos(double x) { return std::acos(x); }
inline double jvalue(double x) { return x; }
template <int N> inline double jvalue(const Jet<N>& f) { return f.a; }

// ---------------------------------------------------------------- V3 / M3 / Quat
template <typename T> struct V3 { T x, y, z; };
template <typename T> struct M3 { T m[3][3]; };
template <typename T> struct Quat { T w, x, y, z; };  // storage order irrelevant; ctor order (w,x,y,z) like Eigen

template <typename T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> inline V3<T> operator-(const V3<T>& a) { return {-a.x, -a.y, -a.z}; }
template <typename T> inline V3<T> operator*(const V3<T>& a, const T& s) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> inline V3<T> operator*(const T& s, const V3<T>& a) { return {a.x * s, a.y * s, a.z * s}; }
template <typename T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <typename T> inline T norm(const V3<T>& a) { return jsqrt(dot(a, a)); }

template <typename T> inline M3<T> mul(const M3<T>& A, const M3<T>& B) {
    M3<T> C;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { T s = A.m[i][0] * B.m[0][j]; s = s + A.m[i][1] * B.m[1][j]; s = s + A.m[i][2] * B.m[2][j]; C.m[i][j] = s; }
    return C;
}
template <typename T> inline V3<T> mul(const M3<T>& A, const V3<T>& v) {
    return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z,
            A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
            A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
template <typename T> inline M3<T> transpose(const M3<T>& A) { M3<T> C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[j][i]; return C; }
template <typename T> inline M3<T> neg(const M3<T>& A) { M3<T> C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = -A.m[i][j]; return C; }
template <typename T> inline M3<T> skew(const V3<T>& q) {   // utility.h:26-34
    M3<T> S;
    S.m[0][0] = T(0.0); S.m[0][1] = -q.z;   S.m[0][2] = q.y;
    S.m[1][0] = q.z;    S.m[1][1] = T(0.0); S.m[1][2] = -q.x;
    S.m[2][0] = -q.y;   S.m[2][1] = q.x;    S.m[2][2] = T(0.0);
    return S;
}
template <typename T> inline M3<T> identity3() { M3<T> I; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) I.m[i][j] = T(i == j ? 1.0 : 0.0); return I; }

// Eigen quaternion product
template <typename T> inline Quat<T> qmul(const Quat<T>& a, const Quat<T>& b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
            a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
            a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
template <typename T> inline Quat<T> qconj(const Quat<T>& q) { return {q.w, -q.x, -q.y, -q.z}; }
// Eigen QuaternionBase::inverse(): conjugate / squaredNorm
template <typename T> inline Quat<T> qinv(const Quat<T>& q) {
    T n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
    return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
template <typename T> inline Quat<T> qnormalized(const Quat<T>& q) {
    T n = jsqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return {q.w / n, q.x / n, q.y / n, q.z / n};
}
// Eigen QuaternionBase::toRotationMatrix() (polynomial, no normalisation)
template <typename T> inline M3<T> qmat(const Quat<T>& q) {
    const T tx = T(2.0) * q.x, ty = T(2.0) * q.y, tz = T(2.0) * q.z;
    const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    M3<T> R;
    R.m[0][0] = T(1.0) - (tyy + tzz); R.m[0][1] = txy - twz;            R.m[0][2] = txz + twy;
    R.m[1][0] = txy + twz;            R.m[1][1] = T(1.0) - (txx + tzz); R.m[1][2] = tyz - twx;
    R.m[2][0] = txz - twy;            R.m[2][1] = tyz + twx;            R.m[2][2] = T(1.0) - (txx + tyy);
    return R;
}
// Eigen QuaternionBase::_transformVector
template <typename T> inline V3<T> qrot(const Quat<T>& q, const V3<T>& v) {
    V3<T> u = {q.x, q.y, q.z};
    V3<T> uv = cross(u, v);
    uv = uv + uv;
    return v + q.w * uv + cross(u, uv);
}
// Eigen Quaternion(AngleAxis): w = cos(a/2), vec = sin(a/2)*axis
template <typename T> inline Quat<T> qaxis(const T& angle, int axis) {
    T ha = T(0.5) * angle;
    T c = jcos(ha), s = jsin(ha);
    Quat<T> q = {c, T(0.0), T(0.0), T(0.0)};
    if (axis == 0) q.x = s; else if (axis == 1) q.y = s; else q.z = s;
    return q;
}

template <typename T, typename S> inline V3<T> cast3(const V3<S>& a) { return {T(a.x), T(a.y), T(a.z)}; }
template <typename T, typename S> inline M3<T> castm(const M3<S>& A) { M3<T> C; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) C.m[i][j] = T(A.m[i][j]); return C; }

typedef V3<double> V3d;
typedef M3<double> M3d;
typedef Quat<double> Qd;

inline V3d v3(const double* p) { return {p[0], p[1], p[2]}; }
inline Qd quat_xyzw(const double* p) { return {p[3], p[0], p[1], p[2]}; }   // p = (x,y,z,w)

}  // namespace orc
