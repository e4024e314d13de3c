// utility.h -- mirror of vins_estimator/src/utility/utility.h (the helpers on the hot path, a13).
#pragma once
#include <cmath>
#include "eigen_lite.h"
class Utility {
  public:
    static Eigen::Quaterniond deltaQ(const Eigen::Vector3d& theta) { return Eigen::Quaterniond(1.0, theta.x() / 2.0, theta.y() / 2.0, theta.z() / 2.0); }   // :11-24 (not normalised)
    static Eigen::Matrix3d skewSymmetric(const Eigen::Vector3d& q) {      // :26-34  [q]x
        Eigen::Matrix3d S;
        for (int i = 0; i < 3; ++i) { const int j = (i + 1) % 3, k = (i + 2) % 3; S(j, k) = -q(i); S(k, j) = q(i); }
        return S;
    }
    static Eigen::Vector3d R2ypr(const Eigen::Matrix3d& R) {      // :66-81, DEGREES: yaw from the first column, pitch / roll after undoing the yaw
        const double rad2deg = 180.0;
        const double yaw = std::atan2(R(1, 0), R(0, 0)), cy = std::cos(yaw), sy = std::sin(yaw);
        const double pitch = std::atan2(-R(2, 0), R(0, 0) * cy + R(1, 0) * sy);
        const double roll = std::atan2(R(0, 2) * sy - R(1, 2) * cy, R(1, 1) * cy - R(0, 1) * sy);
        return Eigen::Vector3d(yaw / M_PI * rad2deg, pitch / M_PI * rad2deg, roll / M_PI * rad2deg);
    }
    static Eigen::Matrix3d ypr2R(const Eigen::Vector3d& ypr) {    // :83-108, DEGREES: Rz(yaw) Ry(pitch) Rx(roll) multiplied out
        const double d2r = M_PI;
        const double y = ypr(0) / 180.0 * d2r, p = ypr(1) / 180.0 * d2r, r = ypr(2) / 180.0 * d2r;
        const double cy = std::cos(y), sy = std::sin(y), cp = std::cos(p), sp = std::sin(p), cr = std::cos(r), sr = std::sin(r);
        Eigen::Matrix3d R;
        R(0, 0) = cy * cp; R(0, 1) = (cy * sp) * sr - sy * cr; R(0, 2) = (cy * sp) * cr + sy * sr;
        R(1, 0) = sy * cp; R(1, 1) = (sy * sp) * sr + cy * cr; R(1, 2) = (sy * sp) * cr - cy * sr;
        R(2, 0) = -sp;     R(2, 1) = cp * sr;                  R(2, 2) = cp * cr;
        return R;
    }
    static double normalizeAngle(double deg) {                    // :130-139: into (-180, 180], whole turns removed symmetrically around zero
        const double turns = std::floor((std::fabs(deg) + 180.0) / 360.0);
        return deg > 0 ? deg - 360.0 * turns : deg + 360.0 * turns;
    }
};
