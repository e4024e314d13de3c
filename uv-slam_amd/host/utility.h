// utility.h -- mirror of vins_estimator/src/utility/utility.h (the helpers on the hot path, a13).
#pragma once
#include <cmath>
#include "eigen_lite.h"
class Utility {
  public:
    static Eigen::Quaterniond deltaQ(const Eigen::Vector3d& theta) { return Eigen::Quaterniond(1.0, theta.x() / 2.0, theta.y() / 2.0, theta.z() / 2.0); }   // :11-24 (not normalised)
    static Eigen::Matrix3d skewSymmetric(const Eigen::Vector3d& q) { Eigen::Matrix3d a; a(0, 1) = -q(2); a(0, 2) = q(1); a(1, 0) = q(2); a(1, 2) = -q(0); a(2, 0) = -q(1); a(2, 1) = q(0); return a; }
    static Eigen::Vector3d R2ypr(const Eigen::Matrix3d& R) {      // :66-81, DEGREES
        Eigen::Vector3d n = R.col(0), o = R.col(1), a = R.col(2);
        const double y = atan2(n(1), n(0));
        const double p = atan2(-n(2), n(0) * cos(y) + n(1) * sin(y));
        const double r = atan2(a(0) * sin(y) - a(1) * cos(y), -o(0) * sin(y) + o(1) * cos(y));
        return Eigen::Vector3d(y, p, r) / M_PI * 180.0;
    }
    static Eigen::Matrix3d ypr2R(const Eigen::Vector3d& ypr) {    // :83-108, DEGREES
        const double y = ypr(0) / 180.0 * M_PI, p = ypr(1) / 180.0 * M_PI, r = ypr(2) / 180.0 * M_PI;
        Eigen::Matrix3d Rz, Ry, Rx;
        Rz(0, 0) = cos(y); Rz(0, 1) = -sin(y); Rz(1, 0) = sin(y); Rz(1, 1) = cos(y); Rz(2, 2) = 1;
        Ry(0, 0) = cos(p); Ry(0, 2) = sin(p); Ry(1, 1) = 1; Ry(2, 0) = -sin(p); Ry(2, 2) = cos(p);
        Rx(0, 0) = 1; Rx(1, 1) = cos(r); Rx(1, 2) = -sin(r); Rx(2, 1) = sin(r); Rx(2, 2) = cos(r);
        return Rz * Ry * Rx;
    }
    static double normalizeAngle(double angle_degrees) {          // :130-139
        const double two_pi = 2.0 * 180;
        if (angle_degrees > 0) return angle_degrees - two_pi * std::floor((angle_degrees + 180) / two_pi);
        return angle_degrees + two_pi * std::floor((-angle_degrees + 180) / two_pi);
    }
};
