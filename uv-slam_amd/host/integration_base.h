// integration_base.h -- host mirror of IntegrationBase (vins_estimator/src/factor/integration_base.h:9-208).
// Midpoint pre-integration with Jacobian / covariance propagation (sequential, ~20 samples
// per frame => host code).  jacobian / covariance are 15x15 ROW-major, the layout uvs_imu_block expects.
#pragma once
#include <cstring>
#include <vector>
#include "parameters.h"
#include "utility.h"

class IntegrationBase {
  public:
    IntegrationBase() = delete;
    IntegrationBase(const Eigen::Vector3d& _acc_0, const Eigen::Vector3d& _gyr_0, const Eigen::Vector3d& _linearized_ba, const Eigen::Vector3d& _linearized_bg)
        : linearized_acc(_acc_0), linearized_gyr(_gyr_0) {
        std::memset(noise, 0, sizeof(noise));                                    // :21-27: diag(acc_n, gyr_n, acc_n, gyr_n, acc_w, gyr_w)^2 (x) I3
        const double sigma[6] = {ACC_N, GYR_N, ACC_N, GYR_N, ACC_W, GYR_W};
        for (int d = 0; d < 18; ++d) noise[d * 18 + d] = sigma[d / 3] * sigma[d / 3];
        restart(_linearized_ba, _linearized_bg);
    }
    // one more IMU sample (:30-36)
    void push_back(double dt, const Eigen::Vector3d& acc, const Eigen::Vector3d& gyr) { dt_buf.push_back(dt); acc_buf.push_back(acc); gyr_buf.push_back(gyr); propagate(dt, acc, gyr); }
    // new bias linearization point: integrate the stored samples again from the first one (:38-52)
    void repropagate(const Eigen::Vector3d& ba, const Eigen::Vector3d& bg) {
        restart(ba, bg);
        for (std::size_t k = 0; k < dt_buf.size(); ++k) propagate(dt_buf[k], acc_buf[k], gyr_buf[k]);
    }
    // empty pre-integration at the given biases: identity delta, identity Jacobian, zero covariance
    void restart(const Eigen::Vector3d& ba, const Eigen::Vector3d& bg) {
        linearized_ba = ba; linearized_bg = bg;
        acc_0 = linearized_acc; gyr_0 = linearized_gyr;
        delta_p.setZero(); delta_v.setZero(); delta_q.setIdentity();
        sum_dt = 0.0;
        std::memset(covariance, 0, sizeof(covariance));
        setIdentity15(jacobian);
    }
    void propagate(double _dt, const Eigen::Vector3d& _acc_1, const Eigen::Vector3d& _gyr_1) {       // :130-158 + midPointIntegration :54-128
        using namespace Eigen;
        const double dt = _dt;
        Vector3d un_acc_0 = delta_q * (acc_0 - linearized_ba);
        Vector3d un_gyr = (gyr_0 + _gyr_1) * 0.5 - linearized_bg;
        Quaterniond rq = delta_q * Quaterniond(1, un_gyr(0) * dt / 2, un_gyr(1) * dt / 2, un_gyr(2) * dt / 2);
        Vector3d un_acc_1 = rq * (_acc_1 - linearized_ba);
        Vector3d un_acc = (un_acc_0 + un_acc_1) * 0.5;
        Vector3d rp = delta_p + delta_v * dt + un_acc * (0.5 * dt * dt);
        Vector3d rv = delta_v + un_acc * dt;
        Matrix3d Rq = delta_q.toRotationMatrix(), Rr = rq.toRotationMatrix(), I3 = Matrix3d::Identity();
        Matrix3d Rw = Utility::skewSymmetric(un_gyr), Ra0 = Utility::skewSymmetric(acc_0 - linearized_ba), Ra1 = Utility::skewSymmetric(_acc_1 - linearized_ba);
        double F[225] = {0}, V[15 * 18] = {0};
        auto putF = [&](int r, int c, const Matrix3d& M) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) F[(r + i) * 15 + c + j] = M(i, j); };
        auto putV = [&](int r, int c, const Matrix3d& M) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[(r + i) * 18 + c + j] = M(i, j); };
        Matrix3d RrA1 = Rr * Ra1, T = I3 - Rw * dt;
        putF(0, 0, I3); putF(0, 3, (Rq * Ra0) * (-0.25 * dt * dt) + (RrA1 * T) * (-0.25 * dt * dt)); putF(0, 6, I3 * dt);
        putF(0, 9, (Rq + Rr) * (-0.25 * dt * dt)); putF(0, 12, RrA1 * (-0.25 * dt * dt * -dt));
        putF(3, 3, T); putF(3, 12, I3 * (-dt));
        putF(6, 3, (Rq * Ra0) * (-0.5 * dt) + (RrA1 * T) * (-0.5 * dt)); putF(6, 6, I3); putF(6, 9, (Rq + Rr) * (-0.5 * dt)); putF(6, 12, RrA1 * (-0.5 * dt * -dt));
        putF(9, 9, I3); putF(12, 12, I3);
        Matrix3d v03 = RrA1 * (-0.25 * dt * dt * 0.5 * dt), v63 = RrA1 * (-0.5 * dt * 0.5 * dt);
        putV(0, 0, Rq * (0.25 * dt * dt)); putV(0, 3, v03); putV(0, 6, Rr * (0.25 * dt * dt)); putV(0, 9, v03);
        putV(3, 3, I3 * (0.5 * dt)); putV(3, 9, I3 * (0.5 * dt));
        putV(6, 0, Rq * (0.5 * dt)); putV(6, 3, v63); putV(6, 6, Rr * (0.5 * dt)); putV(6, 9, v63);
        putV(9, 12, I3 * dt); putV(12, 15, I3 * dt);
        double FJ[225], FC[225], VN[15 * 18];
        for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) { double a = 0, b = 0; for (int k = 0; k < 15; ++k) { a += F[i * 15 + k] * jacobian[k * 15 + j]; b += F[i * 15 + k] * covariance[k * 15 + j]; } FJ[i * 15 + j] = a; FC[i * 15 + j] = b; }
        for (int i = 0; i < 15; ++i) for (int j = 0; j < 18; ++j) { double a = 0; for (int k = 0; k < 18; ++k) a += V[i * 18 + k] * noise[k * 18 + j]; VN[i * 18 + j] = a; }
        for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) {
            double a = 0; for (int k = 0; k < 15; ++k) a += FC[i * 15 + k] * F[j * 15 + k];
            double b = 0; for (int k = 0; k < 18; ++k) b += VN[i * 18 + k] * V[j * 18 + k];
            covariance[i * 15 + j] = a + b;                                     // F cov F^T + V noise V^T  (:125)
        }
        std::memcpy(jacobian, FJ, sizeof(FJ));                                  // jacobian = F * jacobian (:124)
        delta_p = rp; delta_q = rq.normalized(); delta_v = rv;                  // :148-153
        sum_dt += dt; acc_0 = _acc_1; gyr_0 = _gyr_1;
    }
    // fields read by IMUFactor (integration_base.h:188-203)
    Eigen::Vector3d acc_0, gyr_0;
    const Eigen::Vector3d linearized_acc, linearized_gyr;
    Eigen::Vector3d linearized_ba, linearized_bg;
    double jacobian[225], covariance[225], noise[18 * 18];
    double sum_dt;
    Eigen::Vector3d delta_p; Eigen::Quaterniond delta_q; Eigen::Vector3d delta_v;
    std::vector<double> dt_buf; std::vector<Eigen::Vector3d> acc_buf, gyr_buf;
  private:
    static void setIdentity15(double* M) { std::memset(M, 0, 225 * sizeof(double)); for (int i = 0; i < 15; ++i) M[i * 15 + i] = 1.0; }
};
