// factors.h -- host-side factor classes with the reference's names and constructor signatures.
// In the reference each class derives from a Ceres cost function and its Evaluate() runs on the CPU; here the classes are
// DATA CARRIERS: Estimator::optimization() does not build them at all (it fills the flat uvs_window straight from the track lists,
// window_assembly.h) and the residuals / Jacobians are evaluated by the HIP kernels (csrc/uvs_factors.h).  The per-block evaluation surface of the reference is kept all the
// same -- Evaluate(parameters, residuals, jacobians) for the hand-coded factors (projection_factor.h:23, imu_factor.h:19,
// marginalization_factor.h:78), operator()(pose, line, residuals) for the two auto-differentiated functors
// (line_projection_factor.h:16-19, vp_projection_factor.h:19-22) -- and is ROUTED THROUGH THE GPU: each call builds the one-block
// window and runs uvs_evaluate() on the handle given to uvs::set_evaluation_solver() (factor_evaluate.cpp).  There is no CPU evaluation;
// without a handle the calls return false.  Row-major Jacobian buffers of shape rows x global_size, nullptr = not requested.
#pragma once
#include <vector>
#include "../../../include/uvs_solver.h"
#include "../integration_base.h"

namespace uvs {
enum FactorKind { F_IMU, F_PROJECTION, F_PROJECTION_TD, F_LINE, F_VP, F_MARGINALIZATION };
struct CostFunction { virtual ~CostFunction() {} virtual FactorKind kind() const = 0; };
// the handle the per-block Evaluate() / operator() calls run on (the Estimator registers its own); nullptr = none
void set_evaluation_solver(uvs_solver* s);
uvs_solver* evaluation_solver();
}
namespace ceres_like {   // the few Ceres names the reference's optimization() spells out
struct LossFunction { virtual ~LossFunction() {} double a; explicit LossFunction(double a_) : a(a_) {} };
struct CauchyLoss : LossFunction { explicit CauchyLoss(double a_) : LossFunction(a_) {} };
struct LocalParameterization { virtual ~LocalParameterization() {} };
}

class PoseLocalParameterization : public ceres_like::LocalParameterization {   // pose_local_parameterization.cpp:3-27
  public:
    bool Plus(const double* x, const double* delta, double* x_plus_delta) const {
        Eigen::Quaterniond q(x[6], x[3], x[4], x[5]);
        Eigen::Quaterniond dq = Utility::deltaQ(Eigen::Vector3d(delta[3], delta[4], delta[5]));
        Eigen::Quaterniond r = (q * dq).normalized();
        for (int k = 0; k < 3; ++k) x_plus_delta[k] = x[k] + delta[k];
        x_plus_delta[3] = r.x(); x_plus_delta[4] = r.y(); x_plus_delta[5] = r.z(); x_plus_delta[6] = r.w();
        return true;
    }
    bool ComputeJacobian(const double*, double* jacobian) const { for (int i = 0; i < 42; ++i) jacobian[i] = 0.0; for (int i = 0; i < 6; ++i) jacobian[i * 6 + i] = 1.0; return true; }   // [I6;0], 7x6 row-major
    int GlobalSize() const { return 7; }
    int LocalSize() const { return 6; }
};

class IMUFactor : public uvs::CostFunction {               // imu_factor.h:12-17
  public:
    IMUFactor() = delete;
    explicit IMUFactor(IntegrationBase* _pre_integration) : pre_integration(_pre_integration) {}
    uvs::FactorKind kind() const override { return uvs::F_IMU; }
    // parameters: Pose_i[7], SpeedBias_i[9], Pose_j[7], SpeedBias_j[9]; residuals[15]; jacobians 15x7, 15x9, 15x7, 15x9 (whitened)
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const;
    IntegrationBase* pre_integration;
};
class ProjectionFactor : public uvs::CostFunction {        // projection_factor.h:12-24
  public:
    ProjectionFactor(const Eigen::Vector3d& _pts_i, const Eigen::Vector3d& _pts_j) : pts_i(_pts_i), pts_j(_pts_j) {}
    uvs::FactorKind kind() const override { return uvs::F_PROJECTION; }
    // parameters: Pose_i[7], Pose_j[7], Ex_Pose[7], Feature[1]; residuals[2]; jacobians 2x7, 2x7, 2x7, 2x1 (projection_factor.cpp:22-175)
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const;
    // the reference's self-test (projection_factor.cpp:178-283): analytic Jacobians against central... forward differences of Evaluate()
    // along  Q * deltaQ(eps), eps = 1e-6; returns the largest absolute difference instead of printing it
    double check(double** parameters) const;
    Eigen::Vector3d pts_i, pts_j;
    static double sqrt_info;      // FOCAL_LENGTH / 1.6 (estimator.cpp:17); scalar because the reference's matrix is a multiple of I2
};
class ProjectionTdFactor : public uvs::CostFunction {      // projection_td_factor.h:11-33; ctor projection_td_factor.cpp:6-16 (row_i = _row_i - ROW / 2)
  public:
    ProjectionTdFactor(const Eigen::Vector3d& _pts_i, const Eigen::Vector3d& _pts_j, const Eigen::Vector2d& _velocity_i, const Eigen::Vector2d& _velocity_j,
                       const double _td_i, const double _td_j, const double _row_i, const double _row_j)
        : pts_i(_pts_i), pts_j(_pts_j), velocity_i(_velocity_i), velocity_j(_velocity_j), td_i(_td_i), td_j(_td_j), row_i(_row_i - ROW / 2), row_j(_row_j - ROW / 2) {}
    uvs::FactorKind kind() const override { return uvs::F_PROJECTION_TD; }
    // parameters: Pose_i[7], Pose_j[7], Ex_Pose[7], Feature[1], Td[1]; residuals[2]; jacobians 2x7, 2x7, 2x7, 2x1, 2x1
    // (projection_td_factor.cpp:34-145).  Needs an evaluation handle created with estimate_td = 1 (the Estimator's is when ESTIMATE_TD is set).
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const;
    // projection_td_factor.h:17 -- forward differences (eps = 1e-6, Q * deltaQ) against the analytic Jacobians, 20 directions incl. td
    double check(double** parameters) const;
    Eigen::Vector3d pts_i, pts_j; Eigen::Vector2d velocity_i, velocity_j; double td_i, td_j, row_i, row_j;
};
struct LineProjectionFactor : public uvs::CostFunction {   // line_projection_factor.h:11-19
    LineProjectionFactor(Eigen::Matrix3d _ric, Eigen::Vector3d _tic, Eigen::Vector3d _sp, Eigen::Vector3d _ep) : ric(_ric), tic(_tic), sp(_sp), ep(_ep) {}
    uvs::FactorKind kind() const override { return uvs::F_LINE; }
    // the functor the reference hands to ceres::AutoDiffCostFunction<..., 2, 7, 4>: pose[7], line[4] -> residuals[2].  T = double runs on
    // the GPU; the Jet instantiation has no meaning here (the device carries hand-derived Jacobians): use EvaluateWithJacobians().
    template <typename T> bool operator()(const T* const pose, const T* const line, T* residuals) const {
        static_assert(sizeof(T) == sizeof(double), "only the double instantiation exists: Jacobians come from EvaluateWithJacobians()");
        return EvaluateWithJacobians((const double*)pose, (const double*)line, (double*)residuals, nullptr, nullptr);
    }
    // what Ceres sees after the [I6; 0] local parameterization: J_pose 2 x 7 row-major with columns 0..5 =
    // d r / d (px py pz qx qy qz) at fixed qw and column 6 = 0, J_line 2 x 4
    bool EvaluateWithJacobians(const double* pose, const double* line, double* residuals, double* J_pose, double* J_line) const;
    Eigen::Matrix3d ric; Eigen::Vector3d tic, sp, ep;
};
struct VPProjectionFactor : public uvs::CostFunction {     // vp_projection_factor.h:14-22
    VPProjectionFactor(Eigen::Matrix3d _ric, Eigen::Vector3d _tic, Eigen::Vector3d _sp, Eigen::Vector3d _ep, Eigen::Vector3d _vp) : ric(_ric), tic(_tic), sp(_sp), ep(_ep), vp(_vp) {}
    uvs::FactorKind kind() const override { return uvs::F_VP; }
    template <typename T> bool operator()(const T* const pose, const T* const line, T* residuals) const {      // AutoDiffCostFunction<..., 1, 7, 4>
        static_assert(sizeof(T) == sizeof(double), "only the double instantiation exists: Jacobians come from EvaluateWithJacobians()");
        return EvaluateWithJacobians((const double*)pose, (const double*)line, (double*)residuals, nullptr, nullptr);
    }
    bool EvaluateWithJacobians(const double* pose, const double* line, double* residuals, double* J_pose /*1x7*/, double* J_line /*1x4*/) const;
    Eigen::Matrix3d ric; Eigen::Vector3d tic, sp, ep, vp;
};

// MarginalizationInfo (marginalization_factor.h:46-72): the prior is held in the C-ABI form; the reference's address-keyed
// maps become (kind, frame) block tables.  marginalize() = uvs_marginalize() (GPU factor evaluation + host eigen-solves).
class MarginalizationInfo {
  public:
    uvs_prior prior;
    int m = 0, n = 0;
    MarginalizationInfo() { prior.n = 0; prior.n_blocks = 0; }
};
class MarginalizationFactor : public uvs::CostFunction {   // marginalization_factor.h:74-81
  public:
    explicit MarginalizationFactor(MarginalizationInfo* _marginalization_info) : marginalization_info(_marginalization_info) {}
    uvs::FactorKind kind() const override { return uvs::F_MARGINALIZATION; }
    // parameters: the kept blocks in the prior's block order; residuals[n]; jacobians[b] = n x block_size[b] row-major, the columns of
    // linearized_jacobians with a zero last column for 7-wide blocks (marginalization_factor.cpp:333-381)
    bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const;
    MarginalizationInfo* marginalization_info;
};
