// estimator.h -- host mirror of the hot-path part of class Estimator (vins_estimator/src/estimator.h:28-146):
// same member names, same optimization() / vector2double() / double2vector() signatures.  The state machine around it
// (processIMU / processImage / initialStructure / slideWindow / failureDetection) is out of scope.
#pragma once
#include <algorithm>
#include "parameters.h"
#include "feature_manager.h"
#include "problem.h"

class Estimator {
  public:
    Estimator();
    ~Estimator();
    void setParameter();
    void optimization();
    void vector2double();
    void double2vector();
    enum SolverFlag { INITIAL, NON_LINEAR };
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
    SolverFlag solver_flag;
    MarginalizationFlag marginalization_flag;
    Eigen::Matrix3d ric[NUM_OF_CAM]; Eigen::Vector3d tic[NUM_OF_CAM];
    Eigen::Vector3d Ps[(WINDOW_SIZE + 1)], Vs[(WINDOW_SIZE + 1)], Bas[(WINDOW_SIZE + 1)], Bgs[(WINDOW_SIZE + 1)];
    Eigen::Matrix3d Rs[(WINDOW_SIZE + 1)];
    double td;
    Eigen::Matrix3d last_R0; Eigen::Vector3d last_P0;
    IntegrationBase* pre_integrations[(WINDOW_SIZE + 1)];
    FeatureManager f_manager;
    bool failure_occur;
    double para_Pose[WINDOW_SIZE + 1][SIZE_POSE];
    double para_SpeedBias[WINDOW_SIZE + 1][SIZE_SPEEDBIAS];
    double para_Feature[NUM_OF_F][SIZE_FEATURE];
    double para_Ex_Pose[NUM_OF_CAM][SIZE_POSE];
    double para_Td[1][1];
    double para_Ortho_plucker[NUM_OF_LF][SIZE_LINE_FEATURE];
    MarginalizationInfo* last_marginalization_info;
    // the reference keeps a vector<double*> of the prior's parameter blocks; here the block table lives inside uvs_prior
    uvs_solver* solver;          // HIP back-end handle (created in the constructor; throws when no GPU is present)
    uvs::Summary last_summary;   // kept for diagnostics (the reference discards ceres::Solver::Summary)
};
