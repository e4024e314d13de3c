// estimator.h -- host mirror of the hot-path part of class Estimator (vins_estimator/src/estimator.h:28-146):
// same member names, same optimization() / vector2double() / double2vector() signatures.  For closed-loop replay of a frame
// sequence the post-initialization part of the state machine around it is mirrored too: processIMU (:84-118), processImage
// (:120-222, NON_LINEAR branch), solveOdometry (:511-524), failureDetection (:713-760), slideWindow / slideWindowNew /
// slideWindowOld (:1235-1359).  initialStructure (:224-446: SfM + visual-inertial alignment) stays out of scope: the
// replay hands the estimator the aligned initial window instead (setInitialWindow).
#pragma once
#include <algorithm>
#include <map>
#include "parameters.h"
#include "feature_manager.h"
#include "factor/factors.h"
#include "window_assembly.h"

namespace std_msgs { struct Header { struct Stamp { double t = 0; double toSec() const { return t; } } stamp; }; }      // the one field of std_msgs::Header the estimator reads

class Estimator {
  public:
    Estimator();
    ~Estimator();
    void setParameter();
    void optimization();
    void assembleWindow(uvs::WindowAssembly& wa);      // the problem-construction part of optimization() (:803-978) as index-addressed arrays
    void vector2double();
    void double2vector();
    // ---- per-frame state machine (post-initialization)
    void processIMU(double t, const Eigen::Vector3d& linear_acceleration, const Eigen::Vector3d& angular_velocity);
    void processImage(const FeatureManager::ImagePoints& image, const FeatureManager::ImageLines& image_line, const std_msgs::Header& header);
    void solveOdometry();
    void slideWindow();
    void slideWindowNew();
    void slideWindowOld();
    bool failureDetection();
    void clearState();
    void setReloFrame(double _frame_stamp, int _frame_index, std::vector<Eigen::Vector3d>& _match_points, Eigen::Vector3d _relo_t, Eigen::Matrix3d _relo_r);
    // initialStructure() (:224-446) itself is out of scope; the replay supplies what it would leave behind (aligned states of frames
    // 0..WINDOW_SIZE) through setInitialWindow(), and the stand-in below installs them when the window is full
    void setInitialWindow(const double (*pose)[7], const double (*speedbias)[9]);
    bool initialStructure();
    // pieces of processIMU() / solveOdometry() (this mirror's own decomposition)
    void recordSample(int slot, double dt, const Eigen::Vector3d& acc, const Eigen::Vector3d& gyr);
    void deadReckon(int frame, double dt, const Eigen::Vector3d& acc, const Eigen::Vector3d& gyr);
    void triangulateNewLandmarks();
    std::vector<double> initial_window;      // [11][7] poses then [11][9] speed/bias; empty = none supplied => initialStructure() fails
    int frame_count;
    bool first_imu;
    Eigen::Vector3d acc_0, gyr_0;
    std::vector<double> dt_buf[(WINDOW_SIZE + 1)];
    std::vector<Eigen::Vector3d> linear_acceleration_buf[(WINDOW_SIZE + 1)], angular_velocity_buf[(WINDOW_SIZE + 1)];
    std_msgs::Header Headers[(WINDOW_SIZE + 1)];
    Eigen::Matrix3d back_R0, last_R; Eigen::Vector3d back_P0, last_P;
    int sum_of_back, sum_of_front;
    std::vector<Eigen::Vector3d> key_poses;
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
    // round 4: the marginalization of optimization() runs beside the caller's work between two frames (uvs_marginalize_resident_begin / _wait).  While it is in
    // flight the window it reads stays alive here: the assembly (observation arrays), the descriptor that points into it, and the OLD prior (its input).
    // A marginalization may be IN FLIGHT between two optimization() calls (uvs_marginalize_resident_begin: a worker thread of the solver handle).  It owns copies of the
    // observation arrays and of the landmark parameters, and reads last_marginalization_info (the previous prior) in place: call finishMarginalization() before touching
    // last_marginalization_info from outside optimization() -- it blocks until the new prior is installed.
    struct PendingMarginalization;
    PendingMarginalization* pending_marginalization = nullptr;
    void finishMarginalization();      // blocks until the new prior is there and installs it as last_marginalization_info; a no-op when nothing is in flight
    // relocalization variables (estimator.h:131-144)
    bool relocalization_info;
    double relo_frame_stamp;
    double relo_frame_index;
    int relo_frame_local_index;
    std::vector<Eigen::Vector3d> match_points;
    double relo_Pose[SIZE_POSE];
    Eigen::Matrix3d drift_correct_r;
    Eigen::Vector3d drift_correct_t;
    Eigen::Vector3d prev_relo_t;
    Eigen::Matrix3d prev_relo_r;
    Eigen::Vector3d relo_relative_t;
    Eigen::Quaterniond relo_relative_q;
    double relo_relative_yaw;
    // the reference keeps a vector<double*> of the prior's parameter blocks; here the block table lives inside uvs_prior
    uvs_solver* solver;          // HIP back-end handle (created in the constructor; throws when no GPU is present)
    uvs_solver* eval_solver;     // second, small handle for the factor classes' per-block Evaluate() (registered with uvs::set_evaluation_solver unless one is registered already)
    uvs::Summary last_summary;   // kept for diagnostics (the reference discards ceres::Solver::Summary)
    uvs::SolverPath solver_path = uvs::AUTO;      // which single-window form optimization() calls (window_assembly.h)
    // wall-clock spent inside optimization() (sums over the calls): whole call, uvs::Solve() alone, marginalization alone
    double optimization_ms = 0, solve_ms = 0, marginalize_ms = 0; int optimization_calls = 0;
};
