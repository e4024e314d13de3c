// estimator.cpp -- Estimator::optimization() / vector2double() / double2vector() with the reference's structure
// (vins_estimator/src/estimator.cpp:526-711, 761-1233), the Ceres problem replaced by uvs::Problem and the
// marginalization by uvs_marginalize().  See INTEGRATION.md for the diff a maintainer applies to the reference file.
#include "estimator.h"
#include <cstdlib>
#include <stdexcept>
#include "window_io.h"

double FOCAL_LENGTH, INIT_DEPTH, MIN_PARALLAX, ACC_N, ACC_W, GYR_N, GYR_W, SOLVER_TIME, TD, TR, LINE_FACTOR, VP_FACTOR, ROW, COL;
int ESTIMATE_EXTRINSIC, ESTIMATE_TD, NUM_ITERATIONS, LINE_WINDOW;
std::vector<Eigen::Matrix3d> RIC; std::vector<Eigen::Vector3d> TIC;
Eigen::Vector3d G(0.0, 0.0, 9.8);
double ProjectionFactor::sqrt_info;

void setEurocParameters() {          // config/euroc/euroc_config.yaml
    FOCAL_LENGTH = 461.6; SOLVER_TIME = 0.1; NUM_ITERATIONS = 10; ACC_N = 0.08; GYR_N = 0.004; ACC_W = 0.00004; GYR_W = 2.0e-6; G = Eigen::Vector3d(0, 0, 9.81007);
    ESTIMATE_EXTRINSIC = 0; ESTIMATE_TD = 0; TD = 0.0; TR = 0.0; ROW = 480.0; COL = 752.0; LINE_WINDOW = 5; LINE_FACTOR = 300.0; VP_FACTOR = 10.0; INIT_DEPTH = 5.0; MIN_PARALLAX = 10.0 / FOCAL_LENGTH;
    Eigen::Matrix3d R; const double r[9] = {0.0148655429818, -0.999880929698, 0.00414029679422, 0.999557249008, 0.0149672133247, 0.025715529948, -0.0257744366974, 0.00375618835797, 0.999660727178};
    for (int i = 0; i < 9; ++i) R(i / 3, i % 3) = r[i];
    RIC.assign(1, Eigen::Quaterniond(R).normalized().toRotationMatrix());           // parameters.cpp:113-115
    TIC.assign(1, Eigen::Vector3d(-0.0216401454975, -0.064676986768, 0.00981073058949));
}

Estimator::Estimator() : frame_count(0), first_imu(false), sum_of_back(0), sum_of_front(0), solver_flag(NON_LINEAR), marginalization_flag(MARGIN_OLD), td(0), failure_occur(false), last_marginalization_info(nullptr), relocalization_info(false), relo_frame_stamp(0), relo_frame_index(0), relo_frame_local_index(0), relo_relative_yaw(0), solver(nullptr) {
    f_manager.Rs = Rs;      // estimator.cpp:9 `f_manager{Rs}`
    for (auto& p : pre_integrations) p = nullptr;
    for (int i = 0; i <= WINDOW_SIZE; ++i) Rs[i].setIdentity();
    uvs_options o; uvs_default_options(&o);
    o.estimate_td = ESTIMATE_TD; o.estimate_extrinsic = ESTIMATE_EXTRINSIC != 0;      // fixed for the lifetime of the handle, like the reference's globals (parameters.cpp)
    const int rc = uvs_create(&o, 0, 1, NUM_OF_F, NUM_OF_F * (WINDOW_SIZE + 1), NUM_OF_LF, NUM_OF_LF * (WINDOW_SIZE + 1), &solver);
    if (rc != UVS_OK) throw std::runtime_error(std::string("uvs_create: ") + uvs_status_string(rc));     // no CPU fallback
}
Estimator::~Estimator() { uvs_destroy(solver); delete last_marginalization_info; for (auto* p : pre_integrations) delete p; }

void Estimator::setParameter() {      // estimator.cpp:9-21
    for (int i = 0; i < NUM_OF_CAM; i++) { tic[i] = TIC[i]; ric[i] = RIC[i]; }
    ProjectionFactor::sqrt_info = FOCAL_LENGTH / 1.6;
    td = TD;
}

void Estimator::vector2double() {     // estimator.cpp:526-594
    for (int i = 0; i <= WINDOW_SIZE; i++) {
        para_Pose[i][0] = Ps[i].x(); para_Pose[i][1] = Ps[i].y(); para_Pose[i][2] = Ps[i].z();
        Eigen::Quaterniond q{Rs[i]};
        para_Pose[i][3] = q.x(); para_Pose[i][4] = q.y(); para_Pose[i][5] = q.z(); para_Pose[i][6] = q.w();
        for (int k = 0; k < 3; ++k) { para_SpeedBias[i][k] = Vs[i](k); para_SpeedBias[i][3 + k] = Bas[i](k); para_SpeedBias[i][6 + k] = Bgs[i](k); }
    }
    for (int i = 0; i < NUM_OF_CAM; i++) {
        para_Ex_Pose[i][0] = tic[i].x(); para_Ex_Pose[i][1] = tic[i].y(); para_Ex_Pose[i][2] = tic[i].z();
        Eigen::Quaterniond q{ric[i]};
        para_Ex_Pose[i][3] = q.x(); para_Ex_Pose[i][4] = q.y(); para_Ex_Pose[i][5] = q.z(); para_Ex_Pose[i][6] = q.w();
    }
    Eigen::VectorXd dep = f_manager.getDepthVector();
    for (int i = 0; i < f_manager.getFeatureCount(); i++) para_Feature[i][0] = dep[i];
    if (ESTIMATE_TD) para_Td[0][0] = td;
    std::vector<Eigen::Vector4d> get_lineOrtho = f_manager.getLineOrthonormal();
    for (int i = 0; i < f_manager.getLineFeatureCount(); i++) for (int k = 0; k < 4; ++k) para_Ortho_plucker[i][k] = get_lineOrtho.at(i)[k];
}

void Estimator::double2vector() {     // estimator.cpp:596-711
    using namespace Eigen;
    Vector3d origin_R0 = Utility::R2ypr(Rs[0]);
    Vector3d origin_P0 = Ps[0];
    if (failure_occur) { origin_R0 = Utility::R2ypr(last_R0); origin_P0 = last_P0; failure_occur = 0; }
    Matrix3d R00 = Quaterniond(para_Pose[0][6], para_Pose[0][3], para_Pose[0][4], para_Pose[0][5]).toRotationMatrix();
    Vector3d origin_R00 = Utility::R2ypr(R00);
    double y_diff = origin_R0.x() - origin_R00.x();
    Matrix3d rot_diff = Utility::ypr2R(Vector3d(y_diff, 0, 0));
    if (std::abs(std::abs(origin_R0.y()) - 90) < 1.0 || std::abs(std::abs(origin_R00.y()) - 90) < 1.0) rot_diff = Rs[0] * R00.transpose();   // euler singular point (:616-625)
    for (int i = 0; i <= WINDOW_SIZE; i++) {
        Rs[i] = rot_diff * Quaterniond(para_Pose[i][6], para_Pose[i][3], para_Pose[i][4], para_Pose[i][5]).normalized().toRotationMatrix();
        Ps[i] = rot_diff * Vector3d(para_Pose[i][0] - para_Pose[0][0], para_Pose[i][1] - para_Pose[0][1], para_Pose[i][2] - para_Pose[0][2]) + origin_P0;
        Vs[i] = rot_diff * Vector3d(para_SpeedBias[i][0], para_SpeedBias[i][1], para_SpeedBias[i][2]);
        Bas[i] = Vector3d(para_SpeedBias[i][3], para_SpeedBias[i][4], para_SpeedBias[i][5]);
        Bgs[i] = Vector3d(para_SpeedBias[i][6], para_SpeedBias[i][7], para_SpeedBias[i][8]);
    }
    for (int i = 0; i < NUM_OF_CAM; i++) {
        tic[i] = Vector3d(para_Ex_Pose[i][0], para_Ex_Pose[i][1], para_Ex_Pose[i][2]);
        ric[i] = Quaterniond(para_Ex_Pose[i][6], para_Ex_Pose[i][3], para_Ex_Pose[i][4], para_Ex_Pose[i][5]).toRotationMatrix();
    }
    VectorXd dep = f_manager.getDepthVector();
    for (int i = 0; i < f_manager.getFeatureCount(); i++) dep[i] = para_Feature[i][0];
    f_manager.setDepth(dep);
    if (ESTIMATE_TD) td = para_Td[0][0];
    std::vector<Vector4d> get_lineOrtho = f_manager.getLineOrthonormal();
    for (int i = 0; i < f_manager.getLineFeatureCount(); i++) for (int k = 0; k < 4; ++k) get_lineOrtho.at(i)[k] = para_Ortho_plucker[i][k];
    f_manager.setLineOrtho(get_lineOrtho, Ps, Rs, tic[0], ric[0]);
    if (relocalization_info) {        // relative info between two loop frames (:671-691)
        Matrix3d relo_r = rot_diff * Quaterniond(relo_Pose[6], relo_Pose[3], relo_Pose[4], relo_Pose[5]).normalized().toRotationMatrix();
        Vector3d relo_t = rot_diff * Vector3d(relo_Pose[0] - para_Pose[0][0], relo_Pose[1] - para_Pose[0][1], relo_Pose[2] - para_Pose[0][2]) + origin_P0;
        const double drift_correct_yaw = Utility::R2ypr(prev_relo_r).x() - Utility::R2ypr(relo_r).x();
        drift_correct_r = Utility::ypr2R(Vector3d(drift_correct_yaw, 0, 0));
        drift_correct_t = prev_relo_t - drift_correct_r * relo_t;
        relo_relative_t = relo_r.transpose() * (Ps[relo_frame_local_index] - relo_t);
        relo_relative_q = Quaterniond(relo_r.transpose() * Rs[relo_frame_local_index]);
        relo_relative_yaw = Utility::normalizeAngle(Utility::R2ypr(Rs[relo_frame_local_index]).x() - Utility::R2ypr(relo_r).x());
        relocalization_info = 0;
    }
}

void Estimator::setReloFrame(double _frame_stamp, int _frame_index, std::vector<Eigen::Vector3d>& _match_points, Eigen::Vector3d _relo_t, Eigen::Matrix3d _relo_r) {   // estimator.cpp:1361-1379
    relo_frame_stamp = _frame_stamp;
    relo_frame_index = _frame_index;
    match_points.clear();
    match_points = _match_points;
    prev_relo_t = _relo_t;
    prev_relo_r = _relo_r;
    for (int i = 0; i < WINDOW_SIZE; i++) {
        if (relo_frame_stamp == Headers[i].stamp.toSec()) {
            relo_frame_local_index = i;
            relocalization_info = 1;
            for (int j = 0; j < SIZE_POSE; j++) relo_Pose[j] = para_Pose[i][j];
        }
    }
}

void Estimator::optimization() {      // estimator.cpp:761-1233
    uvs::AddressMap amap{para_Pose, para_SpeedBias, para_Ex_Pose, para_Feature, para_Ortho_plucker, para_Td, relo_Pose};
    uvs::Problem problem(amap);
    ceres_like::LossFunction* loss_function = new ceres_like::CauchyLoss(1.0);
    ceres_like::LossFunction* line_loss_function = new ceres_like::CauchyLoss(0.1);
    ceres_like::LossFunction* vp_loss_function = new ceres_like::CauchyLoss(1.0);
    for (int i = 0; i < WINDOW_SIZE + 1; i++) {
        problem.AddParameterBlock(para_Pose[i], SIZE_POSE, new PoseLocalParameterization());
        problem.AddParameterBlock(para_SpeedBias[i], SIZE_SPEEDBIAS);
    }
    for (int i = 0; i < NUM_OF_CAM; i++) {
        problem.AddParameterBlock(para_Ex_Pose[i], SIZE_POSE, new PoseLocalParameterization());
        if (!ESTIMATE_EXTRINSIC) problem.SetParameterBlockConstant(para_Ex_Pose[i]);
    }
    if (ESTIMATE_TD) problem.AddParameterBlock(para_Td[0], 1);                                    // estimator.cpp:790-797
    vector2double();
    if (last_marginalization_info && last_marginalization_info->prior.n > 0)
        problem.AddResidualBlock(new MarginalizationFactor(last_marginalization_info), NULL, std::vector<double*>{});
    for (int i = 0; i < WINDOW_SIZE; i++) {
        int j = i + 1;
        if (pre_integrations[j]->sum_dt > 10.0) continue;
        problem.AddResidualBlock(new IMUFactor(pre_integrations[j]), NULL, para_Pose[i], para_SpeedBias[i], para_Pose[j], para_SpeedBias[j]);
    }
    int feature_index = -1;
    for (auto& it_per_id : f_manager.feature) {
        it_per_id.used_num = it_per_id.feature_per_frame.size();
        if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
        ++feature_index;
        int imu_i = it_per_id.start_frame, imu_j = imu_i - 1;
        Eigen::Vector3d pts_i = it_per_id.feature_per_frame[0].point;
        for (auto& it_per_frame : it_per_id.feature_per_frame) {
            imu_j++;
            if (imu_i == imu_j) continue;
            if (ESTIMATE_TD)                                                                      // estimator.cpp:853-858
                problem.AddResidualBlock(new ProjectionTdFactor(pts_i, it_per_frame.point, it_per_id.feature_per_frame[0].velocity, it_per_frame.velocity,
                                                                it_per_id.feature_per_frame[0].cur_td, it_per_frame.cur_td, it_per_id.feature_per_frame[0].uv.y(), it_per_frame.uv.y()),
                                         loss_function, para_Pose[imu_i], para_Pose[imu_j], para_Ex_Pose[0], para_Feature[feature_index], para_Td[0]);
            else
                problem.AddResidualBlock(new ProjectionFactor(pts_i, it_per_frame.point), loss_function, para_Pose[imu_i], para_Pose[imu_j], para_Ex_Pose[0], para_Feature[feature_index]);
        }
    }
    int line_feature_index = -1;
    for (auto& it_per_id : f_manager.line_feature) {
        it_per_id.used_num = it_per_id.line_feature_per_frame.size();
        if (it_per_id.used_num < LINE_WINDOW) continue;
        ++line_feature_index;
        int imu_j = it_per_id.start_frame - 1;
        for (auto& it_per_frame : it_per_id.line_feature_per_frame) {
            imu_j++;
            problem.AddResidualBlock(new LineProjectionFactor(ric[0], tic[0], it_per_frame.start_point, it_per_frame.end_point), line_loss_function, para_Pose[imu_j], para_Ortho_plucker[line_feature_index]);
            if (it_per_frame.vp(2) == 1)
                problem.AddResidualBlock(new VPProjectionFactor(ric[0], tic[0], it_per_frame.start_point, it_per_frame.end_point, it_per_frame.vp), vp_loss_function, para_Pose[imu_j], para_Ortho_plucker[line_feature_index]);
        }
    }
    if (relocalization_info) {        // estimator.cpp:944-978
        problem.AddParameterBlock(relo_Pose, SIZE_POSE, new PoseLocalParameterization());
        int retrive_feature_index = 0;
        int relo_feature_index = -1;
        for (auto& it_per_id : f_manager.feature) {
            it_per_id.used_num = it_per_id.feature_per_frame.size();
            if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
            ++relo_feature_index;
            int start = it_per_id.start_frame;
            if (start <= relo_frame_local_index) {
                // (the reference reads match_points[retrive_feature_index] without a bound; an exhausted list simply matches nothing more)
                while (retrive_feature_index < (int)match_points.size() && (int)match_points[retrive_feature_index].z() < it_per_id.feature_id) retrive_feature_index++;
                if (retrive_feature_index < (int)match_points.size() && (int)match_points[retrive_feature_index].z() == it_per_id.feature_id) {
                    Eigen::Vector3d pts_j = Eigen::Vector3d(match_points[retrive_feature_index].x(), match_points[retrive_feature_index].y(), 1.0);
                    Eigen::Vector3d pts_i = it_per_id.feature_per_frame[0].point;
                    problem.AddResidualBlock(new ProjectionFactor(pts_i, pts_j), loss_function, para_Pose[start], relo_Pose, para_Ex_Pose[0], para_Feature[relo_feature_index]);
                    retrive_feature_index++;
                }
            }
        }
    }
    // record hook (the reference has no serialisation): UVS_DUMP_WINDOWS=<dir> writes every window exactly as the
    // solver receives it (the state after vector2double(), estimator.cpp:800) to <dir>/window_NNNN.bin for replay without ROS
    if (const char* dump_dir = std::getenv("UVS_DUMP_WINDOWS")) {
        static int dump_index = 0;
        uvs_window dw; problem.fill(&dw, feature_index + 1, line_feature_index + 1);
        char name[32]; std::snprintf(name, sizeof(name), "/window_%04d.bin", dump_index++);
        WindowFile::save(std::string(dump_dir) + name, dw, relo_frame_local_index);
    }
    uvs::Options options; options.max_num_iterations = NUM_ITERATIONS;
    uvs::Solve(options, &problem, &last_summary, solver, feature_index + 1, line_feature_index + 1);
    // ---- marginalization on the post-solve para_* arrays, BEFORE double2vector() re-anchors the gauge: the reference calls
    // vector2double() again at :1004, i.e. it marginalizes at the re-anchored state; we follow it exactly below.
    double2vector();
    vector2double();
    {
        uvs_window w; problem.fill(&w, feature_index + 1, line_feature_index + 1);
        std::memcpy(w.pose, para_Pose, sizeof(w.pose)); std::memcpy(w.speedbias, para_SpeedBias, sizeof(w.speedbias));
        MarginalizationInfo* marginalization_info = new MarginalizationInfo();
        // the factors are the ones uvs::Solve() just uploaded; only the (re-anchored) state goes to the device again
        const int rc = uvs_marginalize_resident(solver, &w, marginalization_flag == MARGIN_OLD ? 0 : 1, &marginalization_info->prior);
        if (rc == UVS_OK) { delete last_marginalization_info; last_marginalization_info = marginalization_info; }
        else delete marginalization_info;
    }
    // losses that were never attached to a residual block are not owned by the Problem
    if (problem.pt_lm.empty()) delete loss_function;
    if (problem.ln_lm.empty()) delete line_loss_function;
    bool any_vp = false; for (int v : problem.ln_has_vp) any_vp |= (v != 0);
    if (!any_vp) delete vp_loss_function;
}

// ====================================================================== per-frame state machine (post-initialization part)
void Estimator::clearState() {        // estimator.cpp:23-82 (the members this mirror has)
    for (int i = 0; i < WINDOW_SIZE + 1; i++) {
        Rs[i].setIdentity(); Ps[i].setZero(); Vs[i].setZero(); Bas[i].setZero(); Bgs[i].setZero();
        dt_buf[i].clear(); linear_acceleration_buf[i].clear(); angular_velocity_buf[i].clear();
        delete pre_integrations[i]; pre_integrations[i] = nullptr;
    }
    for (int i = 0; i < NUM_OF_CAM; i++) { tic[i] = Eigen::Vector3d::Zero(); ric[i] = Eigen::Matrix3d::Identity(); }
    solver_flag = INITIAL; first_imu = false; sum_of_back = 0; sum_of_front = 0; frame_count = 0; td = TD;
    delete last_marginalization_info; last_marginalization_info = nullptr;
    f_manager.clearState();
    failure_occur = 0;
    relocalization_info = 0;
    drift_correct_r = Eigen::Matrix3d::Identity(); drift_correct_t = Eigen::Vector3d::Zero();
}

void Estimator::processIMU(double dt, const Eigen::Vector3d& linear_acceleration, const Eigen::Vector3d& angular_velocity) {   // estimator.cpp:84-118
    if (!first_imu) { first_imu = true; acc_0 = linear_acceleration; gyr_0 = angular_velocity; }
    if (!pre_integrations[frame_count]) pre_integrations[frame_count] = new IntegrationBase{acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]};
    if (frame_count != 0) {
        pre_integrations[frame_count]->push_back(dt, linear_acceleration, angular_velocity);
        dt_buf[frame_count].push_back(dt);
        linear_acceleration_buf[frame_count].push_back(linear_acceleration);
        angular_velocity_buf[frame_count].push_back(angular_velocity);
        const int j = frame_count;
        Eigen::Vector3d un_acc_0 = Rs[j] * (acc_0 - Bas[j]) - G;
        Eigen::Vector3d un_gyr = (gyr_0 + angular_velocity) * 0.5 - Bgs[j];
        Rs[j] = Rs[j] * Utility::deltaQ(un_gyr * dt).toRotationMatrix();
        Eigen::Vector3d un_acc_1 = Rs[j] * (linear_acceleration - Bas[j]) - G;
        Eigen::Vector3d un_acc = (un_acc_0 + un_acc_1) * 0.5;
        Ps[j] = Ps[j] + Vs[j] * dt + un_acc * (0.5 * dt * dt);
        Vs[j] = Vs[j] + un_acc * dt;
    }
    acc_0 = linear_acceleration; gyr_0 = angular_velocity;
}

void Estimator::processImage(const FeatureManager::ImagePoints& image, const FeatureManager::ImageLines& image_line, const std_msgs::Header& header) {   // estimator.cpp:120-222
    marginalization_flag = f_manager.addFeatureCheckParallax(frame_count, image, image_line, td) ? MARGIN_OLD : MARGIN_SECOND_NEW;
    Headers[frame_count] = header;
    // (all_image_frame / tmp_pre_integration feed initialStructure only; the ESTIMATE_EXTRINSIC == 2 rotation calibration is initialization too)
    if (solver_flag == INITIAL) {      // :161-190
        if (frame_count == WINDOW_SIZE) {
            if (ESTIMATE_EXTRINSIC != 2 && initialStructure()) {
                solver_flag = NON_LINEAR;
                solveOdometry();
                slideWindow();
                f_manager.removeFailures();
                f_manager.removeLineFailures();
                last_R = Rs[WINDOW_SIZE]; last_P = Ps[WINDOW_SIZE]; last_R0 = Rs[0]; last_P0 = Ps[0];
            } else slideWindow();
        } else frame_count++;
        return;
    }
    solveOdometry();
    if (failureDetection()) { failure_occur = 1; clearState(); setParameter(); return; }
    slideWindow();
    f_manager.removeFailures();
    f_manager.removeLineFailures();
    key_poses.clear();
    for (int i = 0; i <= WINDOW_SIZE; i++) key_poses.push_back(Ps[i]);
    last_R = Rs[WINDOW_SIZE]; last_P = Ps[WINDOW_SIZE]; last_R0 = Rs[0]; last_P0 = Ps[0];
}

void Estimator::setInitialWindow(const double (*pose)[7], const double (*speedbias)[9]) {
    initial_window.assign(&pose[0][0], &pose[0][0] + 7 * (WINDOW_SIZE + 1));
    initial_window.insert(initial_window.end(), &speedbias[0][0], &speedbias[0][0] + 9 * (WINDOW_SIZE + 1));
}

bool Estimator::initialStructure() {
    // stand-in: installs what initialStructure() + visualInitialAlign() leave behind (estimator.cpp:370-446) -- window states,
    // pre-integrations re-propagated with the aligned biases (:385-390), depths cleared for re-triangulation (:392-398)
    if (initial_window.empty()) return false;
    const double* pose = initial_window.data(); const double* sb = pose + 7 * (WINDOW_SIZE + 1);
    for (int i = 0; i <= WINDOW_SIZE; ++i, pose += 7, sb += 9) {
        Ps[i] = Eigen::Vector3d(pose[0], pose[1], pose[2]);
        Rs[i] = Eigen::Quaterniond(pose[6], pose[3], pose[4], pose[5]).normalized().toRotationMatrix();
        Vs[i] = Eigen::Vector3d(sb[0], sb[1], sb[2]); Bas[i] = Eigen::Vector3d(sb[3], sb[4], sb[5]); Bgs[i] = Eigen::Vector3d(sb[6], sb[7], sb[8]);
        if (pre_integrations[i] && i > 0) pre_integrations[i]->repropagate(Bas[i], Bgs[i]);
    }
    for (auto& it : f_manager.feature) it.estimated_depth = -1;
    initial_window.clear();
    return true;
}

void Estimator::solveOdometry() {      // estimator.cpp:511-524
    if (frame_count < WINDOW_SIZE) return;
    if (solver_flag == NON_LINEAR) {
        f_manager.triangulate(Ps, tic, ric);
        f_manager.triangulateLine(Ps, Rs, tic, ric);
        optimization();
    }
}

bool Estimator::failureDetection() {   // estimator.cpp:713-760 (the checks that return true)
    if (Bas[WINDOW_SIZE].norm() > 2.5) return true;
    if (Bgs[WINDOW_SIZE].norm() > 1.0) return true;
    Eigen::Vector3d tmp_P = Ps[WINDOW_SIZE];
    if ((tmp_P - last_P).norm() > 5) return true;
    if (std::abs(tmp_P.z() - last_P.z()) > 1) return true;
    return false;
}

void Estimator::slideWindow() {        // estimator.cpp:1235-1331
    if (marginalization_flag == MARGIN_OLD) {
        back_R0 = Rs[0]; back_P0 = Ps[0];
        if (frame_count == WINDOW_SIZE) {
            for (int i = 0; i < WINDOW_SIZE; i++) {
                std::swap(Rs[i], Rs[i + 1]);
                std::swap(pre_integrations[i], pre_integrations[i + 1]);
                dt_buf[i].swap(dt_buf[i + 1]); linear_acceleration_buf[i].swap(linear_acceleration_buf[i + 1]); angular_velocity_buf[i].swap(angular_velocity_buf[i + 1]);
                Headers[i] = Headers[i + 1];
                std::swap(Ps[i], Ps[i + 1]); std::swap(Vs[i], Vs[i + 1]); std::swap(Bas[i], Bas[i + 1]); std::swap(Bgs[i], Bgs[i + 1]);
            }
            Headers[WINDOW_SIZE] = Headers[WINDOW_SIZE - 1];
            Ps[WINDOW_SIZE] = Ps[WINDOW_SIZE - 1]; Vs[WINDOW_SIZE] = Vs[WINDOW_SIZE - 1]; Rs[WINDOW_SIZE] = Rs[WINDOW_SIZE - 1];
            Bas[WINDOW_SIZE] = Bas[WINDOW_SIZE - 1]; Bgs[WINDOW_SIZE] = Bgs[WINDOW_SIZE - 1];
            delete pre_integrations[WINDOW_SIZE];
            pre_integrations[WINDOW_SIZE] = new IntegrationBase{acc_0, gyr_0, Bas[WINDOW_SIZE], Bgs[WINDOW_SIZE]};
            dt_buf[WINDOW_SIZE].clear(); linear_acceleration_buf[WINDOW_SIZE].clear(); angular_velocity_buf[WINDOW_SIZE].clear();
            slideWindowOld();
        }
    } else if (frame_count == WINDOW_SIZE) {
        for (unsigned int i = 0; i < dt_buf[frame_count].size(); i++) {
            const double tmp_dt = dt_buf[frame_count][i];
            const Eigen::Vector3d tmp_linear_acceleration = linear_acceleration_buf[frame_count][i], tmp_angular_velocity = angular_velocity_buf[frame_count][i];
            pre_integrations[frame_count - 1]->push_back(tmp_dt, tmp_linear_acceleration, tmp_angular_velocity);
            dt_buf[frame_count - 1].push_back(tmp_dt);
            linear_acceleration_buf[frame_count - 1].push_back(tmp_linear_acceleration);
            angular_velocity_buf[frame_count - 1].push_back(tmp_angular_velocity);
        }
        Headers[frame_count - 1] = Headers[frame_count];
        Ps[frame_count - 1] = Ps[frame_count]; Vs[frame_count - 1] = Vs[frame_count]; Rs[frame_count - 1] = Rs[frame_count];
        Bas[frame_count - 1] = Bas[frame_count]; Bgs[frame_count - 1] = Bgs[frame_count];
        delete pre_integrations[WINDOW_SIZE];
        pre_integrations[WINDOW_SIZE] = new IntegrationBase{acc_0, gyr_0, Bas[WINDOW_SIZE], Bgs[WINDOW_SIZE]};
        dt_buf[WINDOW_SIZE].clear(); linear_acceleration_buf[WINDOW_SIZE].clear(); angular_velocity_buf[WINDOW_SIZE].clear();
        slideWindowNew();
    }
}

void Estimator::slideWindowNew() { sum_of_front++; f_manager.removeFront(frame_count); f_manager.removeLineFront(frame_count); }      // :1333-1338

void Estimator::slideWindowOld() {     // :1340-1359
    sum_of_back++;
    if (solver_flag == NON_LINEAR) {
        Eigen::Matrix3d R0 = back_R0 * ric[0], R1 = Rs[0] * ric[0];
        Eigen::Vector3d P0 = back_P0 + back_R0 * tic[0], P1 = Ps[0] + Rs[0] * tic[0];
        f_manager.removeBackShiftDepth(R0, P0, R1, P1);
    } else f_manager.removeBack();
    f_manager.removeLineBack();
}
