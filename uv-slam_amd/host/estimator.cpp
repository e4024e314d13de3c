// estimator.cpp -- Estimator::optimization() / vector2double() / double2vector() (vins_estimator/src/estimator.cpp:526-711, 761-1233):
// the Ceres problem construction becomes a flat window descriptor (window_assembly.h), ceres::Solve one C-ABI call, the marginalization
// uvs_marginalize_resident().  See INTEGRATION.md for the diff a maintainer applies to the reference file.
#include <chrono>
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

Estimator::Estimator() : frame_count(0), first_imu(false), sum_of_back(0), sum_of_front(0), solver_flag(NON_LINEAR), marginalization_flag(MARGIN_OLD), td(0), failure_occur(false), last_marginalization_info(nullptr), relocalization_info(false), relo_frame_stamp(0), relo_frame_index(0), relo_frame_local_index(0), relo_relative_yaw(0), solver(nullptr), eval_solver(nullptr) {
    f_manager.Rs = Rs;      // estimator.cpp:9 `f_manager{Rs}`
    for (auto& p : pre_integrations) p = nullptr;
    for (int i = 0; i <= WINDOW_SIZE; ++i) Rs[i].setIdentity();
    uvs_options o; uvs_default_options(&o);
    o.estimate_td = ESTIMATE_TD; o.estimate_extrinsic = ESTIMATE_EXTRINSIC != 0;      // fixed for the lifetime of the handle, like the reference's globals (parameters.cpp)
    const int rc = uvs_create(&o, 0, 1, NUM_OF_F, NUM_OF_F * (WINDOW_SIZE + 1), NUM_OF_LF, NUM_OF_LF * (WINDOW_SIZE + 1), &solver);
    if (rc != UVS_OK) throw std::runtime_error(std::string("uvs_create: ") + uvs_status_string(rc));     // no CPU fallback
    // The factor classes' per-block Evaluate() gets a handle of its OWN (one-block windows): a call between the solve and the marginalization of optimization()
    // would otherwise replace the window that uvs_marginalize_resident() expects to find on the solver's handle.
    if (!uvs::evaluation_solver() && uvs_create(&o, 0, 1, 8, 16, 8, 16, &eval_solver) == UVS_OK) uvs::set_evaluation_solver(eval_solver);
}
Estimator::~Estimator() {
    if (eval_solver) { if (uvs::evaluation_solver() == eval_solver) uvs::set_evaluation_solver(nullptr); uvs_destroy(eval_solver); }
    finishMarginalization();
    uvs_destroy(solver); delete last_marginalization_info; for (auto* p : pre_integrations) delete p;
}

// ---- small helpers of this file: a 7-double parameter block (px py pz qx qy qz qw, estimator.cpp:530-537) <-> (translation, rotation)
namespace {
using Eigen::Matrix3d; using Eigen::Quaterniond; using Eigen::Vector3d;
inline void put_pose(double* blk, const Vector3d& t, const Matrix3d& R) {
    const Quaterniond q{R};
    const double v[7] = {t.x(), t.y(), t.z(), q.x(), q.y(), q.z(), q.w()};
    std::copy(v, v + 7, blk);
}
inline Vector3d vec3_at(const double* p) { return Vector3d(p[0], p[1], p[2]); }
inline Quaterniond quat_at(const double* blk) { return Quaterniond(blk[6], blk[3], blk[4], blk[5]); }
inline double yaw_of(const Matrix3d& R) { return Utility::R2ypr(R).x(); }      // degrees
// oldest element of a window array to the back, everything else one slot towards the front
template <class T, std::size_t N> inline void oldest_to_back(T (&a)[N]) { std::rotate(a, a + 1, a + N); }
}  // namespace

void Estimator::setParameter() {      // estimator.cpp:9-21: camera extrinsics, the pixel-noise weight of the projection factors, td
    std::copy(TIC.begin(), TIC.begin() + NUM_OF_CAM, tic);
    std::copy(RIC.begin(), RIC.begin() + NUM_OF_CAM, ric);
    td = TD;
    ProjectionFactor::sqrt_info = FOCAL_LENGTH / 1.6;
}

// Eigen state -> the flat parameter arrays the solver reads (estimator.cpp:526-594)
void Estimator::vector2double() {
    for (int f = 0; f <= WINDOW_SIZE; ++f) {
        put_pose(para_Pose[f], Ps[f], Rs[f]);
        double* sb = para_SpeedBias[f];
        for (int a = 0; a < 3; ++a) { sb[a] = Vs[f](a); sb[3 + a] = Bas[f](a); sb[6 + a] = Bgs[f](a); }
    }
    for (int cam = 0; cam < NUM_OF_CAM; ++cam) put_pose(para_Ex_Pose[cam], tic[cam], ric[cam]);
    if (ESTIMATE_TD) para_Td[0][0] = td;
    const Eigen::VectorXd inverse_depth = f_manager.getDepthVector();
    for (std::size_t k = 0; k < inverse_depth.size(); ++k) para_Feature[k][0] = inverse_depth[k];
    int l = 0;
    for (const Eigen::Vector4d& orth : f_manager.getLineOrthonormal()) { std::copy(orth.v, orth.v + 4, para_Ortho_plucker[l]); ++l; }
}

// Solver output -> Eigen state (estimator.cpp:596-711).  The problem is free in yaw and translation (4-dof gauge), so the result is
// first moved into the gauge in which the oldest pose keeps its pre-solve yaw and position: a yaw-only
// rotation, or the full relative rotation when a pitch sits within 1 degree of the Euler singularity (:616-625).
void Estimator::double2vector() {
    const Matrix3d R0_before = failure_occur ? last_R0 : Rs[0];
    const Vector3d P0_before = failure_occur ? last_P0 : Ps[0];
    failure_occur = false;
    const Matrix3d R0_after = quat_at(para_Pose[0]).toRotationMatrix();
    const Vector3d ypr_before = Utility::R2ypr(R0_before), ypr_after = Utility::R2ypr(R0_after);
    const auto at_euler_singularity = [](const Vector3d& ypr) { return std::abs(std::abs(ypr.y()) - 90) < 1.0; };
    const Matrix3d gauge = (at_euler_singularity(ypr_before) || at_euler_singularity(ypr_after))
                               ? Rs[0] * R0_after.transpose()
                               : Utility::ypr2R(Vector3d(ypr_before.x() - ypr_after.x(), 0, 0));
    const Vector3d p0_after = vec3_at(para_Pose[0]);
    const auto regauged_rotation = [&](const double* blk) { return Matrix3d(gauge * quat_at(blk).normalized().toRotationMatrix()); };
    const auto regauged_position = [&](const double* blk) { return Vector3d(gauge * (vec3_at(blk) - p0_after) + P0_before); };
    for (int f = 0; f <= WINDOW_SIZE; ++f) {
        const double* sb = para_SpeedBias[f];
        Rs[f] = regauged_rotation(para_Pose[f]);
        Ps[f] = regauged_position(para_Pose[f]);
        Vs[f] = gauge * vec3_at(sb);
        Bas[f] = vec3_at(sb + 3);
        Bgs[f] = vec3_at(sb + 6);
    }
    for (int cam = 0; cam < NUM_OF_CAM; ++cam) { tic[cam] = vec3_at(para_Ex_Pose[cam]); ric[cam] = quat_at(para_Ex_Pose[cam]).toRotationMatrix(); }
    if (ESTIMATE_TD) td = para_Td[0][0];
    // landmarks: inverse depths (negative depth marks the track as failed, feature_manager.cpp:235-253) and line parameters
    Eigen::VectorXd inverse_depth(f_manager.getFeatureCount());
    for (std::size_t k = 0; k < inverse_depth.size(); ++k) inverse_depth[k] = para_Feature[k][0];
    f_manager.setDepth(inverse_depth);
    std::vector<Eigen::Vector4d> orth(f_manager.getLineFeatureCount());
    for (std::size_t l = 0; l < orth.size(); ++l) std::copy(para_Ortho_plucker[l], para_Ortho_plucker[l] + 4, orth[l].v);
    f_manager.setLineOrtho(orth, Ps, Rs, tic[0], ric[0]);
    if (!relocalization_info) return;
    // loop-closure frame in the same gauge: drift of the odometry frame against the pose graph and the relative pose handed back to it (:671-691)
    relocalization_info = false;
    const Matrix3d loop_R = regauged_rotation(relo_Pose);
    const Vector3d loop_P = regauged_position(relo_Pose);
    const int m = relo_frame_local_index;
    drift_correct_r = Utility::ypr2R(Vector3d(yaw_of(prev_relo_r) - yaw_of(loop_R), 0, 0));
    drift_correct_t = prev_relo_t - drift_correct_r * loop_P;
    relo_relative_t = loop_R.transpose() * (Ps[m] - loop_P);
    relo_relative_q = Quaterniond(loop_R.transpose() * Rs[m]);
    relo_relative_yaw = Utility::normalizeAngle(yaw_of(Rs[m]) - yaw_of(loop_R));
}

// A loop-closure match arrived (estimator.cpp:1361-1379): remember its points and pose-graph pose; when the matched keyframe is still
// in the window, its current pose block seeds relo_Pose and the next optimization() adds the relocalization blocks.
void Estimator::setReloFrame(double stamp, int index, std::vector<Eigen::Vector3d>& points, Eigen::Vector3d pose_graph_t, Eigen::Matrix3d pose_graph_r) {
    prev_relo_r = pose_graph_r;
    prev_relo_t = pose_graph_t;
    match_points = points;
    relo_frame_index = index;
    relo_frame_stamp = stamp;
    const std_msgs::Header* hit = std::find_if(Headers, Headers + WINDOW_SIZE, [stamp](const std_msgs::Header& h) { return h.stamp.toSec() == stamp; });
    if (hit == Headers + WINDOW_SIZE) return;
    relo_frame_local_index = int(hit - Headers);
    std::copy(para_Pose[relo_frame_local_index], para_Pose[relo_frame_local_index] + SIZE_POSE, relo_Pose);
    relocalization_info = true;
}

namespace { double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } }

// Everything the solver reads besides the parameter values: the prior, the IMU links, and one index-addressed observation per residual
// block, in the order the landmark rows of para_Feature / para_Ortho_plucker were filled by vector2double().  Which tracks take part
// (FeatureManager::usedPoint / usedLine) and the emission order are behaviour of estimator.cpp:803-978 that the ABI relies on.
void Estimator::assembleWindow(uvs::WindowAssembly& wa) {      // (the prior is attached by optimization(), after the wait for the marginalization that produces it)
    for (int later = 1; later <= WINDOW_SIZE; ++later) {
        const IntegrationBase& pre = *pre_integrations[later];
        if (pre.sum_dt <= 10.0) wa.addImu(later - 1, pre);      // a pre-integration over more than 10 s is too uncertain to constrain anything
    }
    // loop-closure matches arrive sorted by feature id, like the track list: one forward cursor serves all tracks
    auto match = match_points.cbegin();
    const auto matches_end = relocalization_info ? match_points.cend() : match_points.cbegin();
    for (FeaturePerId& track : f_manager.feature) {
        if (!FeatureManager::usedPoint(track)) continue;
        const int landmark = wa.n_points;
        wa.addPointTrack(track, ESTIMATE_TD != 0);
        if (match == matches_end || track.start_frame > relo_frame_local_index) continue;
        match = std::find_if(match, matches_end, [&](const Eigen::Vector3d& m) { return (int)m.z() >= track.feature_id; });
        if (match != matches_end && (int)match->z() == track.feature_id) { wa.addReloMatch(landmark, track.feature_per_frame.front().point, match->x(), match->y()); ++match; }
    }
    for (LineFeaturePerId& track : f_manager.line_feature)
        if (FeatureManager::usedLine(track)) wa.addLineTrack(track);
}

// The window a marginalization in flight reads (uvs_solver.h: uvs_marginalize_resident_begin -- everything `w` points to must outlive the call): the assembly moves
// here (a moved std::vector keeps its buffer, so the descriptor's pointers stay good); inverse depths / line parameters are copied (the para_* members are public and
// vector2double() rewrites them); the OLD prior (w.prior = &last_marginalization_info->prior) is read in place: see the rule at pending_marginalization in estimator.h.
struct Estimator::PendingMarginalization {
    uvs::WindowAssembly wa; uvs_window w;
    std::vector<double> inv_depth, line_orth;      // the worker's OWN copies of the landmark parameters: the para_* members they come from are public and rewritten by vector2double()
    PendingMarginalization(uvs::WindowAssembly&& a, const uvs_window& v) : wa(std::move(a)), w(v) {
        inv_depth.assign(v.inv_depth, v.inv_depth + std::max(v.n_points, 0)); line_orth.assign(v.line_orth, v.line_orth + 4 * std::max(v.n_lines, 0));
        inv_depth.push_back(0.0); line_orth.push_back(0.0);      // (never empty: the descriptor wants non-null arrays)
        w.inv_depth = inv_depth.data(); w.line_orth = line_orth.data();
    }
};

void Estimator::finishMarginalization() {
    if (!pending_marginalization) return;
    const auto t0 = std::chrono::steady_clock::now();
    MarginalizationInfo* fresh = new MarginalizationInfo();
    const int rc = uvs_marginalize_wait(solver, &fresh->prior);
    delete last_marginalization_info;      // the input of the call that has just ended
    last_marginalization_info = fresh;
    if (rc != UVS_OK) {      // the reference has no error path here; an un-shifted old prior would attach to the wrong frames, so the prior is dropped
        std::fprintf(stderr, "Estimator::optimization: marginalization failed (%s: %s); continuing without a prior\n", uvs_status_string(rc), uvs_last_error(solver));
        delete last_marginalization_info; last_marginalization_info = nullptr;
    }
    delete pending_marginalization; pending_marginalization = nullptr;
    marginalize_ms += ms_since(t0);      // the part of the marginalization the caller had to wait for
}

void Estimator::optimization() {      // estimator.cpp:761-1233
    const auto t_begin = std::chrono::steady_clock::now();
    // packing the state and walking the track lists need nothing of the new prior: they run BESIDE the tail of the previous frame's marginalization (the worker reads its own copies of
    // the landmark parameters and the old prior, see PendingMarginalization), the wait comes after them
    vector2double();
    uvs::WindowAssembly wa;
    assembleWindow(wa);
    finishMarginalization();             // the prior of the previous frame (in a live system there already: it was computed beside slideWindow / processIMU / processImage)
    if (last_marginalization_info && last_marginalization_info->prior.n > 0) wa.prior = &last_marginalization_info->prior;
    uvs_window w = wa.view(para_Pose, para_SpeedBias, para_Ex_Pose[0], para_Td[0][0], &para_Feature[0][0], &para_Ortho_plucker[0][0], relo_Pose);
    // record hook (the reference has no serialisation): UVS_DUMP_WINDOWS=<dir> writes every window exactly as the
    // solver receives it (the state after vector2double(), estimator.cpp:800) to <dir>/window_NNNN.bin for replay without ROS
    if (const char* dump_dir = std::getenv("UVS_DUMP_WINDOWS")) {
        static int dump_index = 0;
        char name[32]; std::snprintf(name, sizeof(name), "/window_%04d.bin", dump_index++);
        WindowFile::save(std::string(dump_dir) + name, w, relo_frame_local_index);
    }
    {   // == ceres::Solve(options, &problem, &summary) at :992; the result goes back into the para_* arrays, which double2vector() reads
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<double> depths(std::max(wa.n_points, 1)), lines(4 * std::max(wa.n_lines, 1));
        uvs_state result; std::memset(&result, 0, sizeof(result));
        result.inv_depth = depths.data(); result.line_orth = lines.data();
        const bool multi = uvs::resolve_path(solver_path) == uvs::MULTI_WORKGROUP;      // (relocalization blocks beside a free extrinsic: both forms take them since round 6)
        last_summary.status = multi ? uvs_large_solve_fused(solver, &w, &result, &last_summary.report, nullptr) : uvs_solve_window(solver, &w, &result, &last_summary.report);
        if (last_summary.status == UVS_OK || (last_summary.status == UVS_ERR_NUMERIC && last_summary.report.num_iterations > 0)) {      // like the reference, nobody looks at the summary; a call that failed before the solve leaves the state alone
            std::memcpy(para_Pose, result.pose, sizeof(para_Pose)); std::memcpy(para_SpeedBias, result.speedbias, sizeof(para_SpeedBias));
            std::memcpy(para_Ex_Pose[0], result.ex_pose, sizeof(result.ex_pose));      // unchanged unless ESTIMATE_EXTRINSIC
            para_Td[0][0] = result.td;
            if (w.n_relo_obs > 0) std::memcpy(relo_Pose, result.relo_pose, sizeof(relo_Pose));
            for (int k = 0; k < wa.n_points; ++k) para_Feature[k][0] = depths[k];
            std::copy(lines.begin(), lines.begin() + 4 * wa.n_lines, &para_Ortho_plucker[0][0]);
        }
        solve_ms += ms_since(t0);
    }
    // ---- marginalization: the reference re-anchors first (double2vector) and packs again (vector2double at :1004 / :1167), i.e. it
    // marginalizes at the re-anchored state.  The factors are the ones the solve just uploaded; only that state goes to the device again.
    double2vector();
    vector2double();
    {
        // every by-value state field of the descriptor is refreshed: the solve moved para_Ex_Pose / para_Td too when they are estimated, and the prior
        // must be linearized at -- and remember as x0 -- the post-solve values (the reference packs again at :1004 before it marginalizes)
        std::memcpy(w.pose, para_Pose, sizeof(w.pose)); std::memcpy(w.speedbias, para_SpeedBias, sizeof(w.speedbias));
        std::memcpy(w.ex_pose, para_Ex_Pose[0], sizeof(w.ex_pose)); w.td = para_Td[0][0];
        const auto t0 = std::chrono::steady_clock::now();
        if (std::getenv("UVS_HOST_SYNC_MARGINALIZATION")) {      // the one-call form (diagnostics: its time is then all on the critical path)
            MarginalizationInfo* marginalization_info = new MarginalizationInfo();
            const int rc = uvs_marginalize_resident(solver, &w, marginalization_flag == MARGIN_OLD ? 0 : 1, &marginalization_info->prior);
            delete last_marginalization_info;
            last_marginalization_info = marginalization_info;
            if (rc != UVS_OK) {
                std::fprintf(stderr, "Estimator::optimization: marginalization failed (%s: %s); continuing without a prior\n", uvs_status_string(rc), uvs_last_error(solver));
                delete last_marginalization_info; last_marginalization_info = nullptr;
            }
        } else {
            // begin now, wait at the top of the next optimization(): the old prior (w.prior) and the observation arrays stay alive in pending_marginalization
            pending_marginalization = new PendingMarginalization(std::move(wa), w);
            const int rc = uvs_marginalize_resident_begin(solver, &pending_marginalization->w, marginalization_flag == MARGIN_OLD ? 0 : 1);
            if (rc != UVS_OK) {
                std::fprintf(stderr, "Estimator::optimization: marginalization could not start (%s: %s); continuing without a prior\n", uvs_status_string(rc), uvs_last_error(solver));
                delete pending_marginalization; pending_marginalization = nullptr;
                delete last_marginalization_info; last_marginalization_info = nullptr;
            }
        }
        marginalize_ms += ms_since(t0);
    }
    optimization_ms += ms_since(t_begin); ++optimization_calls;
}

// ====================================================================== per-frame state machine (post-initialization part)
// Written from the behaviour of estimator.cpp:23-222, 511-524, 713-760, 1235-1359 (what each call must leave behind), not from its text:
// the window arrays are rotated as a whole, IMU samples are merged sample by sample, and the point / line bookkeeping is one template.

void Estimator::clearState() {        // back to "nothing seen yet"
    for (IntegrationBase*& pre : pre_integrations) { delete pre; pre = nullptr; }
    for (auto& samples : dt_buf) samples.clear();
    for (auto& samples : linear_acceleration_buf) samples.clear();
    for (auto& samples : angular_velocity_buf) samples.clear();
    std::fill(Ps, Ps + WINDOW_SIZE + 1, Eigen::Vector3d::Zero());
    std::fill(Vs, Vs + WINDOW_SIZE + 1, Eigen::Vector3d::Zero());
    std::fill(Bas, Bas + WINDOW_SIZE + 1, Eigen::Vector3d::Zero());
    std::fill(Bgs, Bgs + WINDOW_SIZE + 1, Eigen::Vector3d::Zero());
    std::fill(Rs, Rs + WINDOW_SIZE + 1, Eigen::Matrix3d::Identity());
    std::fill(tic, tic + NUM_OF_CAM, Eigen::Vector3d::Zero());
    std::fill(ric, ric + NUM_OF_CAM, Eigen::Matrix3d::Identity());
    f_manager.clearState();
    finishMarginalization();
    delete last_marginalization_info;
    last_marginalization_info = nullptr;
    drift_correct_t = Eigen::Vector3d::Zero();
    drift_correct_r = Eigen::Matrix3d::Identity();
    frame_count = sum_of_back = sum_of_front = 0;
    first_imu = failure_occur = relocalization_info = false;
    solver_flag = INITIAL;
    td = TD;
}

// One IMU sample (estimator.cpp:84-118): it feeds the pre-integration of the newest frame, is kept for the re-propagation / merge of
// slideWindow(), and dead-reckons the newest frame's state (the initial guess of the next solve).
void Estimator::processIMU(double dt, const Eigen::Vector3d& acc, const Eigen::Vector3d& gyr) {
    if (!first_imu) { first_imu = true; acc_0 = acc; gyr_0 = gyr; }
    const int newest = frame_count;
    IntegrationBase*& pre = pre_integrations[newest];
    if (pre == nullptr) pre = new IntegrationBase{acc_0, gyr_0, Bas[newest], Bgs[newest]};
    if (newest > 0) {
        pre->push_back(dt, acc, gyr);
        recordSample(newest, dt, acc, gyr);
        deadReckon(newest, dt, acc, gyr);
    }
    acc_0 = acc;
    gyr_0 = gyr;
}
void Estimator::recordSample(int slot, double dt, const Eigen::Vector3d& acc, const Eigen::Vector3d& gyr) {
    dt_buf[slot].push_back(dt); linear_acceleration_buf[slot].push_back(acc); angular_velocity_buf[slot].push_back(gyr);
}
// midpoint rule in the world frame, gravity removed, biases of the frame itself: rotation first (mean body rate over the interval),
// then the mean of the world accelerations at both ends moves position and velocity
void Estimator::deadReckon(int f, double dt, const Eigen::Vector3d& acc, const Eigen::Vector3d& gyr) {
    const Eigen::Matrix3d R_begin = Rs[f];
    const Eigen::Vector3d rate = 0.5 * (gyr_0 + gyr) - Bgs[f];
    Rs[f] = R_begin * Utility::deltaQ(rate * dt).toRotationMatrix();
    const Eigen::Vector3d a_mean = 0.5 * ((R_begin * (acc_0 - Bas[f]) - G) + (Rs[f] * (acc - Bas[f]) - G));
    Ps[f] = Ps[f] + dt * Vs[f] + 0.5 * dt * dt * a_mean;
    Vs[f] = Vs[f] + dt * a_mean;
}

// One image: track bookkeeping + keyframe decision, then (window full) solve, failure check, slide (estimator.cpp:120-222).
// all_image_frame / tmp_pre_integration feed initialStructure() only and are not kept; ESTIMATE_EXTRINSIC == 2 is initialization too.
void Estimator::processImage(const FeatureManager::ImagePoints& image, const FeatureManager::ImageLines& image_line, const std_msgs::Header& header) {
    const bool second_newest_is_keyframe = f_manager.addFeatureCheckParallax(frame_count, image, image_line, td);
    marginalization_flag = second_newest_is_keyframe ? MARGIN_OLD : MARGIN_SECOND_NEW;
    Headers[frame_count] = header;
    const bool tracking = solver_flag == NON_LINEAR;
    if (!tracking) {
        if (frame_count < WINDOW_SIZE) { ++frame_count; return; }
        if (ESTIMATE_EXTRINSIC == 2 || !initialStructure()) { slideWindow(); return; }
        solver_flag = NON_LINEAR;
    }
    solveOdometry();
    if (tracking && failureDetection()) {      // restart from scratch (estimator.cpp:198-204): the flag is raised BEFORE clearState(), which lowers it
        failure_occur = true;                  // again (:77) -- so, as in the reference, the re-initialised window is NOT re-anchored on last_R0 / last_P0
        clearState();
        setParameter();
        return;
    }
    slideWindow();
    f_manager.removeFailures();
    f_manager.removeLineFailures();
    if (tracking) key_poses.assign(Ps, Ps + WINDOW_SIZE + 1);
    last_R0 = Rs[0]; last_P0 = Ps[0];
    last_R = Rs[WINDOW_SIZE]; last_P = Ps[WINDOW_SIZE];
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
    if (!(frame_count == WINDOW_SIZE && solver_flag == NON_LINEAR)) return;      // nothing to do until the window is full and initialised
    triangulateNewLandmarks();
    optimization();
}
void Estimator::triangulateNewLandmarks() { f_manager.triangulate(Ps, tic, ric); f_manager.triangulateLine(Ps, Rs, tic, ric); }

bool Estimator::failureDetection() {   // estimator.cpp:713-760, the tests that report a failure: runaway biases, a jump of the newest position
    const Eigen::Vector3d jump = Ps[WINDOW_SIZE] - last_P;
    const bool bias_blown = Bas[WINDOW_SIZE].norm() > 2.5 || Bgs[WINDOW_SIZE].norm() > 1.0;
    return bias_blown || jump.norm() > 5 || std::abs(jump.z()) > 1;
}

// The window moves on (estimator.cpp:1235-1331).  MARGIN_OLD: the oldest frame leaves, every slot moves one towards the front and the
// newest slot starts as a copy of its predecessor with an empty pre-integration.  MARGIN_SECOND_NEW: the second-newest frame is
// dropped -- its successor's IMU samples are appended to its pre-integration, and the newest state takes its slot.
void Estimator::slideWindow() {
    if (frame_count != WINDOW_SIZE) return;
    const int last = WINDOW_SIZE;
    const auto restart_newest_preintegration = [&] {
        delete pre_integrations[last];
        pre_integrations[last] = new IntegrationBase{acc_0, gyr_0, Bas[last], Bgs[last]};
        dt_buf[last].clear(); linear_acceleration_buf[last].clear(); angular_velocity_buf[last].clear();
    };
    if (marginalization_flag == MARGIN_OLD) {
        back_R0 = Rs[0];
        back_P0 = Ps[0];
        oldest_to_back(Ps); oldest_to_back(Vs); oldest_to_back(Rs); oldest_to_back(Bas); oldest_to_back(Bgs);
        oldest_to_back(Headers); oldest_to_back(pre_integrations);
        oldest_to_back(dt_buf); oldest_to_back(linear_acceleration_buf); oldest_to_back(angular_velocity_buf);
        Ps[last] = Ps[last - 1]; Vs[last] = Vs[last - 1]; Rs[last] = Rs[last - 1]; Bas[last] = Bas[last - 1]; Bgs[last] = Bgs[last - 1];
        Headers[last] = Headers[last - 1];
        restart_newest_preintegration();
        slideWindowOld();
        return;
    }
    const int kept = last - 1;
    for (std::size_t k = 0; k < dt_buf[last].size(); ++k) {
        pre_integrations[kept]->push_back(dt_buf[last][k], linear_acceleration_buf[last][k], angular_velocity_buf[last][k]);
        dt_buf[kept].push_back(dt_buf[last][k]);
        linear_acceleration_buf[kept].push_back(linear_acceleration_buf[last][k]);
        angular_velocity_buf[kept].push_back(angular_velocity_buf[last][k]);
    }
    Ps[kept] = Ps[last]; Vs[kept] = Vs[last]; Rs[kept] = Rs[last]; Bas[kept] = Bas[last]; Bgs[kept] = Bgs[last];
    Headers[kept] = Headers[last];
    restart_newest_preintegration();
    slideWindowNew();
}

void Estimator::slideWindowNew() {     // :1333-1338
    ++sum_of_front;
    f_manager.removeFront(frame_count);
    f_manager.removeLineFront(frame_count);
}

// :1340-1359 -- tracks anchored in the departed frame move their depth into the camera of the new oldest frame (only meaningful once
// depths exist, i.e. after initialization); lines keep their world parameters
void Estimator::slideWindowOld() {
    struct Camera { Eigen::Matrix3d R; Eigen::Vector3d t; };
    const auto camera_of = [this](const Eigen::Matrix3d& R_body, const Eigen::Vector3d& P_body) { return Camera{R_body * ric[0], P_body + R_body * tic[0]}; };
    sum_of_back += 1;
    if (solver_flag == NON_LINEAR) {
        const Camera departed = camera_of(back_R0, back_P0), oldest = camera_of(Rs[0], Ps[0]);
        f_manager.removeBackShiftDepth(departed.R, departed.t, oldest.R, oldest.t);
    } else f_manager.removeBack();
    f_manager.removeLineBack();
}
