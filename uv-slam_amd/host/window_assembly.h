// window_assembly.h -- the flat window descriptor Estimator::optimization() hands to the C ABI ("inside, instead of
// ceres::Problem, it fills a flat window descriptor").  What the reference expresses as ~1 300 heap-allocated cost functions attached to
// parameter-block ADDRESSES (estimator.cpp:763-978) is, at this boundary, a handful of index-addressed arrays:
//     IMU blocks            frame_i -> frame_i + 1                                 (one per pre-integration with sum_dt <= 10 s, :811-818)
//     point observations    (landmark, anchor frame, observing frame, pts_i, pts_j [, image velocities and capture time offsets])  (:823-866)
//     line observations     (landmark, observing frame, sp, ep [, vp])             (every view of a used line, the anchor included, :868-927)
//     relocalization blocks (landmark, pts_i, match point)                          (:944-978)
// `WindowAssembly` owns those arrays while the uvs_window that points into them is in use; the landmark index is the running count of
// used tracks in list order, i.e. the row of para_Feature / para_Ortho_plucker that vector2double() filled (feature_manager.cpp:290-331).
#pragma once
#include <cstdlib>
#include <cstring>
#include <vector>
#include "feature_manager.h"
#include "integration_base.h"
#include "../../include/uvs_solver.h"

namespace uvs {

// How ONE window is solved (both forms run the same LM controller):
//   PERSISTENT_KERNEL  uvs_solve_window(): the whole solve in one workgroup on one compute unit (what a batch of windows uses per window)
//   MULTI_WORKGROUP    uvs_large_solve_fused(): landmark chunks on many compute units, one workgroup for the reduced solve, control on
//                      the device -- lower latency for a single window on an otherwise idle GPU
//   AUTO               MULTI_WORKGROUP; UVS_HOST_SOLVER_PATH=persistent|multi overrides
enum SolverPath { AUTO = 0, PERSISTENT_KERNEL, MULTI_WORKGROUP };
struct Summary { uvs_report report; int status = 0; int iterations() const { return report.num_iterations; } };

struct WindowAssembly {
    std::vector<uvs_imu_block> imu;
    std::vector<int32_t> pt_lm, pt_fi, pt_fj;  std::vector<double> pt_pi, pt_pj, pt_vel_i, pt_vel_j, pt_td_i, pt_td_j;
    std::vector<int32_t> ln_lm, ln_fj, ln_has_vp;  std::vector<double> ln_sp, ln_ep, ln_vp;
    std::vector<int32_t> relo_lm;  std::vector<double> relo_pi, relo_pj;
    const uvs_prior* prior = nullptr;
    int n_points = 0, n_lines = 0;      // landmarks = used tracks

    static void put3(std::vector<double>& dst, const Eigen::Vector3d& v) { dst.push_back(v(0)); dst.push_back(v(1)); dst.push_back(v(2)); }

    // pre-integration between frame_i and frame_i + 1 in the ABI's row-major layout (IntegrationBase keeps its 15 x 15 matrices row-major already)
    void addImu(int frame_i, const IntegrationBase& pre) {
        uvs_imu_block b; std::memset(&b, 0, sizeof(b));
        b.frame_i = frame_i; b.sum_dt = pre.sum_dt;
        for (int a = 0; a < 3; ++a) { b.delta_p[a] = pre.delta_p(a); b.delta_v[a] = pre.delta_v(a); b.linearized_ba[a] = pre.linearized_ba(a); b.linearized_bg[a] = pre.linearized_bg(a); }
        const double q[4] = {pre.delta_q.x(), pre.delta_q.y(), pre.delta_q.z(), pre.delta_q.w()};
        std::memcpy(b.delta_q, q, sizeof(q));
        std::memcpy(b.jacobian, pre.jacobian, sizeof(b.jacobian));
        std::memcpy(b.covariance, pre.covariance, sizeof(b.covariance));
        imu.push_back(b);
    }
    // a used point track becomes landmark `n_points`: one observation per view after the anchor view.  `time_offset` (ESTIMATE_TD):
    // also the image-plane velocities and each view's capture offset  cur_td - TR / ROW * (row - ROW / 2)  (projection_td_factor.cpp:6-16, 51-52)
    void addPointTrack(const FeaturePerId& track, bool time_offset) {
        const std::vector<FeaturePerFrame>& views = track.feature_per_frame;
        const FeaturePerFrame& anchor = views.front();
        const auto capture_offset = [](const FeaturePerFrame& v) { return v.cur_td - TR / ROW * (v.uv.y() - ROW / 2); };
        for (std::size_t v = 1; v < views.size(); ++v) {
            pt_lm.push_back(n_points); pt_fi.push_back(track.start_frame); pt_fj.push_back(track.start_frame + (int)v);
            put3(pt_pi, anchor.point); put3(pt_pj, views[v].point);
            if (!time_offset) continue;
            pt_vel_i.push_back(anchor.velocity.x()); pt_vel_i.push_back(anchor.velocity.y());
            pt_vel_j.push_back(views[v].velocity.x()); pt_vel_j.push_back(views[v].velocity.y());
            pt_td_i.push_back(capture_offset(anchor)); pt_td_j.push_back(capture_offset(views[v]));
        }
        ++n_points;
    }
    // a used line track becomes landmark `n_lines`: every view observes it; a view carries a vanishing-point residual when the front end
    // tagged it (vp.z == 1, line_feature_tracker.cpp:379-385)
    void addLineTrack(const LineFeaturePerId& track) {
        int frame = track.start_frame;
        for (const LineFeaturePerFrame& view : track.line_feature_per_frame) {
            const bool tagged = view.vp(2) == 1;
            ln_lm.push_back(n_lines); ln_fj.push_back(frame++); ln_has_vp.push_back(tagged ? 1 : 0);
            put3(ln_sp, view.start_point); put3(ln_ep, view.end_point); put3(ln_vp, tagged ? view.vp : Eigen::Vector3d(0, 0, 0));
        }
        ++n_lines;
    }
    void addReloMatch(int landmark, const Eigen::Vector3d& pts_i, double match_x, double match_y) {
        relo_lm.push_back(landmark); put3(relo_pi, pts_i); put3(relo_pj, Eigen::Vector3d(match_x, match_y, 1.0));
    }

    // the C-ABI view of the assembly; parameter values are read straight from the caller's para_* arrays (valid while both live)
    uvs_window view(const double (*pose)[7], const double (*speedbias)[9], const double* ex_pose, double td, const double* inv_depth, const double* line_orth, const double* relo_pose) const {
        uvs_window w; std::memset(&w, 0, sizeof(w));
        std::memcpy(w.pose, pose, sizeof(w.pose)); std::memcpy(w.speedbias, speedbias, sizeof(w.speedbias)); std::memcpy(w.ex_pose, ex_pose, sizeof(w.ex_pose));
        w.td = td; w.prior = prior;
        w.n_imu = (int)imu.size(); w.imu = imu.data();
        w.n_points = n_points; w.inv_depth = inv_depth; w.n_point_obs = (int)pt_lm.size();
        w.pt_lm = pt_lm.data(); w.pt_fi = pt_fi.data(); w.pt_fj = pt_fj.data(); w.pt_pi = pt_pi.data(); w.pt_pj = pt_pj.data();
        if (!pt_td_i.empty()) { w.pt_vel_i = pt_vel_i.data(); w.pt_vel_j = pt_vel_j.data(); w.pt_td_i = pt_td_i.data(); w.pt_td_j = pt_td_j.data(); }
        w.n_lines = n_lines; w.line_orth = line_orth; w.n_line_obs = (int)ln_lm.size();
        w.ln_lm = ln_lm.data(); w.ln_fj = ln_fj.data(); w.ln_sp = ln_sp.data(); w.ln_ep = ln_ep.data(); w.ln_has_vp = ln_has_vp.data(); w.ln_vp = ln_vp.data();
        if (!relo_lm.empty()) { w.n_relo_obs = (int)relo_lm.size(); w.relo_lm = relo_lm.data(); w.relo_pi = relo_pi.data(); w.relo_pj = relo_pj.data(); std::memcpy(w.relo_pose, relo_pose, sizeof(w.relo_pose)); }
        return w;
    }
};

inline SolverPath resolve_path(SolverPath requested) {
    if (const char* env = std::getenv("UVS_HOST_SOLVER_PATH")) requested = env[0] == 'p' ? PERSISTENT_KERNEL : env[0] == 'm' ? MULTI_WORKGROUP : requested;
    return requested == AUTO ? MULTI_WORKGROUP : requested;
}

}  // namespace uvs
