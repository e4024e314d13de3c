// host_capi.cpp -- test hook: replay a recorded window through the mirrored Estimator::optimization().
// Builds the Estimator members (Ps/Rs/..., f_manager track lists, pre_integrations, last_marginalization_info) from a window
// file, runs optimization() (HIP solve + marginalization), and writes the resulting members back to a flat file.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include "estimator.h"
#include "window_io.h"

extern "C" int uvs_host_replay_window(const char* in_path, const char* out_path, int marg_flag) {
    WindowFile wf;
    if (!wf.load(in_path)) return -1;
    const uvs_window& w = wf.w;
    setEurocParameters();
    ESTIMATE_TD = wf.has_td ? 1 : 0;      // a window recorded with the ProjectionTdFactor inputs replays with ESTIMATE_TD (euroc_config.yaml: estimate_td)
    if (const char* e = std::getenv("UVS_HOST_ESTIMATE_EXTRINSIC")) ESTIMATE_EXTRINSIC = std::atoi(e);      // (euroc_config.yaml: estimate_extrinsic; the window file does not record it)
    try {
        Estimator est;
        est.td = w.td;
        est.setParameter();
        for (int i = 0; i <= WINDOW_SIZE; ++i) {
            est.Ps[i] = Eigen::Vector3d(w.pose[i][0], w.pose[i][1], w.pose[i][2]);
            est.Rs[i] = Eigen::Quaterniond(w.pose[i][6], w.pose[i][3], w.pose[i][4], w.pose[i][5]).toRotationMatrix();
            est.Vs[i] = Eigen::Vector3d(w.speedbias[i][0], w.speedbias[i][1], w.speedbias[i][2]);
            est.Bas[i] = Eigen::Vector3d(w.speedbias[i][3], w.speedbias[i][4], w.speedbias[i][5]);
            est.Bgs[i] = Eigen::Vector3d(w.speedbias[i][6], w.speedbias[i][7], w.speedbias[i][8]);
        }
        est.tic[0] = Eigen::Vector3d(w.ex_pose[0], w.ex_pose[1], w.ex_pose[2]);
        est.ric[0] = Eigen::Quaterniond(w.ex_pose[6], w.ex_pose[3], w.ex_pose[4], w.ex_pose[5]).toRotationMatrix();
        // tracks (consecutive frames from start_frame, as FeatureManager builds them)
        for (int k = 0, o = 0; k < w.n_points; ++k) {
            if (o >= w.n_point_obs || w.pt_lm[o] != k) return -2;
            FeaturePerId f(k, w.pt_fi[o]);
            // uv.y = ROW / 2 => row - ROW / 2 = 0: the recorded td_i / td_j already carry the rolling-shutter term
            auto ppf = [&](const double* p3, const double* v2, const double* tdp) {
                return wf.has_td ? FeaturePerFrame(Eigen::Vector3d(p3[0], p3[1], p3[2]), Eigen::Vector2d(0.0, ROW / 2), Eigen::Vector2d(v2[0], v2[1]), *tdp) : FeaturePerFrame(Eigen::Vector3d(p3[0], p3[1], p3[2])); };
            f.feature_per_frame.push_back(ppf(w.pt_pi + 3 * o, wf.has_td ? w.pt_vel_i + 2 * o : nullptr, wf.has_td ? w.pt_td_i + o : nullptr));
            int expect = w.pt_fi[o] + 1;
            while (o < w.n_point_obs && w.pt_lm[o] == k) { if (w.pt_fj[o] != expect++) return -3; f.feature_per_frame.push_back(ppf(w.pt_pj + 3 * o, wf.has_td ? w.pt_vel_j + 2 * o : nullptr, wf.has_td ? w.pt_td_j + o : nullptr)); ++o; }
            f.estimated_depth = 1.0 / w.inv_depth[k];
            est.f_manager.feature.push_back(f);
        }
        for (int l = 0, o = 0; l < w.n_lines; ++l) {
            if (o >= w.n_line_obs || w.ln_lm[o] != l) return -4;
            LineFeaturePerId f(l, w.ln_fj[o]);
            int expect = w.ln_fj[o];
            while (o < w.n_line_obs && w.ln_lm[o] == l) {
                if (w.ln_fj[o] != expect++) return -5;
                LineFeaturePerFrame pf;
                pf.start_point = Eigen::Vector3d(w.ln_sp[3 * o], w.ln_sp[3 * o + 1], w.ln_sp[3 * o + 2]); pf.end_point = Eigen::Vector3d(w.ln_ep[3 * o], w.ln_ep[3 * o + 1], w.ln_ep[3 * o + 2]);
                pf.vp = w.ln_has_vp[o] ? Eigen::Vector3d(w.ln_vp[3 * o], w.ln_vp[3 * o + 1], w.ln_vp[3 * o + 2]) : Eigen::Vector3d(0, 0, 0);
                f.line_feature_per_frame.push_back(pf); ++o;
            }
            f.orthonormal_vec = Eigen::Vector4d(w.line_orth[4 * l], w.line_orth[4 * l + 1], w.line_orth[4 * l + 2], w.line_orth[4 * l + 3]);
            est.f_manager.line_feature.push_back(f);
        }
        for (int j = 1; j <= WINDOW_SIZE; ++j) {      // every frame needs an IntegrationBase; blocks absent from the file are marked sum_dt > 10
            IntegrationBase* p = new IntegrationBase(Eigen::Vector3d(), Eigen::Vector3d(), Eigen::Vector3d(), Eigen::Vector3d());
            p->sum_dt = 11.0; est.pre_integrations[j] = p;
        }
        for (int b = 0; b < w.n_imu; ++b) {
            const uvs_imu_block& ib = w.imu[b]; IntegrationBase* p = est.pre_integrations[ib.frame_i + 1];
            p->sum_dt = ib.skip ? 11.0 : ib.sum_dt; p->delta_p = Eigen::Vector3d(ib.delta_p[0], ib.delta_p[1], ib.delta_p[2]); p->delta_v = Eigen::Vector3d(ib.delta_v[0], ib.delta_v[1], ib.delta_v[2]);
            p->delta_q = Eigen::Quaterniond(ib.delta_q[3], ib.delta_q[0], ib.delta_q[1], ib.delta_q[2]);
            p->linearized_ba = Eigen::Vector3d(ib.linearized_ba[0], ib.linearized_ba[1], ib.linearized_ba[2]); p->linearized_bg = Eigen::Vector3d(ib.linearized_bg[0], ib.linearized_bg[1], ib.linearized_bg[2]);
            std::memcpy(p->jacobian, ib.jacobian, sizeof(ib.jacobian)); std::memcpy(p->covariance, ib.covariance, sizeof(ib.covariance));
        }
        if (w.prior && w.prior->n > 0) { est.last_marginalization_info = new MarginalizationInfo(); est.last_marginalization_info->prior = *w.prior; }
        est.marginalization_flag = marg_flag ? Estimator::MARGIN_SECOND_NEW : Estimator::MARGIN_OLD;
        if (w.n_relo_obs > 0) {      // relocalization blocks: what Estimator::setReloFrame (estimator.cpp:1361-1379) is handed by the pose graph
            for (int i = 0; i <= WINDOW_SIZE; ++i) est.Headers[i].stamp.t = 100.0 + i;
            std::vector<Eigen::Vector3d> match_points;      // (x, y, feature_id); feature ids of this harness = landmark indices
            for (int q = 0; q < w.n_relo_obs; ++q) match_points.push_back(Eigen::Vector3d(w.relo_pj[3 * q], w.relo_pj[3 * q + 1], (double)w.relo_lm[q]));
            est.vector2double();                             // para_Pose as the previous optimization() left it
            // pose of the old keyframe in the pose-graph frame: any rigid transform works for the test; take a fixed yaw + shift
            Eigen::Matrix3d old_r = Utility::ypr2R(Eigen::Vector3d(30.0, 0.0, 0.0)); Eigen::Vector3d old_t(1.0, -2.0, 0.5);
            est.setReloFrame(100.0 + wf.relo_frame_local_index, 7, match_points, old_t, old_r);
            std::memcpy(est.relo_Pose, w.relo_pose, sizeof(est.relo_Pose));      // replay the recorded start value exactly
        }
        est.optimization();
        est.finishMarginalization();      // (the prior is read right below)
        FILE* f = std::fopen(out_path, "wb"); if (!f) return -6;
        double hdr[4] = {(double)est.last_summary.status, (double)est.last_summary.report.num_iterations, est.last_summary.report.initial_cost, est.last_summary.report.final_cost};
        std::fwrite(hdr, 8, 4, f);
        for (int i = 0; i <= WINDOW_SIZE; ++i) {
            Eigen::Quaterniond q(est.Rs[i]);
            double row[16] = {est.Ps[i].x(), est.Ps[i].y(), est.Ps[i].z(), q.x(), q.y(), q.z(), q.w(), est.Vs[i].x(), est.Vs[i].y(), est.Vs[i].z(),
                              est.Bas[i].x(), est.Bas[i].y(), est.Bas[i].z(), est.Bgs[i].x(), est.Bgs[i].y(), est.Bgs[i].z()};
            std::fwrite(row, 8, 16, f);
        }
        for (auto& it : est.f_manager.feature) { double d[2] = {it.estimated_depth, (double)it.solve_flag}; std::fwrite(d, 8, 2, f); }
        for (auto& it : est.f_manager.line_feature) { double d[5] = {it.orthonormal_vec[0], it.orthonormal_vec[1], it.orthonormal_vec[2], it.orthonormal_vec[3], (double)it.solve_flag}; std::fwrite(d, 8, 5, f); }
        const uvs_prior& p = est.last_marginalization_info ? est.last_marginalization_info->prior : uvs_prior();
        double pn = est.last_marginalization_info ? p.n : 0; std::fwrite(&pn, 8, 1, f);
        if (pn > 0) { std::fwrite(p.linearized_residuals, 8, p.n, f); std::fwrite(p.linearized_jacobians, 8, (size_t)p.n * p.n, f); }
        std::fwrite(&est.td, 8, 1, f);      // trailing: Estimator::td after double2vector (estimator.cpp:706-707)
        if (w.n_relo_obs > 0) {             // then the relocalization outputs of double2vector (estimator.cpp:671-691): 7 + 9 + 3 + 3 + 4 + 1 + 1 doubles
            std::fwrite(est.relo_Pose, 8, 7, f);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { const double v = est.drift_correct_r(r, c); std::fwrite(&v, 8, 1, f); }
            double t[3] = {est.drift_correct_t.x(), est.drift_correct_t.y(), est.drift_correct_t.z()}; std::fwrite(t, 8, 3, f);
            double rt[3] = {est.relo_relative_t.x(), est.relo_relative_t.y(), est.relo_relative_t.z()}; std::fwrite(rt, 8, 3, f);
            double rq[4] = {est.relo_relative_q.x(), est.relo_relative_q.y(), est.relo_relative_q.z(), est.relo_relative_q.w()}; std::fwrite(rq, 8, 4, f);
            double tail[2] = {est.relo_relative_yaw, est.relocalization_info ? 1.0 : 0.0}; std::fwrite(tail, 8, 2, f);
        }
        std::fclose(f);
        // side file <out>.x0: the linearization point the new prior remembers, per kept block {kind, frame, size, x0[9]} -- the test of "the prior is built
        // at the POST-solve extrinsic / time offset" (estimator.cpp:1004 packs again before marginalizing) reads it
        if (FILE* fx = std::fopen((std::string(out_path) + ".x0").c_str(), "wb")) {
            for (int b = 0; pn > 0 && b < p.n_blocks; ++b) {
                double rec[12] = {(double)p.block_kind[b], (double)p.block_frame[b], (double)p.block_size[b]};
                for (int k = 0; k < 9; ++k) rec[3 + k] = k < p.block_size[b] ? p.x0[p.x0_off[b] + k] : 0.0;
                std::fwrite(rec, 8, 12, fx);
            }
            std::fclose(fx);
        }
    } catch (const std::exception& e) { std::fprintf(stderr, "uvs_host_replay_window: %s\n", e.what()); return -7; }
    return 0;
}

// CPU-only probe used by the record/replay round-trip test: loads a window file and returns counts + checksums.
extern "C" int uvs_host_window_probe(const char* path, double* out /*[12]*/) {
    WindowFile wf;
    if (!wf.load(path)) return -1;
    const uvs_window& w = wf.w;
    out[0] = w.n_points; out[1] = w.n_point_obs; out[2] = w.n_lines; out[3] = w.n_line_obs; out[4] = w.n_imu; out[5] = w.prior ? w.prior->n : 0;
    double s = 0; for (int i = 0; i < 11; ++i) for (int k = 0; k < 7; ++k) s += w.pose[i][k]; out[6] = s;
    s = 0; for (int k = 0; k < 3 * w.n_point_obs; ++k) s += w.pt_pj[k]; out[7] = s;
    s = 0; for (int k = 0; k < 3 * w.n_line_obs; ++k) s += w.ln_vp[k] + w.ln_sp[k]; out[8] = s;
    s = 0; for (int b = 0; b < w.n_imu; ++b) for (int k = 0; k < 225; ++k) s += w.imu[b].covariance[k] * 1e6 + w.imu[b].jacobian[k]; out[9] = s;
    s = 0; if (w.prior) for (int k = 0; k < w.prior->n * w.prior->n; ++k) s += w.prior->linearized_jacobians[k]; out[10] = s;
    s = 0; for (int k = 0; k < w.n_point_obs; ++k) s += w.pt_lm[k] + 3 * w.pt_fi[k] + 7 * w.pt_fj[k]; out[11] = s;
    return 0;
}

// same for the relocalization section of a window file: count, relo_frame_local_index, checksums
extern "C" int uvs_host_window_probe_relo(const char* path, double* out /*[4]*/) {
    WindowFile wf;
    if (!wf.load(path)) return -1;
    const uvs_window& w = wf.w;
    out[0] = w.n_relo_obs; out[1] = wf.relo_frame_local_index;
    double s = 0; for (int k = 0; k < 7; ++k) s += w.relo_pose[k];
    for (int k = 0; k < 3 * w.n_relo_obs; ++k) s += w.relo_pi[k] + 2.0 * w.relo_pj[k];
    out[2] = s;
    s = 0; for (int k = 0; k < w.n_relo_obs; ++k) s += w.relo_lm[k]; out[3] = s;
    return 0;
}

// CPU-only hook for the FeatureManager producers: triangulate() / triangulateLine() on flat arrays.
//   poses[11][7] = (p, q xyzw) ; ex[7] ; point tracks: pt_start[n_pt], pt_nobs[n_pt], pt_obs (concatenated normalized-plane xyz) ;
//   line tracks: ln_start[n_ln], ln_nobs[n_ln], ln_sp / ln_ep (concatenated xyz).  depth_io[n_pt]: <= 0 means "not triangulated yet";
//   orth_io[n_ln][4]: orth[3] == 0 means "no parameters yet".  Both are updated in place exactly as the members would be.
extern "C" int uvs_host_triangulate(const double* poses, const double* ex, int n_pt, const int* pt_start, const int* pt_nobs, const double* pt_obs,
                                    int n_ln, const int* ln_start, const int* ln_nobs, const double* ln_sp, const double* ln_ep,
                                    double* depth_io, double* orth_io) {
    setEurocParameters();
    Eigen::Vector3d Ps[WINDOW_SIZE + 1]; Eigen::Matrix3d Rs[WINDOW_SIZE + 1];
    for (int i = 0; i <= WINDOW_SIZE; ++i) {
        Ps[i] = Eigen::Vector3d(poses[7 * i], poses[7 * i + 1], poses[7 * i + 2]);
        Rs[i] = Eigen::Quaterniond(poses[7 * i + 6], poses[7 * i + 3], poses[7 * i + 4], poses[7 * i + 5]).toRotationMatrix();
    }
    Eigen::Vector3d tic[1] = {Eigen::Vector3d(ex[0], ex[1], ex[2])};
    Eigen::Matrix3d ric[1] = {Eigen::Quaterniond(ex[6], ex[3], ex[4], ex[5]).toRotationMatrix()};
    FeatureManager fm(Rs);
    for (int k = 0, o = 0; k < n_pt; ++k) {
        FeaturePerId f(k, pt_start[k]);
        for (int q = 0; q < pt_nobs[k]; ++q, ++o) f.feature_per_frame.emplace_back(Eigen::Vector3d(pt_obs[3 * o], pt_obs[3 * o + 1], pt_obs[3 * o + 2]));
        f.estimated_depth = depth_io[k];
        fm.feature.push_back(f);
    }
    for (int l = 0, o = 0; l < n_ln; ++l) {
        LineFeaturePerId f(l, ln_start[l]);
        for (int q = 0; q < ln_nobs[l]; ++q, ++o) {
            LineFeaturePerFrame pf;
            pf.start_point = Eigen::Vector3d(ln_sp[3 * o], ln_sp[3 * o + 1], ln_sp[3 * o + 2]); pf.end_point = Eigen::Vector3d(ln_ep[3 * o], ln_ep[3 * o + 1], ln_ep[3 * o + 2]);
            f.line_feature_per_frame.push_back(pf);
        }
        f.orthonormal_vec = Eigen::Vector4d(orth_io[4 * l], orth_io[4 * l + 1], orth_io[4 * l + 2], orth_io[4 * l + 3]);
        fm.line_feature.push_back(f);
    }
    fm.triangulate(Ps, tic, ric);
    fm.triangulateLine(Ps, Rs, tic, ric);
    int k = 0; for (auto& it : fm.feature) depth_io[k++] = it.estimated_depth;
    int l = 0; for (auto& it : fm.line_feature) { for (int q = 0; q < 4; ++q) orth_io[4 * l + q] = it.orthonormal_vec[q]; ++l; }
    return 0;
}

// Closed-loop replay of a FRAME SEQUENCE through the mirrored per-frame state machine (processIMU / processImage / solveOdometry /
// slideWindow around optimization()): what `rosbag play` drives in the reference (estimator_node.cpp:352-509 feeds exactly these two
// calls), minus initialStructure() -- the sequence file carries the aligned initial window instead.
// File (all float64):  magic 0x55565351 ("UVSQ"), n_frames, pose0[11][7], speedbias0[11][9], then per frame:
//   stamp, n_imu, n_imu x (dt, acc[3], gyr[3]), n_pts, n_pts x (id, x, y, z, u, v, vx, vy), n_lines, n_lines x (id, 15 values of the line message)
// Output (float64): n_rows, then per frame processed after initialization 24 values:
//   frame, marginalization_flag, Ps[W](3), q(Rs[W]) xyzw(4), Vs[W](3), Bas[W](3), Bgs[W](3), initial cost, final cost, iterations, points, lines, status
// timing of the last uvs_host_replay_sequence(): mean milliseconds per Estimator::optimization() call -- {whole call, uvs::Solve alone
// (packing + H2D + k_solve + D2H), marginalization alone, number of calls}
static double g_replay_timing[4] = {0, 0, 0, 0};
extern "C" void uvs_host_replay_timing(double* out4) { for (int k = 0; k < 4; ++k) out4[k] = g_replay_timing[k]; }

extern "C" int uvs_host_replay_sequence(const char* in_path, const char* out_path) {
    FILE* f = std::fopen(in_path, "rb");
    if (!f) return -1;
    std::vector<double> d;
    { std::fseek(f, 0, SEEK_END); const long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET); d.resize(sz / 8); if (std::fread(d.data(), 8, d.size(), f) != d.size()) { std::fclose(f); return -1; } std::fclose(f); }
    size_t p = 0;
    auto next = [&]() -> double { return p < d.size() ? d[p++] : 0.0; };
    if (next() != (double)0x55565351) return -2;
    const int n_frames = (int)next();
    if (d.size() < 2 + 11 * 16) return -2;
    const double* pose0 = d.data() + p; p += 77;
    const double* sb0 = d.data() + p; p += 99;
    setEurocParameters();
    std::vector<double> out;
    // VINS_RESULT_PATH of the reference (utility/visualization.cpp:195-207: one line per solved frame, "stamp x y z qx qy qz qw", stamp with 9 decimals, the rest
    // with 6): UVS_VINS_RESULT_PATH names the file; tools/ate.py scores it against an EuRoC ground-truth data.csv
    FILE* tum = nullptr;
    // APPENDED to, as the reference does (std::ios::app, visualization.cpp:195): a second replay into the same path adds its lines behind the first one's.  The
    // position / attitude written are last_P / last_R = Ps[WINDOW_SIZE] / Rs[WINDOW_SIZE] as processImage() leaves them (what pubOdometry reads after the same call).
    if (const char* rp = std::getenv("UVS_VINS_RESULT_PATH")) {
        tum = std::fopen(rp, "a");
        if (!tum) std::fprintf(stderr, "uvs_host_replay_sequence: cannot open UVS_VINS_RESULT_PATH=%s for appending; no result file is written\n", rp);
    }
    struct TumCloser { FILE*& f; ~TumCloser() { if (f) std::fclose(f); } } tum_closer{tum};
    try {
        Estimator est;
        est.clearState();
        est.setParameter();
        est.setInitialWindow((const double (*)[7])pose0, (const double (*)[9])sb0);
        for (int fr = 0; fr < n_frames; ++fr) {
            std_msgs::Header header; header.stamp.t = next();
            const int n_imu = (int)next();
            for (int k = 0; k < n_imu; ++k) {
                const double dt = next(); double a[3], g[3];
                for (double& v : a) v = next();
                for (double& v : g) v = next();
                est.processIMU(dt, Eigen::Vector3d(a[0], a[1], a[2]), Eigen::Vector3d(g[0], g[1], g[2]));
            }
            FeatureManager::ImagePoints image; FeatureManager::ImageLines image_line;
            const int n_pts = (int)next();
            for (int k = 0; k < n_pts; ++k) {
                const int id = (int)next(); Eigen::Matrix<double, 7, 1> m;
                for (int q = 0; q < 7; ++q) m(q) = next();
                image[id].emplace_back(0, m);
            }
            const int n_lines = (int)next();
            for (int k = 0; k < n_lines; ++k) {
                const int id = (int)next(); Eigen::Matrix<double, 15, 1> m;
                for (int q = 0; q < 15; ++q) m(q) = next();
                image_line[id].push_back(m);
            }
            if (p > d.size()) return -3;
            const bool was_initial = est.solver_flag == Estimator::INITIAL;
            const int fc = est.frame_count;
            est.processImage(image, image_line, header);
            if (was_initial && fc < WINDOW_SIZE) continue;      // still filling the window
            if (est.solver_flag == Estimator::INITIAL) return -4;      // initialization did not happen / failure detection rebooted the estimator
            const Eigen::Quaterniond q(est.last_R);
            const uvs_report& rep = est.last_summary.report;
            const double row[24] = {(double)fr, (double)est.marginalization_flag, est.last_P.x(), est.last_P.y(), est.last_P.z(), q.x(), q.y(), q.z(), q.w(),
                                    est.Vs[WINDOW_SIZE].x(), est.Vs[WINDOW_SIZE].y(), est.Vs[WINDOW_SIZE].z(), est.Bas[WINDOW_SIZE].x(), est.Bas[WINDOW_SIZE].y(), est.Bas[WINDOW_SIZE].z(),
                                    est.Bgs[WINDOW_SIZE].x(), est.Bgs[WINDOW_SIZE].y(), est.Bgs[WINDOW_SIZE].z(), rep.initial_cost, rep.final_cost, (double)rep.num_iterations,
                                    (double)est.f_manager.getFeatureCount(), (double)est.f_manager.getLineFeatureCount(), (double)est.last_summary.status};
            out.insert(out.end(), row, row + 24);
            if (tum) std::fprintf(tum, "%.9f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n", header.stamp.t, est.last_P.x(), est.last_P.y(), est.last_P.z(), q.x(), q.y(), q.z(), q.w());
        }
        const double nc = est.optimization_calls > 0 ? est.optimization_calls : 1;
        g_replay_timing[0] = est.optimization_ms / nc; g_replay_timing[1] = est.solve_ms / nc; g_replay_timing[2] = est.marginalize_ms / nc; g_replay_timing[3] = est.optimization_calls;
    } catch (const std::exception& e) { std::fprintf(stderr, "uvs_host_replay_sequence: %s\n", e.what()); return -10; }
    FILE* g = std::fopen(out_path, "wb");
    if (!g) return -5;
    const double n_rows = (double)(out.size() / 24);
    std::fwrite(&n_rows, 8, 1, g); std::fwrite(out.data(), 8, out.size(), g); std::fclose(g);
    return 0;
}

// Test hook for the factor classes' per-block evaluation surface (host/factor/factors.h): evaluates, through the CLASS API, the first
// point observation, the first line observation (and its VP block when tagged), IMU block 0 and the prior of the window file, and
// returns the flat outputs  out = [pt r2 | J 2x7 2x7 2x7 2x1 | check() | ln r2 | J 2x7 2x4 | vp r1 | J 1x7 1x4 | imu r15 | J 15x7 15x9 15x7 15x9 |
// prior r[n] | J of kept block 0 (n x size0)].  Returns the number of doubles written, < 0 on failure.
extern "C" int uvs_host_factor_api_probe(const char* path, double* out, int cap) {
    WindowFile wf;
    if (!wf.load(path)) return -1;
    setEurocParameters();
    uvs_options o; uvs_default_options(&o);
    uvs_solver* s = nullptr;
    if (uvs_create(&o, 0, 1, 1000, 16000, 1000, 16000, &s) != UVS_OK) return -2;
    uvs::set_evaluation_solver(s);
    ProjectionFactor::sqrt_info = FOCAL_LENGTH / 1.6;
    const uvs_window& w = wf.w;
    std::vector<double> v;
    auto fail = [&](int code) { uvs::set_evaluation_solver(nullptr); uvs_destroy(s); return code; };
    const Eigen::Matrix3d ric = Eigen::Quaterniond(w.ex_pose[6], w.ex_pose[3], w.ex_pose[4], w.ex_pose[5]).toRotationMatrix();
    const Eigen::Vector3d tic(w.ex_pose[0], w.ex_pose[1], w.ex_pose[2]);
    if (w.n_point_obs > 0) {
        ProjectionFactor f(Eigen::Vector3d(w.pt_pi[0], w.pt_pi[1], w.pt_pi[2]), Eigen::Vector3d(w.pt_pj[0], w.pt_pj[1], w.pt_pj[2]));
        double lam = w.inv_depth[w.pt_lm[0]], r[2], J0[14], J1[14], J2[14], J3[2]; double* jac[4] = {J0, J1, J2, J3};
        double pi_[7], pj_[7], ex_[7]; std::memcpy(pi_, w.pose[w.pt_fi[0]], 56); std::memcpy(pj_, w.pose[w.pt_fj[0]], 56); std::memcpy(ex_, w.ex_pose, 56);
        double* params[4] = {pi_, pj_, ex_, &lam};
        if (!f.Evaluate(params, r, jac)) return fail(-3);
        v.insert(v.end(), r, r + 2); v.insert(v.end(), J0, J0 + 14); v.insert(v.end(), J1, J1 + 14); v.insert(v.end(), J2, J2 + 14); v.insert(v.end(), J3, J3 + 2);
        v.push_back(f.check(params));
    }
    if (w.n_line_obs > 0) {
        const Eigen::Vector3d sp(w.ln_sp[0], w.ln_sp[1], w.ln_sp[2]), ep(w.ln_ep[0], w.ln_ep[1], w.ln_ep[2]), vp(w.ln_vp[0], w.ln_vp[1], w.ln_vp[2]);
        const double* pose = w.pose[w.ln_fj[0]]; const double* line = w.line_orth + 4 * w.ln_lm[0];
        LineProjectionFactor lf(ric, tic, sp, ep);
        double r[2], r2[2], Jp[14], Jl[8];
        if (!lf(pose, line, r2) || !lf.EvaluateWithJacobians(pose, line, r, Jp, Jl) || r[0] != r2[0] || r[1] != r2[1]) return fail(-4);
        v.insert(v.end(), r, r + 2); v.insert(v.end(), Jp, Jp + 14); v.insert(v.end(), Jl, Jl + 8);
        if (w.ln_has_vp[0]) {
            VPProjectionFactor vf(ric, tic, sp, ep, vp);
            double rv[1], Jvp[7], Jvl[4];
            if (!vf.EvaluateWithJacobians(pose, line, rv, Jvp, Jvl)) return fail(-5);
            v.push_back(rv[0]); v.insert(v.end(), Jvp, Jvp + 7); v.insert(v.end(), Jvl, Jvl + 4);
        }
    }
    if (w.n_imu > 0) {
        const uvs_imu_block& b = w.imu[0];
        IntegrationBase pre(Eigen::Vector3d(), Eigen::Vector3d(), Eigen::Vector3d(b.linearized_ba[0], b.linearized_ba[1], b.linearized_ba[2]), Eigen::Vector3d(b.linearized_bg[0], b.linearized_bg[1], b.linearized_bg[2]));
        pre.sum_dt = b.sum_dt; pre.delta_p = Eigen::Vector3d(b.delta_p[0], b.delta_p[1], b.delta_p[2]); pre.delta_v = Eigen::Vector3d(b.delta_v[0], b.delta_v[1], b.delta_v[2]);
        pre.delta_q = Eigen::Quaterniond(b.delta_q[3], b.delta_q[0], b.delta_q[1], b.delta_q[2]);
        std::memcpy(pre.jacobian, b.jacobian, sizeof(pre.jacobian)); std::memcpy(pre.covariance, b.covariance, sizeof(pre.covariance));
        IMUFactor f(&pre);
        const int i = b.frame_i;
        const double* params[4] = {w.pose[i], w.speedbias[i], w.pose[i + 1], w.speedbias[i + 1]};
        double r[15]; std::vector<double> J0(105), J1(135), J2(105), J3(135); double* jac[4] = {J0.data(), J1.data(), J2.data(), J3.data()};
        if (!f.Evaluate(params, r, jac)) return fail(-6);
        v.insert(v.end(), r, r + 15);
        for (auto* J : {&J0, &J1, &J2, &J3}) v.insert(v.end(), J->begin(), J->end());
    }
    if (w.prior && w.prior->n > 0) {
        MarginalizationInfo info; info.prior = *w.prior;
        MarginalizationFactor f(&info);
        const uvs_prior& p = info.prior;
        std::vector<const double*> params(p.n_blocks);
        for (int b = 0; b < p.n_blocks; ++b)
            params[b] = p.block_kind[b] == UVS_BLOCK_POSE ? w.pose[p.block_frame[b]] : p.block_kind[b] == UVS_BLOCK_SPEEDBIAS ? w.speedbias[p.block_frame[b]] : p.block_kind[b] == UVS_BLOCK_EX_POSE ? w.ex_pose : &w.td;
        std::vector<double> r(p.n), J0((size_t)p.n * p.block_size[0]);
        std::vector<double*> jac(p.n_blocks, nullptr); jac[0] = J0.data();
        if (!f.Evaluate(params.data(), r.data(), jac.data())) return fail(-7);
        v.insert(v.end(), r.begin(), r.end()); v.insert(v.end(), J0.begin(), J0.end());
    }
    if ((int)v.size() > cap) return fail(-8);
    std::memcpy(out, v.data(), v.size() * sizeof(double));
    return fail((int)v.size());
}

// Test hook for ProjectionTdFactor::Evaluate / check (projection_td_factor.h:16-17): the first point observation of a window file that
// carries the time-offset inputs, through the CLASS API on a handle created with estimate_td = 1.
// out = [r2 | J 2x7 2x7 2x7 | J_lambda 2 | J_td 2 | check()]; returns the number of doubles written, < 0 on failure.
extern "C" int uvs_host_td_factor_probe(const char* path, double* out, int cap) {
    WindowFile wf;
    if (!wf.load(path) || !wf.has_td || wf.w.n_point_obs < 1 || cap < 2 + 42 + 2 + 2 + 1) return -1;
    setEurocParameters();
    uvs_options o; uvs_default_options(&o); o.estimate_td = 1;
    uvs_solver* s = nullptr;
    if (uvs_create(&o, 0, 1, 1000, 16000, 1000, 16000, &s) != UVS_OK) return -2;
    uvs::set_evaluation_solver(s);
    ProjectionFactor::sqrt_info = FOCAL_LENGTH / 1.6;
    const uvs_window& w = wf.w;
    // TR = 0 in the EuRoC parameters, so the row arguments drop out and the file's folded capture offsets are td_i / td_j themselves
    ProjectionTdFactor f(Eigen::Vector3d(w.pt_pi[0], w.pt_pi[1], w.pt_pi[2]), Eigen::Vector3d(w.pt_pj[0], w.pt_pj[1], w.pt_pj[2]),
                         Eigen::Vector2d(w.pt_vel_i[0], w.pt_vel_i[1]), Eigen::Vector2d(w.pt_vel_j[0], w.pt_vel_j[1]), w.pt_td_i[0], w.pt_td_j[0], ROW / 2, ROW / 2);
    double lam = w.inv_depth[w.pt_lm[0]], td = w.td, r[2], J0[14], J1[14], J2[14], J3[2], J4[2]; double* jac[5] = {J0, J1, J2, J3, J4};
    double pi_[7], pj_[7], ex_[7]; std::memcpy(pi_, w.pose[w.pt_fi[0]], 56); std::memcpy(pj_, w.pose[w.pt_fj[0]], 56); std::memcpy(ex_, w.ex_pose, 56);
    double* params[5] = {pi_, pj_, ex_, &lam, &td};
    int n = -3;
    if (f.Evaluate(params, r, jac)) {
        std::vector<double> v(r, r + 2);
        v.insert(v.end(), J0, J0 + 14); v.insert(v.end(), J1, J1 + 14); v.insert(v.end(), J2, J2 + 14); v.insert(v.end(), J3, J3 + 2); v.insert(v.end(), J4, J4 + 2);
        v.push_back(f.check(params));
        std::memcpy(out, v.data(), v.size() * sizeof(double)); n = (int)v.size();
    }
    uvs::set_evaluation_solver(nullptr); uvs_destroy(s);
    return n;
}
