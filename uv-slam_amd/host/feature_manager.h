// feature_manager.h -- the part of FeatureManager (vins_estimator/src/feature_manager.{h,cpp}) that Estimator::optimization()
// and vector2double()/double2vector() touch: the per-id track lists and the get/set of solver parameters.
// SURVEY.md 8f row 3 (the producers / consumers either side of the solve) is mirrored too: triangulate (:427-481),
// triangulateLine (:504-589) with calcPluckerLine (:827-902), getDepthVector / getLineOrthonormal (:290-331), setDepth (:235-253),
// setLineOrtho (:333-423); and, for closed-loop replay of a frame sequence, the track bookkeeping either side of the solve:
// addFeatureCheckParallax (:73-158) with compensatedParallax2 (:727-760), removeFailures / removeLineFailures (:255-276),
// removeBackShiftDepth / removeBack / removeFront (:607-685), removeLineBack / removeLineFront (:687-725).
#pragma once
#include <algorithm>
#include <cmath>
#include <list>
#include <map>
#include <vector>
#include "parameters.h"
#include "utility.h"

class FeaturePerFrame {      // feature_manager.h:18-43: point = normalised (x, y, 1), uv = pixel, velocity = image-plane velocity, cur_td = td at capture
  public:
    FeaturePerFrame(const Eigen::Vector3d& p) : point(p), cur_td(0) {}
    FeaturePerFrame(const Eigen::Matrix<double, 7, 1>& _point, double td) : point(_point(0), _point(1), _point(2)), uv(_point(3), _point(4)), velocity(_point(5), _point(6)), cur_td(td) {}   // feature_manager.h:139-150
    FeaturePerFrame(const Eigen::Vector3d& p, const Eigen::Vector2d& _uv, const Eigen::Vector2d& _velocity, double td) : point(p), uv(_uv), velocity(_velocity), cur_td(td) {}
    Eigen::Vector3d point; Eigen::Vector2d uv, velocity; double cur_td;
};
class FeaturePerId {
  public:
    const int feature_id; int start_frame; std::vector<FeaturePerFrame> feature_per_frame; int used_num; double estimated_depth; int solve_flag;
    FeaturePerId(int id, int start) : feature_id(id), start_frame(start), used_num(0), estimated_depth(-1.0), solve_flag(0) {}
    int endFrame() const { return start_frame + (int)feature_per_frame.size() - 1; }      // feature_manager.cpp:8-11
};
class LineFeaturePerFrame {
  public:
    LineFeaturePerFrame() {}
    LineFeaturePerFrame(const Eigen::Matrix<double, 15, 1>& _point, double /*td*/)      // feature_manager.h:32-53: (sp.xy, ep.xy, uv x4, velocities x4, vp.xyz); only sp/ep/vp reach the solve
        : start_point(_point(0), _point(1), 1.0), end_point(_point(2), _point(3), 1.0), vp(_point(12), _point(13), _point(14)) {}
    Eigen::Vector3d start_point, end_point, vp;
};
class LineFeaturePerId {
  public:
    const int feature_id; int start_frame; std::vector<LineFeaturePerFrame> line_feature_per_frame; int used_num; Eigen::Vector4d orthonormal_vec; int solve_flag;
    LineFeaturePerId(int id, int start) : feature_id(id), start_frame(start), used_num(0), solve_flag(0) {}
    int endFrame() const { return start_frame + (int)line_feature_per_frame.size() - 1; }      // feature_manager.cpp:3-6
};

// eigenvector of the smallest eigenvalue of a symmetric 4x4 (cyclic Jacobi): the right singular vector the reference takes from
// JacobiSVD(svd_A, ComputeThinV).matrixV().rightCols<1>() is that eigenvector of svd_A^T svd_A (up to sign, which cancels in v2/v3)
inline void uvs_smallest_eigvec4(double A[4][4], double v[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0; for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
            if (A[p][q] == 0.0) continue;
            const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
            const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0)), c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
            for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
            for (int k = 0; k < 4; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
            for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
        }
    }
    int m = 0; for (int k = 1; k < 4; ++k) if (A[k][k] < A[m][m]) m = k;
    for (int k = 0; k < 4; ++k) v[k] = V[k][m];
}

class FeatureManager {
  public:
    std::list<FeaturePerId> feature;
    std::list<LineFeaturePerId> line_feature;
    const Eigen::Matrix3d* Rs = nullptr;            // the estimator's Rs[] (feature_manager.h: `const Matrix3d *Rs`, set by the constructor)
    FeatureManager() {}
    explicit FeatureManager(Eigen::Matrix3d _Rs[]) : Rs(_Rs) {}

    // feature_manager.cpp:427-481 -- linear (DLT) triangulation of every used point that has no depth yet, in its start frame
    void triangulate(Eigen::Vector3d Ps[], Eigen::Vector3d tic[], Eigen::Matrix3d ric[]) {
        using namespace Eigen;
        for (auto& it : feature) {
            if (!usedPoint(it)) continue;
            if (it.estimated_depth > 0) continue;
            const int imu_i = it.start_frame;
            const Vector3d t0 = Ps[imu_i] + Rs[imu_i] * tic[0];
            const Matrix3d R0 = Rs[imu_i] * ric[0];
            double AtA[4][4] = {{0}};
            int imu_j = imu_i - 1;
            for (auto& pf : it.feature_per_frame) {
                ++imu_j;
                const Vector3d t1 = Ps[imu_j] + Rs[imu_j] * tic[0];
                const Matrix3d R1 = Rs[imu_j] * ric[0];
                const Vector3d t = R0.transpose() * (t1 - t0);
                const Matrix3d Rt = (R0.transpose() * R1).transpose();      // P = [R^T | -R^T t]
                const Vector3d mt = -(Rt * t);
                const double n = pf.point.norm();
                const Vector3d f = pf.point / n;
                double P[3][4];
                for (int r = 0; r < 3; ++r) { for (int cidx = 0; cidx < 3; ++cidx) P[r][cidx] = Rt(r, cidx); P[r][3] = mt(r); }
                double row[2][4];
                for (int cidx = 0; cidx < 4; ++cidx) { row[0][cidx] = f[0] * P[2][cidx] - f[2] * P[0][cidx]; row[1][cidx] = f[1] * P[2][cidx] - f[2] * P[1][cidx]; }
                for (int r = 0; r < 2; ++r) for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) AtA[a][b] += row[r][a] * row[r][b];
            }
            double v[4]; uvs_smallest_eigvec4(AtA, v);
            it.estimated_depth = v[2] / v[3];
            if (it.estimated_depth < 0.1) it.estimated_depth = INIT_DEPTH;
        }
    }

    // feature_manager.cpp:827-902 -- the line through two back-projected planes, as the dual Plucker matrix pi1 pi2^T - pi2 pi1^T:
    // plane = [sp x ep ; -(sp x ep) . origin]; direction = (L(2,1), L(0,2), L(1,0)) = b x a, normal = (L(0,3), L(1,3), L(2,3)) = a*beta - b*alpha
    static void calcPluckerLine(const Eigen::Vector3d& prev_sp, const Eigen::Vector3d& prev_ep, const Eigen::Vector3d& curr_sp, const Eigen::Vector3d& curr_ep,
                                const Eigen::Vector3d& origin_prev, const Eigen::Vector3d& origin_curr, Eigen::Vector3d& out_direction, Eigen::Vector3d& out_normal,
                                Eigen::Vector4d& prev_plane, Eigen::Vector4d& curr_plane) {
        const Eigen::Vector3d a = prev_sp.cross(prev_ep), b = curr_sp.cross(curr_ep);
        const double alpha = -a.dot(origin_prev), beta = -b.dot(origin_curr);
        prev_plane = Eigen::Vector4d(a(0), a(1), a(2), alpha); curr_plane = Eigen::Vector4d(b(0), b(1), b(2), beta);
        out_direction = b.cross(a);
        out_normal = a * beta - b * alpha;
    }

    // Eigen 3.3 Matrix3d::eulerAngles(0, 1, 2) (the branch the reference gets at feature_manager.cpp:584): R = Rx(a) Ry(b) Rz(c), a in [0, pi]
    static Eigen::Vector3d eulerAnglesXYZ(const Eigen::Matrix3d& m) {
        const double pi = 3.14159265358979323846;
        double r0 = std::atan2(m(1, 2), m(2, 2)), r1;
        const double c2 = std::sqrt(m(0, 0) * m(0, 0) + m(0, 1) * m(0, 1));
        if (r0 > 0.0) { r0 -= pi; r1 = std::atan2(-m(0, 2), -c2); } else r1 = std::atan2(-m(0, 2), c2);
        const double s1 = std::sin(r0), c1 = std::cos(r0);
        const double r2 = std::atan2(s1 * m(2, 0) - c1 * m(1, 0), c1 * m(1, 1) - s1 * m(2, 1));
        return Eigen::Vector3d(-r0, -r1, -r2);
    }

    // feature_manager.cpp:504-589 -- two-view line triangulation (first and last observation) of every line without parameters yet
    // (orthonormal_vec[3] == 0).  The reference's cv::Mat argument only feeds a debug drawing and is dropped.
    void triangulateLine(Eigen::Vector3d Ps[], Eigen::Matrix3d /*Rs_estimate*/[], Eigen::Vector3d tic[], Eigen::Matrix3d ric[]) {
        using namespace Eigen;
        for (auto& it : line_feature) {
            it.used_num = (int)it.line_feature_per_frame.size();
            if (it.orthonormal_vec[3] != 0 || it.line_feature_per_frame.size() < 2) continue;
            const int imu_i = it.start_frame, imu_j = it.start_frame + it.used_num - 1;
            const Matrix3d R_left = Rs[imu_i] * ric[0], R_right = Rs[imu_j] * ric[0];
            const Vector3d t_left = Rs[imu_i] * tic[0] + Ps[imu_i], t_right = Rs[imu_j] * tic[0] + Ps[imu_j];
            const Matrix3d R_rel = R_left.transpose() * R_right;
            const Vector3d t_rel = R_left.transpose() * (t_right - t_left);
            const Vector3d left_sp = it.line_feature_per_frame[0].start_point, left_ep = it.line_feature_per_frame[0].end_point;
            const Vector3d right_sp_l = R_rel * it.line_feature_per_frame[it.used_num - 1].start_point, right_ep_l = R_rel * it.line_feature_per_frame[it.used_num - 1].end_point;
            Vector3d direction_l, normal_l; Vector4d left_plane, right_plane;
            calcPluckerLine(left_sp, left_ep, right_sp_l, right_ep_l, Vector3d(0, 0, 0), t_rel, direction_l, normal_l, left_plane, right_plane);
            // line_w = T_wl line_l with T_wl = [R [t]x R ; 0 R]
            const Vector3d d_w = R_left * direction_l;
            const Vector3d n_w = R_left * normal_l + t_left.cross(d_w);
            const Vector3d u0 = n_w / n_w.norm(), u1 = d_w / d_w.norm(), nxd = n_w.cross(d_w), u2 = nxd / nxd.norm();
            Matrix3d U;
            for (int r = 0; r < 3; ++r) { U(r, 0) = u0(r); U(r, 1) = u1(r); U(r, 2) = u2(r); }
            const Vector3d psi = eulerAnglesXYZ(U);
            it.orthonormal_vec = Vector4d(psi(0), psi(1), psi(2), std::atan2(d_w.norm(), n_w.norm()));
        }
    }
    // ---------------------------------------------------------------- track bookkeeping around the solve (closed-loop replay)
    // Written from the behaviour of feature_manager.cpp:73-158, 255-276, 607-760 -- what each call leaves in the track lists -- with ONE
    // template per operation for points and lines (the reference spells every list out twice).
    int last_track_num = 0;
    typedef std::map<int, std::vector<std::pair<int, Eigen::Matrix<double, 7, 1>>>> ImagePoints;
    typedef std::map<int, std::vector<Eigen::Matrix<double, 15, 1>>> ImageLines;
    static std::vector<FeaturePerFrame>& views(FeaturePerId& t) { return t.feature_per_frame; }
    static std::vector<LineFeaturePerFrame>& views(LineFeaturePerId& t) { return t.line_feature_per_frame; }

    // one more view for the track `id` (a new id opens a track that starts in `frame`); true when the track existed already
    template <class Tracks, class View> static bool appendView(Tracks& tracks, int id, int frame, const View& view) {
        for (auto& t : tracks)
            if (t.feature_id == id) { views(t).push_back(view); return true; }
        tracks.emplace_back(id, frame);
        views(tracks.back()).push_back(view);
        return false;
    }
    // Takes the measurements of the image that just arrived and decides whether the SECOND-NEWEST frame is a keyframe (then the oldest
    // frame will be marginalized, otherwise the second-newest is dropped): yes while the window is filling, when fewer than 20 tracks
    // continued, or when the tracks spanning frames frame_count-2 / frame_count-1 moved by MIN_PARALLAX on average (:73-158).
    bool addFeatureCheckParallax(int frame_count, const ImagePoints& image, const ImageLines& image_line, double td) {
        int continued = 0;
        for (const auto& msg : image) continued += appendView(feature, msg.first, frame_count, FeaturePerFrame(msg.second.front().second, td));
        for (const auto& msg : image_line) appendView(line_feature, msg.first, frame_count, LineFeaturePerFrame(msg.second.front(), td));
        last_track_num = continued;
        if (frame_count < 2 || continued < 20) return true;
        const int older = frame_count - 2, newer = frame_count - 1;
        double total = 0.0; int spanning = 0;
        for (const FeaturePerId& t : feature)
            if (t.start_frame <= older && t.endFrame() >= newer) { total += compensatedParallax2(t, frame_count); ++spanning; }
        return spanning == 0 || total / spanning >= MIN_PARALLAX;
    }
    // displacement on the normalised image plane between the second- and third-newest view of a track (:727-760; upstream has the
    // rotation compensation commented out, so the "compensated" candidate equals the plain one)
    static double compensatedParallax2(const FeaturePerId& track, int frame_count) {
        const Eigen::Vector3d& a = track.feature_per_frame[frame_count - 2 - track.start_frame].point;
        const Eigen::Vector3d& b = track.feature_per_frame[frame_count - 1 - track.start_frame].point;
        const double du = a(0) / a(2) - b(0), dv = a(1) / a(2) - b(1);
        return std::sqrt(du * du + dv * dv);
    }
    // tracks the last solve flagged (negative depth / endpoint behind the camera) are dropped (:255-276)
    void removeFailures() { feature.remove_if([](const FeaturePerId& t) { return t.solve_flag == 2; }); }
    void removeLineFailures() { line_feature.remove_if([](const LineFeaturePerId& t) { return t.solve_flag == 2; }); }

    // The OLDEST frame left the window: later tracks move one slot towards the front; a track anchored in the departed frame loses that
    // view, dies when fewer than `min_views` remain, and otherwise is handed to `reanchor` together with the view it lost.
    template <class Tracks, class Reanchor> static void oldestFrameLeft(Tracks& tracks, std::size_t min_views, Reanchor reanchor) {
        for (auto t = tracks.begin(); t != tracks.end();) {
            bool alive = true;
            if (t->start_frame > 0) --t->start_frame;
            else {
                auto& v = views(*t);
                const auto lost = v.front();
                v.erase(v.begin());
                alive = v.size() >= min_views;
                if (alive) reanchor(*t, lost);
            }
            t = alive ? std::next(t) : tracks.erase(t);
        }
    }
    // The SECOND-NEWEST frame (window slot frame_count - 1) was dropped: a track born in the newest frame moves one slot down, a track
    // that reaches the dropped frame loses that view (and dies with its last one), older tracks are untouched.
    template <class Tracks> static void secondNewestFrameLeft(Tracks& tracks, int frame_count) {
        const int dropped = frame_count - 1;
        for (auto t = tracks.begin(); t != tracks.end();) {
            bool alive = true;
            if (t->start_frame == frame_count) --t->start_frame;
            else if (t->endFrame() >= dropped) {
                auto& v = views(*t);
                v.erase(v.begin() + (dropped - t->start_frame));
                alive = !v.empty();
            }
            t = alive ? std::next(t) : tracks.erase(t);
        }
    }
    // :607-645 -- after initialization a point's depth lives in its anchor camera: moving the anchor re-expresses it in the next view's
    // camera (camera poses of the departed / new oldest frame given), falling back to INIT_DEPTH when it lands behind that camera
    void removeBackShiftDepth(const Eigen::Matrix3d& marg_R, const Eigen::Vector3d& marg_P, const Eigen::Matrix3d& new_R, const Eigen::Vector3d& new_P) {
        oldestFrameLeft(feature, 2, [&](FeaturePerId& t, const FeaturePerFrame& lost) {
            const Eigen::Vector3d in_world = marg_R * (lost.point * t.estimated_depth) + marg_P;
            const double depth = (new_R.transpose() * (in_world - new_P))(2);
            t.estimated_depth = depth > 0 ? depth : INIT_DEPTH;
        });
    }
    void removeBack() { oldestFrameLeft(feature, 1, [](FeaturePerId&, const FeaturePerFrame&) {}); }                          // :647-663
    void removeLineBack() { oldestFrameLeft(line_feature, 1, [](LineFeaturePerId&, const LineFeaturePerFrame&) {}); }         // :687-703 (world-frame parameters: nothing to move)
    void removeFront(int frame_count) { secondNewestFrameLeft(feature, frame_count); }                                        // :665-685
    void removeLineFront(int frame_count) { secondNewestFrameLeft(line_feature, frame_count); }                               // :705-725
    void clearState() { feature.clear(); line_feature.clear(); }                                                              // :28-32

    static bool usedPoint(FeaturePerId& it) { it.used_num = (int)it.feature_per_frame.size(); return it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2; }   // estimator.cpp:826
    static bool usedLine(LineFeaturePerId& it) { it.used_num = (int)it.line_feature_per_frame.size(); return it.used_num >= LINE_WINDOW; }                    // estimator.cpp:873
    int getFeatureCount() { int c = 0; for (auto& it : feature) c += usedPoint(it); return c; }
    int getLineFeatureCount() { int c = 0; for (auto& it : line_feature) c += usedLine(it); return c; }
    Eigen::VectorXd getDepthVector() { Eigen::VectorXd d; for (auto& it : feature) if (usedPoint(it)) d.push_back(1.0 / it.estimated_depth); return d; }     // feature_manager.cpp:290-306
    void setDepth(const Eigen::VectorXd& x) {                                                                                                                 // :235-253
        std::size_t next = 0;
        for (auto& t : feature)
            if (usedPoint(t)) { const double depth = 1.0 / x[next++]; t.estimated_depth = depth; t.solve_flag = depth < 0 ? 2 : 1; }
    }
    std::vector<Eigen::Vector4d> getLineOrthonormal() { std::vector<Eigen::Vector4d> v; for (auto& it : line_feature) if (usedLine(it)) v.push_back(it.orthonormal_vec); return v; }   // :308-331
    // setLineOrtho (:333-423): the endpoint-depth validity test uses the PRE-update orthonormal_vec with the post-update poses (Appendix D11)
    void setLineOrtho(std::vector<Eigen::Vector4d>& ortho, Eigen::Vector3d Ps[], Eigen::Matrix3d Rs[], Eigen::Vector3d tic, Eigen::Matrix3d ric) {
        using namespace Eigen;
        int idx = -1;
        for (auto& it : line_feature) {
            if (!usedLine(it)) continue;
            ++idx;
            const double a = it.orthonormal_vec(0), b = it.orthonormal_vec(1), c = it.orthonormal_vec(2), phi = it.orthonormal_vec(3);
            const double sa = sin(a), ca = cos(a), sb = sin(b), cb = cos(b), sc = sin(c), cc = cos(c);
            Vector3d U0(cb * cc, sa * sb * cc + ca * sc, -ca * sb * cc + sa * sc), U1(-cb * sc, -sa * sb * sc + ca * cc, ca * sb * sc + sa * cc);
            Vector3d n_w = U0 * cos(phi), d_w = U1 * sin(phi);
            const int i = it.start_frame;
            Matrix3d R_wc = Rs[i] * ric; Vector3d t_wc = Rs[i] * tic + Ps[i];
            Matrix3d RT = R_wc.transpose(); Vector3d t_cw = -(RT * t_wc);
            Vector3d d_c = RT * d_w, n_c = RT * n_w + t_cw.cross(d_c);
            Vector3d sp = it.line_feature_per_frame[0].start_point, ep = it.line_feature_per_frame[0].end_point;
            const double slope = -(ep(0) - sp(0)) / (ep(1) - sp(1));
            Vector3d sp2(sp(0) + 1.0, slope + sp(1), 1), ep2(ep(0) + 1.0, slope + ep(1), 1);
            Vector3d pi_s = sp.cross(sp2), pi_e = ep.cross(ep2);
            // D = L_c * pi, L_c = [ [n_c]x d_c ; -d_c^T 0 ]
            Vector3d Ds = n_c.cross(pi_s), De = n_c.cross(pi_e);
            const double ws = -d_c.dot(pi_s), we = -d_c.dot(pi_e);
            if (Ds(2) / ws < 0 || De(2) / we < 0) { it.solve_flag = 2; continue; }
            it.solve_flag = 1;
            it.orthonormal_vec = ortho.at(idx);
        }
    }
};
