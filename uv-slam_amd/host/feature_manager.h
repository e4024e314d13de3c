// feature_manager.h -- the part of FeatureManager (vins_estimator/src/feature_manager.{h,cpp}) that Estimator::optimization()
// and vector2double()/double2vector() touch: the per-id track lists and the get/set of solver parameters.
// Triangulation / parallax / slide-window bookkeeping (feature_manager.cpp:73-158,427-725) are NOT mirrored.
#pragma once
#include <list>
#include <vector>
#include "parameters.h"
#include "utility.h"

class FeaturePerFrame { public: FeaturePerFrame(const Eigen::Vector3d& p) : point(p), cur_td(0) {} Eigen::Vector3d point; double cur_td; };
class FeaturePerId {
  public:
    const int feature_id; int start_frame; std::vector<FeaturePerFrame> feature_per_frame; int used_num; double estimated_depth; int solve_flag;
    FeaturePerId(int id, int start) : feature_id(id), start_frame(start), used_num(0), estimated_depth(-1.0), solve_flag(0) {}
};
class LineFeaturePerFrame { public: Eigen::Vector3d start_point, end_point, vp; };
class LineFeaturePerId {
  public:
    const int feature_id; int start_frame; std::vector<LineFeaturePerFrame> line_feature_per_frame; int used_num; Eigen::Vector4d orthonormal_vec; int solve_flag;
    LineFeaturePerId(int id, int start) : feature_id(id), start_frame(start), used_num(0), solve_flag(0) {}
};

class FeatureManager {
  public:
    std::list<FeaturePerId> feature;
    std::list<LineFeaturePerId> line_feature;
    static bool usedPoint(FeaturePerId& it) { it.used_num = (int)it.feature_per_frame.size(); return it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2; }   // estimator.cpp:826
    static bool usedLine(LineFeaturePerId& it) { it.used_num = (int)it.line_feature_per_frame.size(); return it.used_num >= LINE_WINDOW; }                    // estimator.cpp:873
    int getFeatureCount() { int c = 0; for (auto& it : feature) c += usedPoint(it); return c; }
    int getLineFeatureCount() { int c = 0; for (auto& it : line_feature) c += usedLine(it); return c; }
    Eigen::VectorXd getDepthVector() { Eigen::VectorXd d; for (auto& it : feature) if (usedPoint(it)) d.push_back(1.0 / it.estimated_depth); return d; }     // feature_manager.cpp:290-306
    void setDepth(const Eigen::VectorXd& x) {                                                                                                                 // :235-253
        int k = -1;
        for (auto& it : feature) { if (!usedPoint(it)) continue; it.estimated_depth = 1.0 / x[++k]; it.solve_flag = it.estimated_depth < 0 ? 2 : 1; }
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
