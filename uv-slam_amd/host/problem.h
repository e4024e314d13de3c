// problem.h -- uvs::Problem: the surface of ceres::Problem that Estimator::optimization() uses (estimator.cpp:763-997),
// implemented as a RECORDER: parameter blocks are identified by address (as in Ceres), resolved to (kind, index) through the
// Estimator's para_* arrays, and residual blocks are appended to the flat uvs_window that the HIP solver consumes.
// uvs::Solve() == ceres::Solve(): one uvs_solve_window() / uvs_large_solve_fused() call (Options::path), results written back into the para_* arrays in place.
#pragma once
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <stdexcept>
#include <vector>
#include "factor/factors.h"

namespace uvs {

struct BlockRef { int kind; int index; };      // UVS_BLOCK_* for frames; 100 = point landmark, 101 = line landmark, 102 = relo_Pose
struct AddressMap {                             // filled by the Estimator with its para_* base addresses
    double (*pose)[SIZE_POSE]; double (*speedbias)[SIZE_SPEEDBIAS]; double (*ex_pose)[SIZE_POSE]; double (*feature)[SIZE_FEATURE];
    double (*ortho)[SIZE_LINE_FEATURE]; double (*td)[1];
    double* relo_pose = nullptr;                // relo_Pose[SIZE_POSE] (estimator.h:137)
    BlockRef resolve(const double* p) const {
        auto in = [&](const void* base, size_t elem, int count, int* idx) { const char* b = (const char*)base; const char* q = (const char*)p; if (q < b || q >= b + elem * count) return false; *idx = (int)((q - b) / elem); return (q - b) % elem == 0; };
        int i;
        if (in(pose, sizeof(pose[0]), WINDOW_SIZE + 1, &i)) return {UVS_BLOCK_POSE, i};
        if (in(speedbias, sizeof(speedbias[0]), WINDOW_SIZE + 1, &i)) return {UVS_BLOCK_SPEEDBIAS, i};
        if (in(ex_pose, sizeof(ex_pose[0]), NUM_OF_CAM, &i)) return {UVS_BLOCK_EX_POSE, 0};
        if (in(feature, sizeof(feature[0]), NUM_OF_F, &i)) return {100, i};
        if (in(ortho, sizeof(ortho[0]), NUM_OF_LF, &i)) return {101, i};
        if (in(td, sizeof(td[0]), 1, &i)) return {UVS_BLOCK_TD, 0};
        if (relo_pose && p == relo_pose) return {102, 0};
        throw std::invalid_argument("uvs::Problem: parameter block is not one of the Estimator's para_* arrays");
    }
};

// How ONE window is solved (this repository's extension of ceres::Solver::Options; both forms run the same LM controller):
//   PERSISTENT_KERNEL  uvs_solve_window(): the whole solve in one workgroup on one compute unit (what a batch of windows uses per window)
//   MULTI_WORKGROUP    uvs_large_solve_fused(): landmark chunks on many compute units, one workgroup for the reduced solve, control on the
//                      device -- 17-23 % lower latency for a single window on an otherwise idle GPU; takes no relocalization blocks
//   AUTO               MULTI_WORKGROUP unless the window carries relocalization blocks; UVS_HOST_SOLVER_PATH=persistent|multi overrides
enum SolverPath { AUTO = 0, PERSISTENT_KERNEL, MULTI_WORKGROUP };
struct Options { int max_num_iterations = 10; double max_solver_time_in_seconds = 1e9; /* the wall-clock cap is NOT honoured (Appendix D4): parity needs a deterministic iteration count */
                 SolverPath path = AUTO; };
struct Summary { uvs_report report; int status = 0; int iterations() const { return report.num_iterations; } };

class Problem {
  public:
    explicit Problem(const AddressMap& m) : map(m) {}
    ~Problem() { for (auto* c : owned_costs) delete c; for (auto* l : owned_loss) delete l; for (auto* p : owned_param) delete p; }
    void AddParameterBlock(double* values, int size, ceres_like::LocalParameterization* lp = nullptr) { (void)values; (void)size; if (lp) owned_param.push_back(lp); }
    void SetParameterBlockConstant(double* values) { if (map.resolve(values).kind == UVS_BLOCK_EX_POSE) ex_constant = true; }
    // AddResidualBlock(cost, loss, blocks...) -- Problem takes ownership like ceres::Problem (default options)
    void AddResidualBlock(CostFunction* cost, ceres_like::LossFunction* loss, const std::vector<double*>& blocks) {
        owned_costs.push_back(cost); if (loss && std::find(owned_loss.begin(), owned_loss.end(), loss) == owned_loss.end()) owned_loss.push_back(loss);
        switch (cost->kind()) {
            case F_MARGINALIZATION: prior = &static_cast<MarginalizationFactor*>(cost)->marginalization_info->prior; break;
            case F_IMU: {
                IMUFactor* f = static_cast<IMUFactor*>(cost); const IntegrationBase* p = f->pre_integration;
                uvs_imu_block b; std::memset(&b, 0, sizeof(b));
                b.sum_dt = p->sum_dt;
                for (int k = 0; k < 3; ++k) { b.delta_p[k] = p->delta_p(k); b.delta_v[k] = p->delta_v(k); b.linearized_ba[k] = p->linearized_ba(k); b.linearized_bg[k] = p->linearized_bg(k); }
                b.delta_q[0] = p->delta_q.x(); b.delta_q[1] = p->delta_q.y(); b.delta_q[2] = p->delta_q.z(); b.delta_q[3] = p->delta_q.w();
                std::memcpy(b.jacobian, p->jacobian, sizeof(b.jacobian)); std::memcpy(b.covariance, p->covariance, sizeof(b.covariance));
                b.frame_i = map.resolve(blocks.at(0)).index; b.skip = 0;
                imu.push_back(b); break;
            }
            case F_PROJECTION: {
                ProjectionFactor* f = static_cast<ProjectionFactor*>(cost);
                if (map.resolve(blocks.at(1)).kind == 102) {      // relocalization block (estimator.cpp:966-970): second pose block = relo_Pose
                    relo_lm.push_back(map.resolve(blocks.at(3)).index);
                    for (int k = 0; k < 3; ++k) { relo_pi.push_back(f->pts_i(k)); relo_pj.push_back(f->pts_j(k)); }
                    break;
                }
                pt_fi.push_back(map.resolve(blocks.at(0)).index); pt_fj.push_back(map.resolve(blocks.at(1)).index); pt_lm.push_back(map.resolve(blocks.at(3)).index);
                for (int k = 0; k < 3; ++k) { pt_pi.push_back(f->pts_i(k)); pt_pj.push_back(f->pts_j(k)); }
                break;
            }
            case F_PROJECTION_TD: {      // same point arrays + the time-offset inputs; TR / ROW * row folded into the per-observation td (include/uvs_solver.h)
                ProjectionTdFactor* f = static_cast<ProjectionTdFactor*>(cost);
                pt_fi.push_back(map.resolve(blocks.at(0)).index); pt_fj.push_back(map.resolve(blocks.at(1)).index); pt_lm.push_back(map.resolve(blocks.at(3)).index);
                for (int k = 0; k < 3; ++k) { pt_pi.push_back(f->pts_i(k)); pt_pj.push_back(f->pts_j(k)); }
                for (int k = 0; k < 2; ++k) { pt_vel_i.push_back(f->velocity_i(k)); pt_vel_j.push_back(f->velocity_j(k)); }
                pt_td_i.push_back(f->td_i - TR / ROW * f->row_i); pt_td_j.push_back(f->td_j - TR / ROW * f->row_j);
                break;
            }
            case F_LINE: {
                LineProjectionFactor* f = static_cast<LineProjectionFactor*>(cost);
                ln_fj.push_back(map.resolve(blocks.at(0)).index); ln_lm.push_back(map.resolve(blocks.at(1)).index); ln_has_vp.push_back(0);
                for (int k = 0; k < 3; ++k) { ln_sp.push_back(f->sp(k)); ln_ep.push_back(f->ep(k)); ln_vp.push_back(0.0); }
                break;
            }
            case F_VP: {      // the reference adds it right after the line block of the same observation (estimator.cpp:920-925)
                VPProjectionFactor* f = static_cast<VPProjectionFactor*>(cost);
                if (ln_lm.empty() || ln_lm.back() != map.resolve(blocks.at(1)).index || ln_fj.back() != map.resolve(blocks.at(0)).index) throw std::invalid_argument("VP block must follow its line block");
                ln_has_vp.back() = 1; for (int k = 0; k < 3; ++k) ln_vp[ln_vp.size() - 3 + k] = f->vp(k);
                break;
            }
        }
    }
    void AddResidualBlock(CostFunction* c, ceres_like::LossFunction* l, double* a, double* b) { AddResidualBlock(c, l, std::vector<double*>{a, b}); }
    void AddResidualBlock(CostFunction* c, ceres_like::LossFunction* l, double* a, double* b, double* c2, double* d) { AddResidualBlock(c, l, std::vector<double*>{a, b, c2, d}); }
    void AddResidualBlock(CostFunction* c, ceres_like::LossFunction* l, double* a, double* b, double* c2, double* d, double* e) { AddResidualBlock(c, l, std::vector<double*>{a, b, c2, d, e}); }

    // assembled view (valid while the Problem lives)
    void fill(uvs_window* w, int n_points, int n_lines) const {
        std::memset(w, 0, sizeof(*w));
        std::memcpy(w->pose, map.pose, sizeof(w->pose)); std::memcpy(w->speedbias, map.speedbias, sizeof(w->speedbias)); std::memcpy(w->ex_pose, map.ex_pose[0], sizeof(w->ex_pose));
        w->n_points = n_points; w->n_point_obs = (int)pt_lm.size(); w->inv_depth = &map.feature[0][0];
        w->pt_lm = pt_lm.data(); w->pt_fi = pt_fi.data(); w->pt_fj = pt_fj.data(); w->pt_pi = pt_pi.data(); w->pt_pj = pt_pj.data();
        w->n_lines = n_lines; w->n_line_obs = (int)ln_lm.size(); w->line_orth = &map.ortho[0][0];
        w->ln_lm = ln_lm.data(); w->ln_fj = ln_fj.data(); w->ln_sp = ln_sp.data(); w->ln_ep = ln_ep.data(); w->ln_has_vp = ln_has_vp.data(); w->ln_vp = ln_vp.data();
        w->n_imu = (int)imu.size(); w->imu = imu.data(); w->prior = prior;
        w->td = map.td[0][0];
        if (!relo_lm.empty()) { w->n_relo_obs = (int)relo_lm.size(); w->relo_lm = relo_lm.data(); w->relo_pi = relo_pi.data(); w->relo_pj = relo_pj.data(); std::memcpy(w->relo_pose, map.relo_pose, sizeof(w->relo_pose)); }
        if (!pt_td_i.empty()) { w->pt_vel_i = pt_vel_i.data(); w->pt_vel_j = pt_vel_j.data(); w->pt_td_i = pt_td_i.data(); w->pt_td_j = pt_td_j.data(); }
    }
    AddressMap map;
    bool ex_constant = false;
    const uvs_prior* prior = nullptr;
    std::vector<uvs_imu_block> imu;
    std::vector<int32_t> pt_lm, pt_fi, pt_fj, ln_lm, ln_fj, ln_has_vp, relo_lm;
    std::vector<double> pt_pi, pt_pj, ln_sp, ln_ep, ln_vp, pt_vel_i, pt_vel_j, pt_td_i, pt_td_j, relo_pi, relo_pj;
  private:
    std::vector<CostFunction*> owned_costs; std::vector<ceres_like::LossFunction*> owned_loss; std::vector<ceres_like::LocalParameterization*> owned_param;
};

// == ceres::Solve(options, &problem, &summary) at estimator.cpp:992; n_points / n_lines = f_manager.getFeatureCount() / getLineFeatureCount()
inline void Solve(const Options& options, Problem* problem, Summary* summary, uvs_solver* solver, int n_points, int n_lines) {
    uvs_window w; problem->fill(&w, n_points, n_lines);
    std::vector<double> invd(n_points > 0 ? n_points : 1), lines(4 * (n_lines > 0 ? n_lines : 1));
    uvs_state st; st.inv_depth = invd.data(); st.line_orth = lines.data();
    SolverPath path = options.path;
    if (const char* env = std::getenv("UVS_HOST_SOLVER_PATH")) path = env[0] == 'p' ? PERSISTENT_KERNEL : env[0] == 'm' ? MULTI_WORKGROUP : path;
    if (path == AUTO) path = w.n_relo_obs > 0 ? PERSISTENT_KERNEL : MULTI_WORKGROUP;
    summary->status = path == MULTI_WORKGROUP ? uvs_large_solve_fused(solver, &w, &st, &summary->report, nullptr) : uvs_solve_window(solver, &w, &st, &summary->report);
    if (summary->status != UVS_OK && summary->status != UVS_ERR_NUMERIC) return;      // like the reference, the caller ignores the summary
    std::memcpy(problem->map.pose, st.pose, sizeof(st.pose)); std::memcpy(problem->map.speedbias, st.speedbias, sizeof(st.speedbias));
    problem->map.td[0][0] = st.td;
    if (w.n_relo_obs > 0) std::memcpy(problem->map.relo_pose, st.relo_pose, sizeof(st.relo_pose));
    std::memcpy(problem->map.ex_pose[0], st.ex_pose, sizeof(st.ex_pose));      // unchanged unless ESTIMATE_EXTRINSIC
    for (int k = 0; k < n_points; ++k) problem->map.feature[k][0] = invd[k];
    for (int k = 0; k < 4 * n_lines; ++k) (&problem->map.ortho[0][0])[k] = lines[k];
}

}  // namespace uvs
