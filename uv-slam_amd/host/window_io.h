// window_io.h -- binary record / replay format of one sliding window.
// The reference never serialises the estimator window; this file is what a dump hook placed after vector2double()
// (estimator.cpp:800) writes and what the GPU box replays without ROS.  Little-endian, 8-byte aligned:
//   char magic[8] = "UVSWIN01"; int32 n_points, n_point_obs, n_lines, n_line_obs, n_imu, prior_n, prior_nblocks, flags (1 = has_td, 2 = has_relo);
//   double pose[77] sb[99] ex[7] td; double inv_depth[np]; int32 pt_lm/fi/fj[npo]; double pt_pi[3npo] pt_pj[3npo];
//   (has_td: double pt_vel_i[2npo] pt_vel_j[2npo] pt_td_i[npo] pt_td_j[npo] -- the ProjectionTdFactor inputs);
//   double line_orth[4nl]; int32 ln_lm/fj/has_vp[nlo]; double ln_sp/ep/vp[3nlo]; imu: n_imu x (467 doubles + int32 frame_i, skip);
//   prior (if prior_n): int32 kind/frame/size/idx/x0off[16 each]; double x0[144] r0[n] J0[n*n];
//   (has_relo: int32 n_relo, relo_frame_local_index; double relo_pose[7]; int32 relo_lm[n_relo] (+pad to 8 bytes); double relo_pi[3 n_relo] relo_pj[3 n_relo]
//    -- the relocalization blocks, estimator.cpp:944-978).
// The same layout is written / read by uv-slam_amd/abi.py (Window.save / Window.load).
#pragma once
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/uvs_solver.h"

struct WindowFile {      // owns the arrays a uvs_window points to
    uvs_window w;
    std::vector<double> inv_depth, pt_pi, pt_pj, line_orth, ln_sp, ln_ep, ln_vp, pt_vel_i, pt_vel_j, pt_td_i, pt_td_j, relo_pi, relo_pj;
    std::vector<int32_t> relo_lm;
    int relo_frame_local_index = 0;      // not part of uvs_window (the solver does not need it); Estimator::double2vector does (estimator.cpp:683-685)
    bool has_td = false;
    std::vector<int32_t> pt_lm, pt_fi, pt_fj, ln_lm, ln_fj, ln_has_vp;
    std::vector<uvs_imu_block> imu;
    uvs_prior prior;
    bool load(const std::string& path) {
        FILE* f = std::fopen(path.c_str(), "rb"); if (!f) return false;
        char magic[8]; int32_t hd[8];
        bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "UVSWIN01", 8) == 0 && std::fread(hd, 4, 8, f) == 8;
        auto rd = [&](void* p, size_t sz, size_t n) { if (ok && n) ok = std::fread(p, sz, n, f) == n; };
        std::memset(&w, 0, sizeof(w));
        // header counts are untrusted (a truncated or foreign file must fail the load, not overrun the fixed-size prior arrays)
        const int kMaxCount = 1 << 24;
        for (int q = 0; q < 7 && ok; ++q) ok = hd[q] >= 0 && hd[q] <= kMaxCount;
        ok = ok && hd[4] <= UVS_WINDOW_SIZE && hd[5] <= UVS_MAX_PRIOR_DIM && hd[6] <= UVS_MAX_PRIOR_BLOCKS;
        if (ok) {
            const int np = hd[0], npo = hd[1], nl = hd[2], nlo = hd[3], ni = hd[4], pn = hd[5], pnb = hd[6];
            rd(w.pose, 8, 77); rd(w.speedbias, 8, 99); rd(w.ex_pose, 8, 7); rd(&w.td, 8, 1);
            inv_depth.resize(np); rd(inv_depth.data(), 8, np);
            pt_lm.resize(npo); pt_fi.resize(npo); pt_fj.resize(npo); rd(pt_lm.data(), 4, npo); rd(pt_fi.data(), 4, npo); rd(pt_fj.data(), 4, npo);
            if (npo % 2) { int32_t pad; rd(&pad, 4, 1); }
            pt_pi.resize(3 * npo); pt_pj.resize(3 * npo); rd(pt_pi.data(), 8, 3 * npo); rd(pt_pj.data(), 8, 3 * npo);
            has_td = (hd[7] & 1) != 0;
            if (has_td) { pt_vel_i.resize(2 * npo); pt_vel_j.resize(2 * npo); pt_td_i.resize(npo); pt_td_j.resize(npo);
                          rd(pt_vel_i.data(), 8, 2 * npo); rd(pt_vel_j.data(), 8, 2 * npo); rd(pt_td_i.data(), 8, npo); rd(pt_td_j.data(), 8, npo); }
            line_orth.resize(4 * nl); rd(line_orth.data(), 8, 4 * nl);
            ln_lm.resize(nlo); ln_fj.resize(nlo); ln_has_vp.resize(nlo); rd(ln_lm.data(), 4, nlo); rd(ln_fj.data(), 4, nlo); rd(ln_has_vp.data(), 4, nlo);
            if (nlo % 2) { int32_t pad; rd(&pad, 4, 1); }
            ln_sp.resize(3 * nlo); ln_ep.resize(3 * nlo); ln_vp.resize(3 * nlo); rd(ln_sp.data(), 8, 3 * nlo); rd(ln_ep.data(), 8, 3 * nlo); rd(ln_vp.data(), 8, 3 * nlo);
            imu.resize(ni);
            for (int b = 0; b < ni; ++b) {
                double h[17]; rd(h, 8, 17); uvs_imu_block& ib = imu[b]; std::memset(&ib, 0, sizeof(ib));
                ib.sum_dt = h[0]; std::memcpy(ib.delta_p, h + 1, 24); std::memcpy(ib.delta_q, h + 4, 32); std::memcpy(ib.delta_v, h + 8, 24); std::memcpy(ib.linearized_ba, h + 11, 24); std::memcpy(ib.linearized_bg, h + 14, 24);
                rd(ib.jacobian, 8, 225); rd(ib.covariance, 8, 225); int32_t fs[2]; rd(fs, 4, 2); ib.frame_i = fs[0]; ib.skip = fs[1];
            }
            std::memset(&prior, 0, sizeof(prior));
            if (pn > 0) {
                prior.n = pn; prior.n_blocks = pnb;
                rd(prior.block_kind, 4, 16); rd(prior.block_frame, 4, 16); rd(prior.block_size, 4, 16); rd(prior.block_idx, 4, 16); rd(prior.x0_off, 4, 16);
                rd(prior.x0, 8, 144); rd(prior.linearized_residuals, 8, pn); rd(prior.linearized_jacobians, 8, (size_t)pn * pn);
            }
            w.n_points = np; w.n_point_obs = npo; w.n_lines = nl; w.n_line_obs = nlo; w.n_imu = ni;
            w.inv_depth = inv_depth.data(); w.pt_lm = pt_lm.data(); w.pt_fi = pt_fi.data(); w.pt_fj = pt_fj.data(); w.pt_pi = pt_pi.data(); w.pt_pj = pt_pj.data();
            w.line_orth = line_orth.data(); w.ln_lm = ln_lm.data(); w.ln_fj = ln_fj.data(); w.ln_has_vp = ln_has_vp.data(); w.ln_sp = ln_sp.data(); w.ln_ep = ln_ep.data(); w.ln_vp = ln_vp.data();
            w.imu = imu.data(); w.prior = pn > 0 ? &prior : nullptr;
            if (has_td) { w.pt_vel_i = pt_vel_i.data(); w.pt_vel_j = pt_vel_j.data(); w.pt_td_i = pt_td_i.data(); w.pt_td_j = pt_td_j.data(); }
            if (hd[7] & 2) {
                int32_t nr[2] = {0, 0}; rd(nr, 4, 2); ok = ok && nr[0] >= 0 && nr[0] <= kMaxCount; const int n = ok ? nr[0] : 0; relo_frame_local_index = nr[1];
                rd(w.relo_pose, 8, 7); relo_lm.resize(n); rd(relo_lm.data(), 4, n); if (n % 2) { int32_t pad; rd(&pad, 4, 1); }
                relo_pi.resize(3 * n); relo_pj.resize(3 * n); rd(relo_pi.data(), 8, 3 * n); rd(relo_pj.data(), 8, 3 * n);
                w.n_relo_obs = n; w.relo_lm = relo_lm.data(); w.relo_pi = relo_pi.data(); w.relo_pj = relo_pj.data();
            }
        }
        std::fclose(f);
        return ok;
    }
    // the dump hook's writer: any uvs_window (e.g. the one uvs::Problem::fill() assembled after vector2double(), estimator.cpp:800)
    static bool save(const std::string& path, const uvs_window& w, int relo_frame = 0) {
        FILE* f = std::fopen(path.c_str(), "wb"); if (!f) return false;
        const bool td = w.pt_vel_i && w.pt_vel_j && w.pt_td_i && w.pt_td_j;
        const int np = w.n_points, npo = w.n_point_obs, nl = w.n_lines, nlo = w.n_line_obs, ni = w.n_imu, pn = w.prior ? w.prior->n : 0;
        const int nrl = w.n_relo_obs;
        const int32_t hd[8] = {np, npo, nl, nlo, ni, pn, pn ? w.prior->n_blocks : 0, (td ? 1 : 0) | (nrl > 0 ? 2 : 0)}, pad = 0;
        bool ok = true;
        auto wr = [&](const void* p, size_t sz, size_t n) { if (ok && n) ok = std::fwrite(p, sz, n, f) == n; };
        wr("UVSWIN01", 1, 8); wr(hd, 4, 8);
        wr(w.pose, 8, 77); wr(w.speedbias, 8, 99); wr(w.ex_pose, 8, 7); wr(&w.td, 8, 1);
        wr(w.inv_depth, 8, np); wr(w.pt_lm, 4, npo); wr(w.pt_fi, 4, npo); wr(w.pt_fj, 4, npo); if (npo % 2) wr(&pad, 4, 1);
        wr(w.pt_pi, 8, 3 * npo); wr(w.pt_pj, 8, 3 * npo);
        if (td) { wr(w.pt_vel_i, 8, 2 * npo); wr(w.pt_vel_j, 8, 2 * npo); wr(w.pt_td_i, 8, npo); wr(w.pt_td_j, 8, npo); }
        wr(w.line_orth, 8, 4 * nl); wr(w.ln_lm, 4, nlo); wr(w.ln_fj, 4, nlo); wr(w.ln_has_vp, 4, nlo); if (nlo % 2) wr(&pad, 4, 1);
        wr(w.ln_sp, 8, 3 * nlo); wr(w.ln_ep, 8, 3 * nlo); wr(w.ln_vp, 8, 3 * nlo);
        for (int b = 0; b < ni; ++b) {
            const uvs_imu_block& ib = w.imu[b];
            wr(&ib.sum_dt, 8, 1); wr(ib.delta_p, 8, 3); wr(ib.delta_q, 8, 4); wr(ib.delta_v, 8, 3); wr(ib.linearized_ba, 8, 3); wr(ib.linearized_bg, 8, 3);
            wr(ib.jacobian, 8, 225); wr(ib.covariance, 8, 225); const int32_t fs[2] = {ib.frame_i, ib.skip}; wr(fs, 4, 2);
        }
        if (pn > 0) {
            const uvs_prior& p = *w.prior;
            wr(p.block_kind, 4, 16); wr(p.block_frame, 4, 16); wr(p.block_size, 4, 16); wr(p.block_idx, 4, 16); wr(p.x0_off, 4, 16);
            wr(p.x0, 8, 144); wr(p.linearized_residuals, 8, pn); wr(p.linearized_jacobians, 8, (size_t)pn * pn);
        }
        if (nrl > 0) {
            const int32_t nr[2] = {nrl, relo_frame}; wr(nr, 4, 2); wr(w.relo_pose, 8, 7); wr(w.relo_lm, 4, nrl); if (nrl % 2) wr(&pad, 4, 1);
            wr(w.relo_pi, 8, 3 * nrl); wr(w.relo_pj, 8, 3 * nrl);
        }
        std::fclose(f);
        return ok;
    }
};
