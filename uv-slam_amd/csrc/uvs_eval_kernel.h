// uvs_eval_kernel.h -- one evaluation of every residual block of a window (uvs_evaluate).
//
// Produces, per block, what CostFunction::Evaluate + the Ceres corrector hand to the linear solver
// (LOCAL column sizes).  Used by the marginalization (which must evaluate the to-be-dropped factors
// at the post-solve state, marginalization_factor.cpp:3-69) and by the element-wise parity tests.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <string>
#include <vector>
#include "uvs_solve_kernel.h"

namespace uvsdev {

struct EvalOut {   // device pointers
    double *pt_r, *pt_J, *ln_r, *ln_J, *vp_r, *vp_J, *imu_r, *imu_J, *prior_r, *cost, *pt_Jtd;
};

// mode: bit 0 = apply the loss correction (robust); bit 1 = MARGIN_OLD subset: only the blocks that touch frame 0 are evaluated (prior,
// the IMU block of frame 0, points anchored at frame 0, observations j != 0 of lines that start at frame 0 -- estimator.cpp:1008-1129); the
// outputs of the other blocks are left untouched and `cost` is partial.
__global__ __launch_bounds__(NT) void k_evaluate(char* blob, double* ws, KOpts o, int mode, EvalOut out) {
    const int robust = mode & 1; const bool marg0 = (mode & 2) != 0;
    extern __shared__ __attribute__((aligned(16))) double sh[];   // same LDS map as the solver (launched with LDS_BYTES)
    const int tid = threadIdx.x;
    Ctx c;
    c.hdr = (const DevWin*)blob; c.bd = (const double*)blob; c.bi = (const int*)blob; c.ws = ws; c.sh = sh; c.o = o;
    const DevWin& h = *c.hdr;
    if (tid < UVS_XDIM) sh[L_X + tid] = c.bd[h.d_frames + tid];
    setup_window(c, (double*)blob, false, marg0 ? 0 : -1);
    __syncthreads();
    const double* x = sh + L_X;
    stage_rotations(c, x);
    prior_dx(c, x);
    __syncthreads();
    const double* RF = sh + L_RF; const double* ric = sh + L_EX; const double* tic = sh + L_EX + 9;
    const double* invd = c.bd + h.d_invd; const double* line = c.bd + h.d_line;
    double cost = prior_residual_rows(c);
    if (tid < h.prior_n) out.prior_r[tid] = sh[L_PR + tid];
    for (int ob = tid; ob < h.n_pt_obs; ob += NT) {
        const int lm = c.bi[h.i_pt_lm + ob], fi = c.bi[h.i_pt_fi + ob], fj = c.bi[h.i_pt_fj + ob];
        // relocalization blocks are solve-only (the reference's marginalization does not add them, estimator.cpp:1002-1228); outputs keep the
        // caller's observation numbering
        const int eo = h.relo_on ? c.bi[h.i_pt_eidx + ob] : ob;
        if (eo < 0 || (marg0 && fi != 0)) continue;
        double pi[3], pj[3], vij[4] = {0.0, 0.0, 0.0, 0.0}, jtd[2] = {0.0, 0.0};
        load_point_obs(c, ob, x[183], pi, pj, vij);
        double r[2], A[12], B[12], cl[2], E[12];
        point_eval<true, true>(x + 7 * fi, RF + 9 * fi, x + 7 * fj, RF + 9 * fj, ric, tic, invd[lm], pi, pj, o.sqrt_info, r, A, B, cl, E, vij, vij + 2, h.td_on ? jtd : nullptr);
        double sc = 1.0;
        if (robust) cost += 0.5 * cauchy(o.loss_pt, r[0] * r[0] + r[1] * r[1], &sc); else cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
        out.pt_r[2 * eo] = sc * r[0]; out.pt_r[2 * eo + 1] = sc * r[1];
        double* J = out.pt_J + 38 * (size_t)eo;
        for (int row = 0; row < 2; ++row) {
            for (int q = 0; q < 6; ++q) { J[row * 19 + q] = sc * A[6 * row + q]; J[row * 19 + 6 + q] = sc * B[6 * row + q]; J[row * 19 + 12 + q] = sc * E[6 * row + q]; }
            J[row * 19 + 18] = sc * cl[row];
        }
        if (out.pt_Jtd) { out.pt_Jtd[2 * eo] = sc * jtd[0]; out.pt_Jtd[2 * eo + 1] = sc * jtd[1]; }
    }
    for (int ob = tid; ob < h.n_ln_obs; ob += NT) {
        const int lm = c.bi[h.i_ln_lm + ob], fj = c.bi[h.i_ln_fj + ob], hv = c.bi[h.i_ln_vp + ob];
        if (marg0 && (fj == 0 || c.bi[h.i_ln_fj + c.bi[h.i_ln_beg + lm]] != 0)) continue;      // the line's first observation names its start frame
        const double* m = c.bd + h.d_lnmeas + ob; const int st = h.ln_stride;
        const double sp[3] = {m[0], m[st], m[2 * st]}, ep[3] = {m[3 * st], m[4 * st], m[5 * st]}, vp[3] = {m[6 * st], m[7 * st], m[8 * st]};
        LineGeom g;
        line_geom<true>(x + 7 * fj, x + 7 * fj + 3, RF + 9 * fj, ric, tic, line + 4 * lm, g);
        double r[2], Jp[12], Jl[8], sc = 1.0;
        line_residual<true>(g, sp, ep, o.line_factor, r, Jp, Jl);
        if (robust) cost += 0.5 * cauchy(o.loss_ln, r[0] * r[0] + r[1] * r[1], &sc); else cost += 0.5 * (r[0] * r[0] + r[1] * r[1]);
        out.ln_r[2 * ob] = sc * r[0]; out.ln_r[2 * ob + 1] = sc * r[1];
        double* J = out.ln_J + 20 * (size_t)ob;
        for (int row = 0; row < 2; ++row) { for (int q = 0; q < 6; ++q) J[row * 10 + q] = sc * Jp[6 * row + q]; for (int q = 0; q < 4; ++q) J[row * 10 + 6 + q] = sc * Jl[4 * row + q]; }
        double* Jv = out.vp_J + 10 * (size_t)ob;
        if (hv) {
            double rv, Jvp[6], Jvl[4]; sc = 1.0;
            vp_residual<true>(g, vp, o.vp_factor, &rv, Jvp, Jvl);
            if (robust) cost += 0.5 * cauchy(o.loss_vp, rv * rv, &sc); else cost += 0.5 * rv * rv;
            out.vp_r[ob] = sc * rv;
            for (int q = 0; q < 6; ++q) Jv[q] = sc * Jvp[q];
            for (int q = 0; q < 4; ++q) Jv[6 + q] = sc * Jvl[q];
        } else { out.vp_r[ob] = 0.0; for (int q = 0; q < 10; ++q) Jv[q] = 0.0; }
    }
    if (tid < h.n_imu) {
        const int fi = c.bi[h.i_imu + 2 * tid], skip = c.bi[h.i_imu + 2 * tid + 1] | (marg0 && fi != 0);
        double* wj = c.ws + h.w_imu + (size_t)tid * UVS_WIMU_STRIDE;
        if (!skip) {
            const double* blk = c.bd + h.d_imu + (size_t)tid * UVS_IMU_STRIDE;
            double r[15];
            imu_raw(blk, blk + UVS_IMU_JAC, o.G, x + 7 * fi, x + 77 + 9 * fi, x + 7 * (fi + 1), x + 77 + 9 * (fi + 1), r, wj);
            for (int i = 0; i < 15; ++i) wj[900 + i] = r[i];
        }
    }
    __syncthreads();
    for (int t = tid; t < h.n_imu * 465; t += NT) {
        const int b = t / 465, e = t - b * 465;
        const bool skip = c.bi[h.i_imu + 2 * b + 1] != 0 || (marg0 && c.bi[h.i_imu + 2 * b] != 0);
        const double* W = c.ws + h.w_imu_w + (size_t)b * UVS_IMU_WS;
        const double* wj = c.ws + h.w_imu + (size_t)b * UVS_WIMU_STRIDE;
        if (e < 450) { const int r = e / 30, cc = e - r * 30; double s = 0.0; if (!skip) for (int k = r; k < 15; ++k) s += W[r * 15 + k] * wj[k * 30 + cc]; out.imu_J[450 * (size_t)b + e] = s; }
        else { const int r = e - 450; double s = 0.0; if (!skip) for (int k = r; k < 15; ++k) s += W[r * 15 + k] * wj[900 + k]; out.imu_r[15 * b + r] = s; cost += 0.5 * s * s; }
    }
    double s4[4] = {cost, 0, 0, 0}, mx = 0.0;
    block_reduce(sh, s4, &mx);
    if (tid == 0) out.cost[0] = s4[0];
}

// device + pinned-host staging of one evaluation, kept by the solver handle (no allocation on the per-call path once it has grown)
struct EvalScratch {
    double* d = nullptr; size_t cap = 0;       // device, doubles
    double* h = nullptr; size_t hcap = 0;      // pinned host, doubles
    std::vector<double> work[11];              // host work arrays of the marginalization, kept between calls (a fresh 160 KB vector per call is an mmap / page-fault / munmap round trip)
    void release() { if (d) (void)hipFree(d); if (h) (void)hipHostFree(h); d = h = nullptr; cap = hcap = 0; }
};

// host driver: blob of window 0 must already be on the device (uvs_batch_upload)
// `view` (marginalization): *out receives POINTERS into the pinned staging buffer instead of copies into caller arrays (valid until the next
// call on this handle), and the device buffer is not cleared first -- the subset mode writes, and the caller reads, only the selected blocks.
static int run_evaluate(int device, hipStream_t stream, char* d_blob, double* d_ws, const DevWin& h, const KOpts& ko, int robust, uvs_eval* out, std::string& err, EvalScratch& sc,
                        bool view = false) {
    auto chk = [&](hipError_t e, const char* what) { if (e != hipSuccess) { err = std::string(what) + ": " + hipGetErrorString(e); return false; } return true; };
    if (!chk(hipSetDevice(device), "hipSetDevice")) return UVS_ERR_HIP;
    const size_t npo = (size_t)std::max(h.n_pt_obs, 1), nlo = (size_t)std::max(h.n_ln_obs, 1), ni = (size_t)std::max(h.n_imu, 1);
    const size_t sizes[11] = {2 * npo, 38 * npo, 2 * nlo, 20 * nlo, nlo, 10 * nlo, 15 * ni, 450 * ni, (size_t)UVS_MAX_PRIOR_DIM, 8, 2 * npo};
    size_t tot = 0; for (size_t v : sizes) tot += v;
    if (sc.cap < tot) {
        if (sc.d) (void)hipFree(sc.d);
        sc.d = nullptr; sc.cap = 0;
        if (!chk(hipMalloc((void**)&sc.d, tot * 8), "hipMalloc(eval)")) return UVS_ERR_HIP;
        sc.cap = tot;
    }
    if (sc.hcap < tot) {
        if (sc.h) (void)hipHostFree(sc.h);
        sc.h = nullptr; sc.hcap = 0;
        if (!chk(hipHostMalloc((void**)&sc.h, tot * 8, hipHostMallocDefault), "hipHostMalloc(eval)")) return UVS_ERR_HIP;
        sc.hcap = tot;
    }
    double* d = sc.d;
    if (!view && !chk(hipMemsetAsync(d, 0, tot * 8, stream), "memset(eval)")) return UVS_ERR_HIP;
    EvalOut eo; double* p = d;
    eo.pt_r = p; p += sizes[0]; eo.pt_J = p; p += sizes[1]; eo.ln_r = p; p += sizes[2]; eo.ln_J = p; p += sizes[3]; eo.vp_r = p; p += sizes[4];
    eo.vp_J = p; p += sizes[5]; eo.imu_r = p; p += sizes[6]; eo.imu_J = p; p += sizes[7]; eo.prior_r = p; p += sizes[8]; eo.cost = p; p += sizes[9]; eo.pt_Jtd = p;
    hipLaunchKernelGGL(k_evaluate, dim3(1), dim3(NT), LDS_BYTES, stream, d_blob, d_ws, ko, robust, eo);
    // ONE device-to-host copy into pinned memory, then plain host copies into the caller's arrays
    const bool ok = chk(hipGetLastError(), "k_evaluate launch") && chk(hipMemcpyAsync(sc.h, d, tot * 8, hipMemcpyDeviceToHost, stream), "memcpy D2H(eval)") &&
                    chk(hipStreamSynchronize(stream), "k_evaluate");
    if (!ok) return UVS_ERR_HIP;
    if (view) {
        auto at = [&](const double* dsrc) { return sc.h + (dsrc - d); };
        out->pt_r = at(eo.pt_r); out->pt_J = at(eo.pt_J); out->ln_r = at(eo.ln_r); out->ln_J = at(eo.ln_J); out->vp_r = at(eo.vp_r); out->vp_J = at(eo.vp_J);
        out->imu_r = at(eo.imu_r); out->imu_J = at(eo.imu_J); out->prior_r = at(eo.prior_r); out->cost = *at(eo.cost); out->pt_Jtd = h.td_on ? at(eo.pt_Jtd) : nullptr;
        return UVS_OK;
    }
    auto back = [&](double* dst, const double* dsrc, size_t n) { if (dst && n) std::memcpy(dst, sc.h + (dsrc - d), n * 8); };
    back(out->pt_r, eo.pt_r, 2 * (size_t)(h.n_pt_obs - h.n_relo)); back(out->pt_J, eo.pt_J, 38 * (size_t)(h.n_pt_obs - h.n_relo));
    back(out->ln_r, eo.ln_r, 2 * (size_t)h.n_ln_obs); back(out->ln_J, eo.ln_J, 20 * (size_t)h.n_ln_obs);
    back(out->vp_r, eo.vp_r, (size_t)h.n_ln_obs); back(out->vp_J, eo.vp_J, 10 * (size_t)h.n_ln_obs);
    back(out->imu_r, eo.imu_r, 15 * (size_t)h.n_imu); back(out->imu_J, eo.imu_J, 450 * (size_t)h.n_imu);
    back(out->prior_r, eo.prior_r, (size_t)h.prior_n);
    back(&out->cost, eo.cost, 1);
    if (h.td_on) back(out->pt_Jtd, eo.pt_Jtd, 2 * (size_t)(h.n_pt_obs - h.n_relo));
    return UVS_OK;
}

}  // namespace uvsdev
