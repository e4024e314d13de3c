// micro_chain.hip -- cycles per pivot of the 16-pivot chain of the reduced solve's diagonal-block factorization (uvs_solve_kernel.h: chol_factor_impl S2),
// one wavefront on one SIMD, for the variants that were considered:
//   0  round-2 form: readlane of the pivot out of the MFMA result -> rcp -> e -> e + e^2 -> multiplier -> rank-1 MFMA  (+ the trailing MFMA that builds W = L^-1)
//   1  pivot AHEAD of the matrix: next pivot = a[j+1][j+1] - a[j][j+1]^2 / piv_j from values read before update j; its reciprocal overlaps the MFMA
//   2  variant 1 without the W MFMA (timing only: how much of a link is the second MFMA's pipe occupancy)
//   3  variant 0 without the W MFMA
//   4/5  placements of the W MFMA (directly behind the factor's / lagging two pivots)
//   6  TWO pivots per link: the 2 x 2 pivot block's elimination as ONE rank-2 MFMA (two of the four k-slots), rows j and j+1 of the block exchanged
//      between their 16-lane rows with v_permlane16_swap;  7 = variant 6 without the W MFMA
// build: hipcc --offload-arch=gfx950 -O3 tools/micro_chain.hip -o tools/micro_chain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double bcast_lane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pivot_inverse(double piv) { const double y0 = __builtin_amdgcn_rcp(piv); const double e = fma(-piv, y0, 1.0); return fma(y0, fma(e, e, e), y0); }
template <int V> __global__ void k_chain(const double* A, double* out, long long* cyc, int reps) {
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    d4_t d0;
    for (int q = 0; q < 4; ++q) d0[q] = A[(lk + 4 * q) * 16 + li];
    double sink = 0.0;
    double pivs[4] = {1, 1, 1, 1};
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        d4_t dacc = d0, T;
        for (int q = 0; q < 4; ++q) { T[q] = (lk + 4 * q == li) ? 1.0 : 0.0; dacc[q] += 1e-9 * sink; }
        double us_prev = 0.0;
        if (V == 0 || V == 3) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int reg = j >> 2, slot = j & 3;
                const double piv = bcast_lane(dacc[reg], 16 * slot + j);
                if (V == 0 && j > 0) T = __builtin_amdgcn_mfma_f64_16x16x4f64(us_prev, T[(j - 1) >> 2], T, 0, 0, 0);
                const double m = (lk == slot && li > j) ? dacc[reg] : 0.0;
                if (r == 0 && lk == slot) pivs[reg] = piv;
                const double y0 = __builtin_amdgcn_rcp(piv);
                const double e = fma(-piv, y0, 1.0), us0 = -m * y0;
                const double us = fma(us0, fma(e, e, e), us0);
                dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(us, dacc[reg], dacc, 0, 0, 0);
                us_prev = us;
            }
        } else if (V == 4 || V == 5) {
            // 4: the W MFMA right BEHIND the factor's MFMA of the same pivot (two independent MFMAs back to back, then the scalar link)
            // 5: like 4 but the W elimination lags one more pivot
            double usq[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int reg = j >> 2, slot = j & 3;
                const double piv = bcast_lane(dacc[reg], 16 * slot + j);
                const double m = (lk == slot && li > j) ? dacc[reg] : 0.0;
                const double y0 = __builtin_amdgcn_rcp(piv);
                const double e = fma(-piv, y0, 1.0), us0 = -m * y0;
                const double us = fma(us0, fma(e, e, e), us0);
                usq[j] = us;
                dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(us, dacc[reg], dacc, 0, 0, 0);
                if (V == 4 && j < 15) T = __builtin_amdgcn_mfma_f64_16x16x4f64(us, T[j >> 2], T, 0, 0, 0);
                if (V == 5 && j >= 1 && j < 16) T = __builtin_amdgcn_mfma_f64_16x16x4f64(usq[j - 1], T[(j - 1) >> 2], T, 0, 0, 0);
            }
        } else if (V == 6 || V == 7) {
            double ap_prev = 0.0;
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const int reg = j >> 2, s0 = j & 3, s1 = s0 + 1;
                const double a00 = bcast_lane(dacc[reg], 16 * s0 + j), a10 = bcast_lane(dacc[reg], 16 * s0 + j + 1), a11 = bcast_lane(dacc[reg], 16 * s1 + j + 1);
                if (V == 6 && j > 0) T = __builtin_amdgcn_mfma_f64_16x16x4f64(ap_prev, T[(j - 2) >> 2], T, 0, 0, 0);
                // rows j and j+1 of the block, both visible in both of their 16-lane rows
                const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(dacc[reg]), __double2loint(dacc[reg]), false, false);
                const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(dacc[reg]), __double2hiint(dacc[reg]), false, false);
                const double rj = __hiloint2double(hi[0], lo[0]), rj1 = __hiloint2double(hi[1], lo[1]);      // a[j][c], a[j+1][c] at column c = li
                const double y0 = __builtin_amdgcn_rcp(a00);
                const double e0 = fma(-a00, y0, 1.0), l0 = a10 * y0, w0s = -rj * y0;
                const double p0 = fma(e0, e0, e0);
                const double l10 = fma(l0, p0, l0), w0 = fma(w0s, p0, w0s);      // a10 / a00 ; -a[j][c] / a00
                const double d2 = fma(-l10, a10, a11);
                const double t = fma(-l10, rj, rj1);      // a'[j+1][c]
                const double y1 = __builtin_amdgcn_rcp(d2);
                const double e1 = fma(-d2, y1, 1.0), u1s = -t * y1;
                const double p1 = fma(e1, e1, e1);
                double u1 = fma(u1s, p1, u1s);
                u1 = (li > j + 1) ? u1 : 0.0;
                const double u0 = fma(-u1, l10, w0);
                const double ap = (lk == s0) ? ((li > j) ? u0 : 0.0) : ((lk == s1) ? u1 : 0.0);
                dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap, dacc[reg], dacc, 0, 0, 0);
                ap_prev = ap;
                if (r == 0) { if (lk == s0) pivs[reg] = a00; if (lk == s1) pivs[reg] = d2; }
            }
            if (V == 6) T = __builtin_amdgcn_mfma_f64_16x16x4f64(ap_prev, T[3], T, 0, 0, 0);
        } else {
            double piv = bcast_lane(dacc[0], 0), inv = pivot_inverse(piv);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int reg = j >> 2, slot = j & 3;
                if (V == 1 && j > 0) T = __builtin_amdgcn_mfma_f64_16x16x4f64(us_prev, T[(j - 1) >> 2], T, 0, 0, 0);
                const double m = (lk == slot && li > j) ? dacc[reg] : 0.0;
                const double us = -m * inv;
                double ann = 1.0, ajn = 0.0;
                if (j < 15) { ann = bcast_lane(dacc[(j + 1) >> 2], 16 * ((j + 1) & 3) + j + 1); ajn = bcast_lane(dacc[reg], 16 * slot + j + 1); }
                dacc = __builtin_amdgcn_mfma_f64_16x16x4f64(us, dacc[reg], dacc, 0, 0, 0);
                us_prev = us;
                if (j < 15) { piv = fma(-(ajn * ajn), inv, ann); inv = pivot_inverse(piv); }
            }
        }
        if (V == 0) T = __builtin_amdgcn_mfma_f64_16x16x4f64(us_prev, T[3], T, 0, 0, 0);      // (a no-op: the last multiplier is zero)
        if (r == 0) for (int q = 0; q < 4; ++q) { out[64 + (lk + 4 * q) * 16 + li] = dacc[q]; out[64 + 256 + (lk + 4 * q) * 16 + li] = T[q]; out[64 + 512 + (lk + 4 * q) * 16 + li] = pivs[q]; }
        sink += dacc[3] + T[3];
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[0] = t1 - t0;
    out[lane] = sink;
}
int main() {
    std::vector<double> A(256);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 20.0 : 0.0) + 1.0 / (1 + i + j);
    double *dA, *dO; long long* dC;
    hipMalloc(&dA, 256 * 8); hipMalloc(&dO, (64 + 768) * 8); hipMalloc(&dC, 8);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    const int reps = 2000;
    std::vector<double> ref(768), cur(768);
    for (int v = 0; v < 8; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 3) hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 4) hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 5) hipLaunchKernelGGL(k_chain<5>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 6) hipLaunchKernelGGL(k_chain<6>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 7) hipLaunchKernelGGL(k_chain<7>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
        double o; hipMemcpy(&o, dO, 8, hipMemcpyDeviceToHost);
        printf("variant %d: %.1f cycles per pivot (%lld cycles / %d chains of 16; sink %.6g)\n", v, (double)c / reps / 16.0, c, reps, o);
        hipMemcpy(cur.data(), dO + 64, 768 * 8, hipMemcpyDeviceToHost);
        if (v == 0) ref = cur;
        if (v == 6) {      // rows of the eliminated block above their pivot (what becomes L), the elimination of the identity (what becomes W), the pivots
            double dl = 0, dw = 0, dp = 0;
            for (int j = 0; j < 16; ++j) for (int c = 0; c < 16; ++c) {
                if (c > j) dl = fmax(dl, fabs(cur[j * 16 + c] - ref[j * 16 + c]) / fabs(ref[j * 16 + c]));
                if (c < j) dw = fmax(dw, fabs(cur[256 + j * 16 + c] - ref[256 + j * 16 + c]));
                dp = fmax(dp, fabs(cur[512 + j * 16 + c] - ref[512 + j * 16 + c]) / fabs(ref[512 + j * 16 + c]));
            }
            printf("   pair pivots vs single pivots: rows of L rel %.2e, W abs %.2e, pivots rel %.2e\n", dl, dw, dp);
        }
    }
    return 0;
}
