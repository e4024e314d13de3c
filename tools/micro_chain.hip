// micro_chain.hip -- cycles per pivot of the 16-pivot chain of the reduced solve's diagonal-block factorization (uvs_solve_kernel.h: chol_factor_impl S2),
// one wavefront on one SIMD, for the variants that were considered:
//   0  round-2 form: readlane of the pivot out of the MFMA result -> rcp -> e -> e + e^2 -> multiplier -> rank-1 MFMA  (+ the trailing MFMA that builds W = L^-1)
//   1  pivot AHEAD of the matrix: next pivot = a[j+1][j+1] - a[j][j+1]^2 / piv_j from values read before update j; its reciprocal overlaps the MFMA
//   2  variant 1 without the W MFMA (timing only: how much of a link is the second MFMA's pipe occupancy)
//   3  variant 0 without the W MFMA
// build: hipcc --offload-arch=gfx950 -O3 tools/micro_chain.hip -o tools/micro_chain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
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
    hipMalloc(&dA, 256 * 8); hipMalloc(&dO, 64 * 8); hipMalloc(&dC, 8);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int v = 0; v < 6; ++v) {
        for (int rep = 0; rep < 2; ++rep) {
            if (v == 0) hipLaunchKernelGGL(k_chain<0>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 1) hipLaunchKernelGGL(k_chain<1>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 2) hipLaunchKernelGGL(k_chain<2>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 3) hipLaunchKernelGGL(k_chain<3>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 4) hipLaunchKernelGGL(k_chain<4>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            if (v == 5) hipLaunchKernelGGL(k_chain<5>, dim3(1), dim3(64), 0, 0, dA, dO, dC, reps);
            hipDeviceSynchronize();
        }
        long long c; hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
        double o; hipMemcpy(&o, dO, 8, hipMemcpyDeviceToHost);
        printf("variant %d: %.1f cycles per pivot (%lld cycles / %d chains of 16; sink %.6g)\n", v, (double)c / reps / 16.0, c, reps, o);
    }
    return 0;
}
