// micro_dpp.hip -- latency / issue probes for the DPP FP64 operations used by tools/uvs_chol16.h (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "uvs_chol16.h"
using namespace uvsdev;
__device__ __forceinline__ double rsq3(double x) { const double y = __builtin_amdgcn_rsq(x); const double e = fma(-x * y, y, 1.0); return fma(y, e * fma(0.375, e, 0.5), y); }
template <int J> __device__ __forceinline__ double fmac_nonop(double acc, double a, double b) {
    asm("v_fmac_f64_dpp %0, %2, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b), "n"(J));
    return acc;
}
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int n) {
    const int lane = threadIdx.x;
    double x = 1.0 + lane * 1e-3, y = 0.5 + lane * 1e-4, z = 0.25;
    double a0 = x, a1 = y, a2 = z, a3 = x + y, a4 = x - y, a5 = y + z, a6 = x * y, a7 = x * z;
    long long t0, t1;
    t0 = clock64();
    for (int i = 0; i < n; ++i) { x = fmac_row_bcast<3, true>(x, y, z); x = fmac_row_bcast<5, true>(x, y, z); x = fmac_row_bcast<7, true>(x, y, z); x = fmac_row_bcast<9, true>(x, y, z); }
    t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;      // dependent DPP fmac (acc chain), with s_nop
    t0 = clock64();
    for (int i = 0; i < n; ++i) { a0 = fmac_row_bcast<3, true>(a0, y, z); a1 = fmac_row_bcast<5, true>(a1, y, z); a2 = fmac_row_bcast<7, true>(a2, y, z); a3 = fmac_row_bcast<9, true>(a3, y, z);
                                   a4 = fmac_row_bcast<3, true>(a4, y, z); a5 = fmac_row_bcast<5, true>(a5, y, z); a6 = fmac_row_bcast<7, true>(a6, y, z); a7 = fmac_row_bcast<9, true>(a7, y, z); }
    t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;      // 8 independent DPP fmac per iteration
    t0 = clock64();
    for (int i = 0; i < n; ++i) { a0 = fmac_nonop<3>(a0, y, z); a1 = fmac_nonop<5>(a1, y, z); a2 = fmac_nonop<7>(a2, y, z); a3 = fmac_nonop<9>(a3, y, z);
                                   a4 = fmac_nonop<3>(a4, y, z); a5 = fmac_nonop<5>(a5, y, z); a6 = fmac_nonop<7>(a6, y, z); a7 = fmac_nonop<9>(a7, y, z); }
    t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;      // same without the s_nop
    t0 = clock64();
    for (int i = 0; i < n; ++i) { x = fmac_row_bcast<3, true>(x, y, x); x = fmac_row_bcast<5, true>(x, y, x); x = fmac_row_bcast<7, true>(x, y, x); x = fmac_row_bcast<9, true>(x, y, x); }
    t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;      // dependent through the DPP operand
    t0 = clock64();
    for (int i = 0; i < n; ++i) { x = row_bcast<3, true>(x) + 1.0; x = row_bcast<5, true>(x) + 1.0; x = row_bcast<7, true>(x) + 1.0; x = row_bcast<9, true>(x) + 1.0; }
    t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;      // mov_dpp + add chain
    t0 = clock64();
    for (int i = 0; i < n; ++i) { x = rsq3(x + 2.0); x = rsq3(x + 2.0); x = rsq3(x + 2.0); x = rsq3(x + 2.0); }
    t1 = clock64(); if (lane == 0) cyc[5] = t1 - t0;      // add + rsqrt_chain
    t0 = clock64();
    for (int i = 0; i < n; ++i) { double r4[4]; rows_replicate(x, r4); x = r4[0] + r4[1] + r4[2] + r4[3]; rows_replicate(x, r4); x = r4[0] + r4[1] + r4[2] + r4[3]; }
    t1 = clock64(); if (lane == 0) cyc[6] = t1 - t0;      // rows_replicate + 3 adds (x2)
    t0 = clock64();
    for (int i = 0; i < n; ++i) { x = fma(x, 0.999999, y); x = fma(x, 0.999999, y); x = fma(x, 0.999999, y); x = fma(x, 0.999999, y); }
    t1 = clock64(); if (lane == 0) cyc[7] = t1 - t0;      // plain dependent FMA
    t0 = clock64();
    for (int i = 0; i < n; ++i) { a0 = fma(a0, 0.999999, y); a1 = fma(a1, 0.999999, y); a2 = fma(a2, 0.999999, y); a3 = fma(a3, 0.999999, y); a4 = fma(a4, 0.999999, y); a5 = fma(a5, 0.999999, y); a6 = fma(a6, 0.999999, y); a7 = fma(a7, 0.999999, y); }
    t1 = clock64(); if (lane == 0) cyc[8] = t1 - t0;      // 8 independent FMA
    out[lane] = x + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    double* o; long long* c; hipMalloc(&o, 64 * 8); hipMalloc(&c, 16 * 8);
    const int n = 1000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, n); hipDeviceSynchronize();
    long long h[16]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[9] = {"dep fmac_dpp (acc chain, s_nop) /op", "indep fmac_dpp x8 (s_nop) /op", "indep fmac_dpp x8 (no nop) /op", "dep via dpp operand /op", "mov_dpp+add /pair", "add+rsqrt_chain /pair", "rows_replicate+3add /group", "dep fma /op", "indep fma x8 /op"};
    const int per[9] = {4, 8, 8, 4, 4, 4, 2, 4, 8};
    for (int i = 0; i < 9; ++i) printf("%-40s %.1f cycles\n", nm[i], (double)h[i] / (n * per[i]));
    return 0;
}
