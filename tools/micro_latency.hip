// micro_latency.hip -- instruction latency probes for gfx950 (used to size the serial chains of the dense solve).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro_latency.hip -o gpurun_out/micro_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));

__global__ void k_probe(double* out, long long* cyc, int n) {
    const int lane = threadIdx.x & 63;
    double x = 1.0 + lane * 1e-3, y = 0.5;
    long long t0, t1;
    // (a) dependent FMA f64 chain
    t0 = clock64();
    for (int i = 0; i < n; ++i) { x = fma(x, 0.999999, y); x = fma(x, 0.999999, y); x = fma(x, 0.999999, y); x = fma(x, 0.999999, y); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[0] = t1 - t0;
    // (b) dependent rcp f64 chain
    t0 = clock64();
    for (int i = 0; i < n; ++i) { x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); x = __builtin_amdgcn_rcp(x); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[1] = t1 - t0;
    // (c) dependent MFMA f64 16x16x4 chain (same accumulator)
    d4_t acc = {x, y, x, y};
    t0 = clock64();
    for (int i = 0; i < n; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
                                   acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[2] = t1 - t0;
    // (d) two independent MFMA chains
    d4_t acc2 = {y, x, y, x};
    t0 = clock64();
    for (int i = 0; i < n; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc2, 0, 0, 0);
                                   acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc2, 0, 0, 0); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[3] = t1 - t0;
    // (e) MFMA -> readlane -> VALU -> MFMA round trip (the pivot chain skeleton)
    t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(acc[0]), 5), hi = __builtin_amdgcn_readlane(__double2hiint(acc[0]), 5);
            const double p = __hiloint2double(hi, lo);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(p * 1e-9, 1e-3, acc, 0, 0, 0);
        }
    }
    t1 = clock64(); if (threadIdx.x == 0) cyc[4] = t1 - t0;
    // (f) dependent rsq f64
    t0 = clock64();
    for (int i = 0; i < n; ++i) { x = __builtin_amdgcn_rsq(x + 2.0); x = __builtin_amdgcn_rsq(x + 2.0); x = __builtin_amdgcn_rsq(x + 2.0); x = __builtin_amdgcn_rsq(x + 2.0); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[5] = t1 - t0;
    // (g) LDS write -> read round trip (same wave)
    __shared__ double sh[1024];
    t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { sh[lane * 9 + u] = x; x = sh[((lane + 1) & 63) * 9 + u] + 1.0; }
    }
    t1 = clock64(); if (threadIdx.x == 0) cyc[6] = t1 - t0;
    // (h) ds_bpermute round trip
    t0 = clock64();
    int iv = lane;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) iv = __builtin_amdgcn_ds_bpermute(((iv + 1) & 63) << 2, iv);
    }
    t1 = clock64(); if (threadIdx.x == 0) cyc[7] = t1 - t0;
    // (i) __syncthreads cost (all waves arrive together)
    t0 = clock64();
    for (int i = 0; i < n; ++i) { __syncthreads(); __syncthreads(); __syncthreads(); __syncthreads(); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[8] = t1 - t0;
    // (j) independent FMA f64 throughput (8 chains)
    double z[8]; for (int u = 0; u < 8; ++u) z[u] = x + u;
    t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) z[u] = fma(z[u], 0.999999, y);
    }
    t1 = clock64(); if (threadIdx.x == 0) cyc[9] = t1 - t0;
    for (int u = 0; u < 8; ++u) x += z[u];
    // accuracy of the hardware seeds
    double worst_rcp = 0.0, worst_rsq = 0.0;
    for (int i = 0; i < 2000; ++i) {
        const double v = 1.0 + (i * 64 + lane) * (1.0 / 128000.0) * 3.0;
        const double r = __builtin_amdgcn_rcp(v), q = __builtin_amdgcn_rsq(v);
        worst_rcp = fmax(worst_rcp, fabs(r * v - 1.0));
        worst_rsq = fmax(worst_rsq, fabs(q * q * v - 1.0) * 0.5);
    }
    out[threadIdx.x] = x + acc[0] + acc[1] + acc2[0] + iv;
    if (threadIdx.x == 0) { out[1024] = worst_rcp; out[1025] = worst_rsq; }
}

int main() {
    double* out; long long* cyc;
    hipMalloc(&out, 2048 * 8); hipMalloc(&cyc, 16 * 8);
    const int n = 1000;
    for (int nt : {64, 512}) {
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(nt), 0, 0, out, cyc, n);
        hipDeviceSynchronize();
        long long h[16]; double ho[2];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        hipMemcpy(ho, out + 1024, sizeof(ho), hipMemcpyDeviceToHost);
        const char* names[] = {"dependent fma f64", "dependent rcp f64", "dependent mfma f64 16x16x4", "2 independent mfma chains (per mfma)", "mfma->readlane->mul->mfma",
                               "dependent rsq f64 (+add)", "lds write->read round trip", "ds_bpermute round trip", "__syncthreads", "independent fma f64 (per fma)"};
        printf("threads per workgroup %d\n", nt);
        for (int i = 0; i < 10; ++i) printf("  %-40s %8.1f cycles\n", names[i], (double)h[i] / (i == 9 ? 8.0 * n : 4.0 * n));
        printf("  seed accuracy: rcp %.3e rsq %.3e\n", ho[0], ho[1]);
    }
    return 0;
}
