// Does the shader clock drop when all 256 CUs run the solver's kind of work (FP64 FMA chains, one wavefront per SIMD, one workgroup per CU)?
// clock64() counts shader cycles, wall_clock64() a constant 100 MHz: their ratio per workgroup is the clock the workgroup actually saw.
//   hipcc --offload-arch=gfx950 -O3 tools/micro_clock.hip -o /tmp/micro_clock && /tmp/micro_clock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ __launch_bounds__(256) void k(double* out, long long* t, int iters, int mfma) {
    extern __shared__ double sh[];
    double a = threadIdx.x * 1e-3, b = 1.0000001, c = 0.5;
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {0, 0, 0, 0};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (mfma) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
        else { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    sh[threadIdx.x] = a + acc[0];
    out[blockIdx.x * 256 + threadIdx.x] = sh[threadIdx.x];
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    double* out; long long* t; hipMalloc(&out, 1024 * 256 * 8); hipMalloc(&t, 1024 * 16);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int mfma = 0; mfma < 2; ++mfma)
    for (int n : {1, 8, 32, 64, 128, 256, 512}) {
        std::vector<long long> h(2 * n);
        for (int rep = 0; rep < 3; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(n), dim3(256), 150 * 1024, 0, out, t, 400000, mfma); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), t, 16 * n, hipMemcpyDeviceToHost);
            double fmin = 1e9, fmax = 0, cyc = 0;
            for (int i = 0; i < n; ++i) { const double f = (double)h[2 * i] / ((double)h[2 * i + 1] / 100e6) / 1e9; fmin = std::min(fmin, f); fmax = std::max(fmax, f); cyc += h[2 * i]; }
            if (rep == 2) printf("%s workgroups %4d: %.3f ms, shader clock %.3f .. %.3f GHz, cycles per workgroup %.0f\n", mfma ? "mfma" : "fma ", n, ms, fmin, fmax, cyc / n);
        }
    }
    return 0;
}
