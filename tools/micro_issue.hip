// micro_issue.hip -- FP64 VALU issue rate of one SIMD with 1, 2 and 4 resident waves (gfx950): is a single wave per SIMD issue-limited?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(double* out, long long* cyc, int n) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double y = 0.5 + lane * 1e-4;
    double a0 = y, a1 = y + 1, a2 = y + 2, a3 = y + 3, a4 = y + 4, a5 = y + 5, a6 = y + 6, a7 = y + 7;
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) { a0 = fma(a0, 0.999999, y); a1 = fma(a1, 0.999999, y); a2 = fma(a2, 0.999999, y); a3 = fma(a3, 0.999999, y); a4 = fma(a4, 0.999999, y); a5 = fma(a5, 0.999999, y); a6 = fma(a6, 0.999999, y); a7 = fma(a7, 0.999999, y); }
    long long t1 = clock64();
    if (lane == 0) cyc[wv] = t1 - t0;
    float f0 = (float)y, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7, fy = (float)y;
    __syncthreads();
    t0 = clock64();
    for (int i = 0; i < n; ++i) { f0 = fmaf(f0, 0.999f, fy); f1 = fmaf(f1, 0.999f, fy); f2 = fmaf(f2, 0.999f, fy); f3 = fmaf(f3, 0.999f, fy); f4 = fmaf(f4, 0.999f, fy); f5 = fmaf(f5, 0.999f, fy); f6 = fmaf(f6, 0.999f, fy); f7 = fmaf(f7, 0.999f, fy); }
    t1 = clock64();
    if (lane == 0) cyc[16 + wv] = t1 - t0;
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
int main() {
    double* o; long long* c; hipMalloc(&o, 1024 * 8); hipMalloc(&c, 32 * 8);
    const int n = 2000;
    for (int nt : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, o, c, n); hipDeviceSynchronize();
        long long h[32]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        printf("%4d threads (%d waves/SIMD): f64 fma %.2f cycles/instr/wave (wave 0), last wave %.2f ; f32 fma %.2f\n", nt, (nt + 255) / 256, (double)h[0] / (8.0 * n), (double)h[nt / 64 - 1] / (8.0 * n), (double)h[16] / (8.0 * n));
    }
    return 0;
}
