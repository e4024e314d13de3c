// chol16_test.hip -- stand-alone check + timing of tools/uvs_chol16.h (the 16x16 diagonal-block factorization of the reduced solve).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chol16_test.hip -o gpurun_out/chol16_test ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "uvs_chol16.h"
using namespace uvsdev;

__global__ __launch_bounds__(64) void k_test(const double* A, double* out /*[n][16*17 + 16 + 1]*/, long long* cyc, int reps) {
    __shared__ double blk[16 * 17 + 16];
    const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
    const double* Ab = A + 256 * blockIdx.x;
    d4c_t acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = Ab[(lk + 4 * q) * 16 + li];
    bool ok = true;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        ok = chol16_factor(acc, lane, blk, 17, blk + 16 * 17) && ok;
        if (r + 1 < reps) acc[0] += 1e-300 * blk[li];      // serialise the repetitions
    }
    const long long t1 = clock64();
    __syncthreads();
    double* o = out + (size_t)blockIdx.x * (16 * 17 + 17);
    for (int t = lane; t < 16 * 17 + 16; t += 64) o[t] = blk[t];
    if (lane == 0) { o[16 * 17 + 16] = ok ? 1.0 : 0.0; cyc[blockIdx.x] = (t1 - t0) / reps; }
    // timing variants: without the W side chain; the row replication alone
    __syncthreads();
    const long long t2 = clock64();
    for (int r = 0; r < reps; ++r) { ok = chol16_factor<false>(acc, lane, blk, 17, blk + 16 * 17) && ok; if (r + 1 < reps) acc[0] += 1e-300 * blk[li]; }
    const long long t3 = clock64();
    double sum = 0.0;
    for (int r = 0; r < reps; ++r) { for (int q = 0; q < 4; ++q) { double r4[4]; rows_replicate(acc[q] + sum, r4); sum += r4[0] + r4[1] + r4[2] + r4[3]; } }
    const long long t4 = clock64();
    if (lane == 0) { cyc[gridDim.x + blockIdx.x] = (t3 - t2) / reps; cyc[2 * gridDim.x + blockIdx.x] = (t4 - t3) / reps; o[0] += (ok ? 0.0 : 1e-300) + 1e-300 * sum; }
}

int main() {
    const int n = 8, reps = 64;
    std::vector<double> A(256 * n);
    srand(7);
    for (int b = 0; b < n; ++b) {
        double M[16][16];
        for (auto& r : M) for (double& v : r) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += M[i][k] * M[j][k]; A[256 * b + 16 * i + j] = s * std::pow(10.0, (i + j) * 0.25 * (b % 3)) + (i == j ? 1e-3 : 0.0); }
    }
    double *dA, *dO; long long* dC;
    const size_t on = (size_t)n * (16 * 17 + 17);
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dO, on * 8); hipMalloc(&dC, 3 * n * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_test, dim3(n), dim3(64), 0, 0, dA, dO, dC, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<double> O(on); std::vector<long long> C(3 * n);
    hipMemcpy(O.data(), dO, on * 8, hipMemcpyDeviceToHost); hipMemcpy(C.data(), dC, 3 * n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < n; ++b) {
        // CPU Cholesky + inverse
        double L[16][16] = {}, W[16][16] = {};
        for (int j = 0; j < 16; ++j) {
            double d = A[256 * b + 17 * j]; for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
            L[j][j] = std::sqrt(d);
            for (int i = j + 1; i < 16; ++i) { double s = A[256 * b + 16 * i + j]; for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k]; L[i][j] = s / L[j][j]; }
        }
        for (int c = 0; c < 16; ++c) for (int r = c; r < 16; ++r) { double s = (r == c) ? 1.0 : 0.0; for (int k = c; k < r; ++k) s -= L[r][k] * W[k][c]; W[r][c] = s / L[r][r]; }
        const double* o = O.data() + (size_t)b * (16 * 17 + 17);
        double eL = 0, eW = 0, eD = 0, nL = 0, nW = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            if (j <= i) { eL = std::fmax(eL, std::fabs(o[i * 17 + j] - L[i][j])); nL = std::fmax(nL, std::fabs(L[i][j])); }
            else { eW = std::fmax(eW, std::fabs(o[i * 17 + j] - W[j][i])); nW = std::fmax(nW, std::fabs(W[j][i])); }
        }
        for (int j = 0; j < 16; ++j) eD = std::fmax(eD, std::fabs(o[16 * 17 + j] * L[j][j] - 1.0));
        const bool good = eL <= 1e-12 * nL && eW <= 1e-10 * nW && eD <= 1e-12 && o[16 * 17 + 16] == 1.0;
        printf("block %d: |dL| %.2e (of %.2e)  |dW| %.2e (of %.2e)  |dinv L - 1| %.2e  ok %g  cycles/factor %lld (no W %lld, replicate x4 + adds %lld)  %s\n", b, eL, nL, eW, nW, eD, o[16 * 17 + 16], C[b], C[n + b], C[2 * n + b], good ? "PASS" : "FAIL");
        bad += !good;
    }
    printf(bad ? "FAILED\n" : "ALL PASS\n");
    return bad != 0;
}
