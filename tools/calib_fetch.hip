// calib_fetch.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS code's access widths (MI355X_MICROARCH.md, HBM section:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths and WRITE_SIZE are
// uncalibrated: calibrate on a known byte count in your own access pattern").  k_solve issues 8-byte loads / stores almost exclusively.
//   hipcc --offload-arch=gfx950 -O3 -o calib_fetch tools/calib_fetch.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_f -- ./calib_fetch
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_w -- ./calib_fetch
// Each kernel moves exactly BYTES bytes (printed); compare with the counter value of its dispatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read8(const double* __restrict__ p, size_t n, double* out) {      // 8 B / lane, coalesced
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void read16(const double2* __restrict__ p, size_t n, double* out) {    // 16 B / lane, coalesced (the guide's calibrated case)
    double s = 0.0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
__global__ void write8(double* __restrict__ p, size_t n) {                        // 8 B / lane, coalesced
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)i;
}
__global__ void read8_strided(const double* __restrict__ p, size_t n, double* out) {   // 8 B / lane, lane stride 240 B (a per-lane record walk)
    double s = 0.0;
    const size_t rec = 30;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i * rec < n; i += (size_t)gridDim.x * blockDim.x)
        for (size_t q = 0; q < rec; ++q) s += p[i * rec + q];
    if (s == 12345.678) out[0] = s;
}

int main() {
    const size_t bytes = (size_t)1 << 30;      // 1 GiB: four times the Infinity Cache
    const size_t n = bytes / 8;
    double *p, *out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); return 1; }
    (void)hipMemset(p, 0, bytes);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(read8, dim3(4096), dim3(256), 0, 0, p, n, out);
    hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const double2*)p, n / 2, out);
    hipLaunchKernelGGL(write8, dim3(4096), dim3(256), 0, 0, p, n);
    hipLaunchKernelGGL(read8_strided, dim3(4096), dim3(256), 0, 0, p, n - (n % 30), out);
    (void)hipDeviceSynchronize();
    std::printf("BYTES per kernel: %zu\n", bytes);
    return 0;
}
