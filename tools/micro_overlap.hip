// micro_overlap.hip -- does a host-to-device copy on one stream run WHILE a long kernel occupies the device on another?
// (round 5: uvs_batch_stream's copy of batch k + 1 starts when k_solve of batch k ends -- profiles/r05_stream_timeline.txt.  This probe takes the
// solver out of the picture: a kernel that only spins for a given time, with the launch shape of k_solve or smaller ones, beside a copy of the
// size of a batch.)
//   hipcc --offload-arch=gfx950 -O2 -o micro_overlap micro_overlap.hip && ./micro_overlap
// Prints, per kernel shape: kernel alone, copy alone, both enqueued back to back (kernel first) on two streams, and the same with the copy first.
// overlap = (alone_k + alone_c - both) / min(alone_k, alone_c): 1 = fully hidden, 0 = serial.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void k_spin(long long ticks, double* sink, int touch_lds) {
    extern __shared__ double sh[];
    const long long t0 = wall_clock64();
    double a = threadIdx.x;
    while (wall_clock64() - t0 < ticks) {
        a = a * 1.0000001 + 1e-9;
        if (touch_lds) sh[threadIdx.x] = a;
    }
    if (a == 12345.678) sink[0] = a;
}

// the same with k_solve's register footprint: 128 + 128 registers per lane at 512 threads = both waves of a SIMD hold its whole register file, no other wave fits on the CU
__global__ void __launch_bounds__(512) k_spin_fat(long long ticks, double* sink, int touch_lds) {
    extern __shared__ double sh[];
    const long long t0 = wall_clock64();
    double a = threadIdx.x;
    asm volatile("v_mov_b32 v127, 0\n v_accvgpr_write_b32 a127, 0" ::: "v127", "a127");
    while (wall_clock64() - t0 < ticks) {
        a = a * 1.0000001 + 1e-9;
        if (touch_lds) sh[threadIdx.x] = a;
    }
    if (a == 12345.678) sink[0] = a;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? std::atol(argv[1]) : 46) * 1000000ull;
    const double kernel_ms = argc > 2 ? std::atof(argv[2]) : 1.5;
    int clk_khz = 100000;      // wall_clock64: 100 MHz on gfx950
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0));
    const long long ticks = (long long)(kernel_ms * clk_khz);
    char *h = nullptr, *d = nullptr; double* sink = nullptr;
    CK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault)); CK(hipMalloc((void**)&d, bytes)); CK(hipMalloc((void**)&sink, 64));
    for (size_t i = 0; i < bytes; i += 4096) h[i] = (char)i;
    hipStream_t sk, sc; CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)k_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); CK(hipFuncSetAttribute((const void*)k_spin_fat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct Shape { int grid, block; size_t lds; const char* what; int fat; };
    const Shape shapes[] = { {256, 512, 158 * 1024, "k_solve's shape: 256 x 512 threads, 158 KB LDS", 0}, {256, 512, 158 * 1024, "the same with 256 registers per lane (CU full)", 1}, {248, 512, 158 * 1024, "256 registers per lane, 248 workgroups", 1}, {256, 512, 0, "256 x 512 threads, no LDS"}, {128, 512, 158 * 1024, "128 workgroups (half the CUs idle)"},
                             {32, 64, 0, "32 x 64 threads"}, {1, 64, 0, "one wavefront"} };
    std::printf("wall clock %d kHz, copy %zu bytes, kernel %.2f ms; HSA_ENABLE_SDMA=%s\n", clk_khz, bytes, kernel_ms, std::getenv("HSA_ENABLE_SDMA") ? std::getenv("HSA_ENABLE_SDMA") : "(unset)");
    for (const Shape& s : shapes) {
        auto kern = [&]() { if (s.fat) hipLaunchKernelGGL(k_spin_fat, dim3(s.grid), dim3(s.block), s.lds, sk, ticks, sink, 1); else hipLaunchKernelGGL(k_spin, dim3(s.grid), dim3(s.block), s.lds, sk, ticks, sink, s.lds ? 1 : 0); };
        auto copy = [&]() { CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, sc)); };
        double tk = 1e9, tc = 1e9, tkc = 1e9, tck = 1e9, tthr = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipDeviceSynchronize()); double t0 = now_ms(); kern(); CK(hipDeviceSynchronize()); double t = now_ms() - t0; if (rep && t < tk) tk = t;
            t0 = now_ms(); copy(); CK(hipDeviceSynchronize()); t = now_ms() - t0; if (rep && t < tc) tc = t;
            t0 = now_ms(); kern(); copy(); CK(hipDeviceSynchronize()); t = now_ms() - t0; if (rep && t < tkc) tkc = t;
            t0 = now_ms(); copy(); kern(); CK(hipDeviceSynchronize()); t = now_ms() - t0; if (rep && t < tck) tck = t;
            // the copy enqueued by another host thread while this one has launched the kernel
            t0 = now_ms(); kern(); { std::thread th([&]() { copy(); CK(hipStreamSynchronize(sc)); }); CK(hipStreamSynchronize(sk)); th.join(); } t = now_ms() - t0; if (rep && t < tthr) tthr = t;
        }
        auto ov = [&](double both) { return (tk + tc - both) / (tk < tc ? tk : tc); };
        std::printf("%-50s kernel %.3f  copy %.3f (%.1f GB/s)  kernel+copy %.3f (overlap %.2f)  copy+kernel %.3f (overlap %.2f)  copy from a 2nd thread %.3f (overlap %.2f)\n", s.what, tk, tc, bytes / tc * 1e-6, tkc,
                    ov(tkc), tck, ov(tck), tthr, ov(tthr));
    }
    // several batches in flight, as the stream does: kernel k beside copy k + 1, ten rounds, events chaining them like uvs_batch_stream's two buffer sets
    {
        const Shape& s = shapes[0];
        hipEvent_t ec[2], ek[2]; for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&ec[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ek[i], hipEventDisableTiming)); }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize()); const double t0 = now_ms();
            const int R = 10;
            for (int b = 0; b < R; ++b) {
                const int set = b & 1;
                if (b >= 2) CK(hipStreamWaitEvent(sc, ek[set], 0));       // the buffer set is free when the kernel that read it has ended
                CK(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, sc)); CK(hipEventRecord(ec[set], sc));
                CK(hipStreamWaitEvent(sk, ec[set], 0));
                hipLaunchKernelGGL(k_spin_fat, dim3(s.grid), dim3(s.block), s.lds, sk, ticks, sink, 1); CK(hipEventRecord(ek[set], sk));
            }
            CK(hipDeviceSynchronize());
            std::printf("pipeline of %d batches (two buffer sets, events): %.3f ms per batch\n", R, (now_ms() - t0) / R);
        }
    }
    // the arrangement uvs_batch_stream had until round 5: TWO streams, each carrying copy -> kernel -> small D2H of its own batches, the host waiting for batch k - 2 before it enqueues batch k
    {
        const Shape& s = shapes[0];
        char* d2 = nullptr; CK(hipMalloc((void**)&d2, bytes));
        char* hout = nullptr; CK(hipHostMalloc((void**)&hout, 1 << 20, hipHostMallocDefault));
        hipStream_t st[2] = {sk, sc};
        for (int variant = 0; variant < 2; ++variant)
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipDeviceSynchronize()); const double t0 = now_ms();
                const int R = 10;
                for (int b = 0; b < R; ++b) {
                    const int q = b & 1;
                    if (b >= 2) CK(hipStreamSynchronize(st[q]));
                    if (variant == 1) std::this_thread::sleep_for(std::chrono::microseconds(1200));      // the host's packing time before it can enqueue the copy
                    CK(hipMemcpyAsync(q ? d2 : d, h, bytes, hipMemcpyHostToDevice, st[q]));
                    hipLaunchKernelGGL(k_spin_fat, dim3(s.grid), dim3(s.block), s.lds, st[q], ticks, sink, 1);
                    CK(hipMemcpyAsync(hout, q ? d2 : d, 1 << 18, hipMemcpyDeviceToHost, st[q]));
                }
                CK(hipDeviceSynchronize());
                std::printf("two streams, each copy -> kernel -> D2H%s: %.3f ms per batch\n", variant ? ", 1.2 ms of host work before every enqueue" : "", (now_ms() - t0) / R);
            }
    }
    return 0;
}
