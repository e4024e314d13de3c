// uvs_solve512.hip -- the persistent LM kernel (uvs_solve_kernel.h: k_solve) instantiated with 512 threads per workgroup: two resident
// wavefronts per SIMD, 256 registers per lane, waves 0..3 evaluate observations and waves 4..7 own the gather groups (ROLES).
//
// It is its own translation unit because the workgroup size is a compile-time constant of the kernel header (LDS map, loop strides, batch
// sizes): everything else in the library -- the landmark-sharded kernels, k_evaluate, k_marg_linearize, the 256-thread k_solve that
// UVS_KSOLVE_NT=256 selects for A/B runs -- is built with 256 threads in uvs_solver.hip.  The namespace is renamed so that the two
// instantiations do not collide at link time; the blob and workspace layout (uvs_layout.h: UVS_GT = 256 gather threads either way), the
// kernel arguments and the report are the same, so the host side only picks which launcher to call (launch_solve).
//
// Build flag of THIS file: -mllvm -disable-machine-licm.  Machine LICM hoists loop-invariant address arithmetic out of the LM loop -- the
// whole kernel body -- and the hoisted values (hundreds) are live across every phase: 428 spilled VGPRs with it, 46 without (round 4);
// -mllvm -sink-insts-to-avoid-spills takes another dozen away (34).
#define UVS_NT 512
#define UVS_ALLOW_EXPERIMENTAL_NT 1
#define UVS_SOLVE_KERNEL_ONLY 1
#define UVS_CHUNK_TOUCH 1
#define UVS_TU_512 1
#define uvsdev uvsdev512
#include "uvs_solve_kernel.h"
#include "uvs_large_kernel.h"      // k_large_chunks (the chunk kernel of the landmark-sharded forms with the same wave roles) and k_large_solve (UVS_TU_512)

using namespace uvsdev512;

extern "C" {
// block table of the output-stationary gather (the __constant__ copies of this translation unit) + the LDS opt-in; once per device
int uvs_k_solve512_init(const unsigned char* fa, const unsigned char* fb, int n) {
    if (n != UVS_NBLK) return UVS_ERR_INVALID_ARG;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_blk_fa), fa, n) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(c_blk_fb), fb, n) != hipSuccess) return UVS_ERR_HIP;
    if (hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) return UVS_ERR_HIP;
    if (hipFuncSetAttribute((const void*)k_large_chunks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) return UVS_ERR_HIP;
    if (hipFuncSetAttribute((const void*)k_large_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) return UVS_ERR_HIP;
    return UVS_OK;
}
// k_large_chunks with 512 threads per workgroup (grid = chunk workgroups + the frame-terms workgroup, as for the 256-thread kernel)
int uvs_k_large_chunks512_launch(int grid, hipStream_t stream, char* blob, double* ws, const void* kopts, size_t kopts_bytes, const double* state, int sel, int first, double radius,
                                 double* partials, const double* ctl, int rank, int nranks, int n_chunk_wgs, double* fimg) {
    KOpts ko;
    if (kopts_bytes != sizeof(ko)) return UVS_ERR_INVALID_ARG;      // (nothing is launched: the caller must not read results)
    __builtin_memcpy(&ko, kopts, sizeof(ko));
    hipLaunchKernelGGL(k_large_chunks, dim3(grid), dim3(NT), LDS_BYTES, stream, blob, ws, ko, state, sel, first, radius, partials, LargeCtl{ctl, rank, nranks}, n_chunk_wgs, fimg);
    return UVS_OK;
}
// kopts / dbg: the caller's uvsdev::KOpts / uvsdev::DebugOut (same definitions, other namespace)
int uvs_k_solve512_launch(int n_windows, hipStream_t stream, char* blobs, const long long* blob_off, double* ws_all, const long long* ws_off,
                          const void* kopts, size_t kopts_bytes, uvs_report* reports, const void* dbg, size_t dbg_bytes) {
    KOpts ko; DebugOut d;
    if (kopts_bytes != sizeof(ko) || dbg_bytes != sizeof(d)) return UVS_ERR_INVALID_ARG;      // (nothing is launched: the caller reports an error instead of downloading stale reports)
    __builtin_memcpy(&ko, kopts, sizeof(ko)); __builtin_memcpy(&d, dbg, sizeof(d));
    hipLaunchKernelGGL(k_solve, dim3(n_windows), dim3(NT), LDS_BYTES, stream, blobs, blob_off, ws_all, ws_off, ko, reports, d);
    return UVS_OK;
}
size_t uvs_k_solve512_arg_bytes(int which) { return which == 0 ? sizeof(KOpts) : sizeof(DebugOut); }
// k_large_solve with 512 threads (one workgroup): the frame image arrives with twice the loads in flight, the factorization has six workers and the pivot chain its SIMD alone
int uvs_k_large_solve512_launch(hipStream_t stream, char* blob, double* ws, const void* kopts, size_t kopts_bytes, double* state, const double* reduced, int first, double radius, double* out,
                                const double* ctl, int rank, int nranks, const double* fimg) {
    KOpts ko;
    if (kopts_bytes != sizeof(ko)) return UVS_ERR_INVALID_ARG;
    __builtin_memcpy(&ko, kopts, sizeof(ko));
    hipLaunchKernelGGL(k_large_solve, dim3(1), dim3(NT), LDS_BYTES, stream, blob, ws, ko, state, reduced, first, radius, out, LargeCtl{ctl, rank, nranks}, fimg);
    return UVS_OK;
}
// debug == 7 (UVS_LARGE_PROF): the per-workgroup stamps of the last k_large_chunks launch (tools/large_timeline.py)
int uvs_k_large_chunks512_prof(long long* out, size_t n) {
    if (n != sizeof(g_large_prof) / sizeof(long long)) return UVS_ERR_INVALID_ARG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_large_prof), n * sizeof(long long)) == hipSuccess ? UVS_OK : UVS_ERR_HIP;
}
// debug == 5: the per-wave step log of the last launch (tools/lin_timeline.py)
int uvs_k_solve512_timeline(long long* out, size_t n) {
    if (n != sizeof(g_lin_tl) / sizeof(long long)) return UVS_ERR_INVALID_ARG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lin_tl), n * sizeof(long long)) == hipSuccess ? UVS_OK : UVS_ERR_HIP;
}
}
