// uvs_solve512d.hip -- the persistent LM kernel once more with 512 threads, this time WITH the dense path (uvs_layout.h: UVS_DS_*; uvs_solve_kernel.h: dense_schur / dense_direct):
// the landmark Schur complement and the direct J^T J / J^T r terms on the FP64 matrix cores, no gather lists.  What uvs_create selects with UVS_DENSE_SCHUR=1.  A translation
// unit (and namespace) of its own because the dense code paths, compiled into the default 512-thread kernel, cost its list walk 1.7 % (same-box A/B, profiles/r05_ab_cur_vs_r04.txt:
// instruction footprint); same build flags as uvs_solve512.hip.
#define UVS_NT 512
#define UVS_ALLOW_EXPERIMENTAL_NT 1
#define UVS_SOLVE_KERNEL_ONLY 1
#define UVS_CHUNK_TOUCH 1
#define UVS_DENSE_TU 1
#define uvsdev uvsdev512d
#include "uvs_solve_kernel.h"

using namespace uvsdev512d;

extern "C" {
int uvs_k_solve512d_init(const unsigned char* fa, const unsigned char* fb, int n) {
    if (n != UVS_NBLK) return UVS_ERR_INVALID_ARG;
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_blk_fa), fa, n) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(c_blk_fb), fb, n) != hipSuccess) return UVS_ERR_HIP;
    if (hipFuncSetAttribute((const void*)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) return UVS_ERR_HIP;
    return UVS_OK;
}
// kopts / dbg: the caller's uvsdev::KOpts / uvsdev::DebugOut (same definitions, other namespace); returns UVS_ERR_INVALID_ARG on a layout mismatch (nothing is launched)
int uvs_k_solve512d_launch(int n_windows, hipStream_t stream, char* blobs, const long long* blob_off, double* ws_all, const long long* ws_off,
                           const void* kopts, size_t kopts_bytes, uvs_report* reports, const void* dbg, size_t dbg_bytes) {
    KOpts ko; DebugOut d;
    if (kopts_bytes != sizeof(ko) || dbg_bytes != sizeof(d)) return UVS_ERR_INVALID_ARG;
    __builtin_memcpy(&ko, kopts, sizeof(ko)); __builtin_memcpy(&d, dbg, sizeof(d));
    hipLaunchKernelGGL(k_solve, dim3(n_windows), dim3(NT), LDS_BYTES, stream, blobs, blob_off, ws_all, ws_off, ko, reports, d);
    return UVS_OK;
}
int uvs_k_solve512d_timeline(long long* out, size_t n) {
    if (n != sizeof(g_lin_tl) / sizeof(long long)) return UVS_ERR_INVALID_ARG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lin_tl), n * sizeof(long long)) == hipSuccess ? UVS_OK : UVS_ERR_HIP;
}
}
