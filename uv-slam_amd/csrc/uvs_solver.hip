// uvs_solver.hip -- C ABI (include/uvs_solver.h) + host-side packing for the MI355X sliding-window solver.
//
// Build (see __graft_entry__.build()):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC uvs_solver.hip -o ../libuvs_solver.so
//
// There is NO CPU path in this library: uvs_create() fails with UVS_ERR_NO_DEVICE when no HIP device
// is present and every compute entry point runs HIP kernels.  The CPU oracle under oracle/ is test
// infrastructure and is never linked or called from here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <thread>
#include <vector>
#include <sched.h>
#include <pthread.h>
#include <future>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>

#include "../../include/uvs_solver.h"
#include "uvs_layout.h"
#include "uvs_factors.h"
#include "uvs_solve_kernel.h"
#include "uvs_eval_kernel.h"
#include "uvs_large_kernel.h"
#include "uvs_marg_kernel.h"
#include "uvs_marg.h"      // LAST: its file-scope `#pragma clang fp contract(...)` must not reach any device code (the kernels are built with the command-line default)

using namespace uvsdev;

// the 512-thread instantiation of the persistent kernel (uvs_solve512.hip)
extern "C" {
int uvs_k_solve512_init(const unsigned char* fa, const unsigned char* fb, int n);
int uvs_k_solve512_launch(int n_windows, hipStream_t stream, char* blobs, const long long* blob_off, double* ws_all, const long long* ws_off,
                           const void* kopts, size_t kopts_bytes, uvs_report* reports, const void* dbg, size_t dbg_bytes);
size_t uvs_k_solve512_arg_bytes(int which);
int uvs_k_solve512_timeline(long long* out, size_t n);
int uvs_k_large_chunks512_prof(long long* out, size_t n);
int uvs_k_large_solve512_launch(hipStream_t stream, char* blob, double* ws, const void* kopts, size_t kopts_bytes, double* state, const double* reduced, int first, double radius, double* out,
                                 const double* ctl, int rank, int nranks, const double* fimg);
int uvs_k_large_chunks512_launch(int grid, hipStream_t stream, char* blob, double* ws, const void* kopts, size_t kopts_bytes, const double* state, int sel, int first, double radius,
                                  double* partials, const double* ctl, int rank, int nranks, int n_chunk_wgs, double* fimg);
}

// Worker threads of a handle for batch packing: created once, woken per batch (sixteen std::thread creations and joins per batch -- twice: packing, then the copy into
// the pinned staging buffer -- were a third of a millisecond of the 2.5 ms a 256-window batch spends on the host).
struct PackPool {
    std::vector<std::thread> th; std::mutex m; std::condition_variable cv_go, cv_done;
    const std::function<void(int)>* job = nullptr; int gen = 0, pending = 0; bool stop = false;
    // A worker joins at the generation that was current when it was created (`seen0`): a pool that grows after it has run must not hand the new thread the job of a
    // run() that has already returned (its std::function lived on that run()'s stack) nor let it decrement a `pending` it was never counted in.
    void worker(int t, int seen0) {
        int seen = seen0;
        // UVS_PACK_PIN=1: worker t stays on the (t + 1)-th CPU of the process's affinity mask (CPU 0 of the mask is left to the calling thread; on the EPYC hosts of the MI355X
        // boxes the SMT sibling of CPU i is i + 128, so the first 32 are distinct cores).  Off by default: on a shared host a pinned worker cannot move away from a core
        // another tenant is using (tools/stream_ab.py measures both; profiles/r06_stream_ab.txt)
        if (const char* e = std::getenv("UVS_PACK_PIN")) if (e[0] == '1') {
            cpu_set_t all; CPU_ZERO(&all);
            if (sched_getaffinity(0, sizeof(all), &all) == 0) {
                int want = t + 1, cpu = -1, count = CPU_COUNT(&all);
                if (count > 1) { want %= count; for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &all) && want-- == 0) { cpu = c; break; } }
                if (cpu >= 0) { cpu_set_t one; CPU_ZERO(&one); CPU_SET(cpu, &one); (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one); }
            }
        }
        for (;;) {
            const std::function<void(int)>* f;
            { std::unique_lock<std::mutex> lk(m); cv_go.wait(lk, [&] { return stop || gen != seen; }); if (stop) return; seen = gen; f = job; }
            if (f == nullptr) continue;      // (a generation whose run() is already over: nothing to do, nothing to count)
            (*f)(t);
            { std::lock_guard<std::mutex> lk(m); if (--pending == 0) cv_done.notify_one(); }
        }
    }
    bool ensure(int n) {      // false: thread creation failed (the caller packs on its own thread).  Called by the thread that calls run(), never beside a run() in flight.
        try {
            while ((int)th.size() < n) {
                const int t = (int)th.size(); int g0;
                { std::lock_guard<std::mutex> lk(m); g0 = gen; }
                th.emplace_back([this, t, g0] { worker(t, g0); });
            }
        } catch (...) { return false; }
        return true;
    }
    void run(int n, const std::function<void(int)>& f) {      // f(0 .. n-1) on n workers (n <= th.size()), the caller waits; workers beyond n see the generation and return at once
        const std::function<void(int)> g = [&](int t) { if (t < n) f(t); };
        { std::lock_guard<std::mutex> lk(m); job = &g; pending = (int)th.size(); ++gen; }
        cv_go.notify_all();
        std::unique_lock<std::mutex> lk(m); cv_done.wait(lk, [&] { return pending == 0; });
        job = nullptr;      // `g` dies with this frame
    }
    ~PackPool() { { std::lock_guard<std::mutex> lk(m); stop = true; } cv_go.notify_all(); for (auto& t : th) t.join(); }
};
// where pack_window may put a blob instead of the caller's vector: a bump allocator over the pinned staging buffer of the handle (batch packing: the windows of a batch go
// straight to where the one host -> device copy starts; off = -1 afterwards: no room, the blob is in the vector)
struct PackDst { std::atomic<size_t>* bump; char* base; size_t cap; long long off = -1; };
struct PackCache;
static void free_pack_cache(PackCache* c);
struct MargDevScratch;
static void free_marg_scratch(MargDevScratch* m);
struct MargBatchBuf;
static void free_marg_batch(MargBatchBuf* m);
struct MargWorker;
static void free_marg_worker(MargWorker* w);
static bool marg_in_flight(const uvs_solver* s);
struct uvs_solver {
    uvs_options opts;
    int device;
    int max_batch;
    int max_points = 0, max_point_obs = 0, max_lines = 0, max_line_obs = 0;      // per-window capacities promised at uvs_create
    uvs_solver* twin = nullptr;              // second buffer set of uvs_batch_stream (created on first use, destroyed with this handle)
    uvs_solver* twin2 = nullptr;             // ... and the third (in flight at once: a batch being packed, one being copied, one being solved)
    uvs_solver* twin3 = nullptr;             // ... and a fourth (UVS_STREAM_SETS=4: one more batch of slack for a host whose packing threads get descheduled)
    hipEvent_t ev_done = nullptr;            // recorded behind a set's k_solve in the stream: the next set's launch waits for it (the kernels of consecutive batches run one after the other)
    int n_cus = 256;                         // compute units of the device
    int large_solve_nt = 512;                // ... and for k_large_solve (UVS_LARGE_SOLVE_NT=256)
    int large_chunks_nt = 512;               // likewise for k_large_chunks (UVS_LARGE_CHUNKS_NT=256 selects the 256-thread kernel of this file)
    int ksolve_nt = 512;                     // which instantiation of the persistent kernel launch_solve uses (uvs_solve512.hip / this file's 256-thread one)
    int chunk_wgs() const { return std::max(1, n_cus - 1); }      // chunk workgroups of the persistent large-window kernels: one compute unit stays free for the frame-terms workgroup of the same launch
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    std::string err;
    // batch state
    int n_loaded = 0;
    std::vector<DevWin> hdrs;                // host copies of the per-window headers
    std::vector<long long> blob_off, ws_off;
    std::vector<char> host_blobs;
    MargDevScratch* marg_dev = nullptr;   // buffers of the device marginalization (sub-window blob, its workspace, the reduced system)
    struct MargBatchBuf* marg_batch = nullptr;      // ... and of uvs_marginalize_batch (allocated on first use)
    struct MargWorker* marg_worker = nullptr;      // uvs_marginalize_resident_begin(): the marginalization runs on this handle's worker thread (created on first use, kept: a thread per call was
    uvs_prior marg_job_out;                        // 30 - 60 us of every optimization() of a replay); its result waits in marg_job_out
    PackCache* pack_cache = nullptr;      // structure of the last large single window (allocated on first use)
    std::vector<std::vector<char>> slot_blobs;      // batch uploads: one packing buffer per batch slot, kept (with its pages) from batch to batch
    PackPool* pool = nullptr;                       // ... and the worker threads that fill them (created on the first threaded batch)
    bool pool_borrowed = false;                     // (a buffer set of uvs_batch_stream uses its owner's pool)
    // ONE host -> device copy per upload: [blobs | blob_off[n] | ws_off[n] | out_tab[3 n]] staged in pinned memory; the three tables
    // live behind the blobs in the same device allocation (d_blob_off / d_ws_off / d_out_tab point into it)
    char* d_blobs = nullptr; size_t d_blobs_cap = 0;
    char* h_up = nullptr; size_t h_up_cap = 0;            // pinned upload staging
    double* d_ws = nullptr; size_t d_ws_cap = 0;
    long long* d_blob_off = nullptr; long long* d_ws_off = nullptr;
    // ONE device -> host copy per download: per window {source offset in d_ws, doubles, destination offset} -> k_pack_outputs gathers the
    // final states AND the reports into one contiguous device buffer [states | reports[n]] -> pinned host buffer
    std::vector<long long> out_tab; long long* d_out_tab = nullptr; double* d_outpack = nullptr; size_t d_outpack_cap = 0; long long out_total = 0;
    char* h_out = nullptr; size_t h_out_cap = 0;          // pinned download staging
    uvs_report* d_reports = nullptr; size_t d_rep_cap = 0;
    double* d_dbg = nullptr;
    EvalScratch eval_scratch;                // uvs_evaluate / uvs_marginalize staging
    // large-window (configs[3]) run state
    struct Large {
        bool active = false; int n_chunks = 0, sel = 0, it = 0, invalid = 0, nsucc = 0, pending = 0, term = 0, status = 0;
        bool need_lin = true, first = true, done = false;
        double radius = 0, decr = 2, cost = 0, gmax = 0, x_norm = 0, local_x2 = 0;
        double *d_state = nullptr, *d_partials = nullptr, *d_reduced = nullptr, *d_bsums = nullptr, *d_out = nullptr, *d_sc5 = nullptr;
        size_t cap_partials = 0, cap_bsums = 0;
        uvs_report rep;
        double frame_x2 = 0;                                    // frame part of ||x||^2 (the landmark part is per rank: local_x2)
        double* d_ctl = nullptr; uvs_report* d_rep = nullptr;   // fused loop: trust-region state and report on the device
        void* comm = nullptr; int rank = 0, nranks = 1;         // RCCL communicator owned by the handle (uvs_large_comm_init)
        double relo_pose_in[7] = {0, 0, 0, 0, 0, 0, 0};      // passes through to uvs_large_finish (this path takes no relocalization blocks)
        int grid = 0;                                           // chunk workgroups of k_large_chunks / k_large_backsub = partial rows (min(n_chunks, compute units)); every launch adds ONE for the frame terms
        double* d_fimg = nullptr;                               // frame image of the reduced system (k_large_chunks' extra workgroup -> k_large_solve)
        std::chrono::steady_clock::time_point t_begin;          // start of the host-driven loop (options.max_solver_time_in_seconds)
    } L;
};

#define HIPCHK(s, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) { (s)->err = std::string(#call) + ": " + hipGetErrorString(e_); return UVS_ERR_HIP; } \
    } while (0)

static KOpts make_kopts(const uvs_options& o, int debug) {
    KOpts k;
    k.max_it = o.max_num_iterations; k.ex_free = o.estimate_extrinsic; k.keep_cand = o.function_tol_keeps_candidate; k.jacobi = o.jacobi_scaling;
    k.sqrt_info = o.point_sqrt_info; k.line_factor = o.line_factor; k.vp_factor = o.vp_factor;
    k.loss_pt = o.loss_point; k.loss_ln = o.loss_line; k.loss_vp = o.loss_vp;
    k.G[0] = o.gravity[0]; k.G[1] = o.gravity[1]; k.G[2] = o.gravity[2];
    k.r0 = o.initial_trust_region_radius; k.rmax = o.max_trust_region_radius; k.rmin = o.min_trust_region_radius;
    k.min_rel = o.min_relative_decrease; k.dlo = o.min_lm_diagonal; k.dhi = o.max_lm_diagonal;
    k.ftol = o.function_tolerance; k.gtol = o.gradient_tolerance; k.ptol = o.parameter_tolerance;
    k.max_ticks = o.max_solver_time_in_seconds > 0.0 ? std::max(1LL, (long long)(o.max_solver_time_in_seconds * 1e8)) : 0LL;      // wall_clock64(): 100 MHz
    k.max_invalid = o.max_consecutive_invalid_steps; k.debug = debug;
    { const char* e = std::getenv("UVS_REDAMP"); k.redamp = (e && e[0] == '0') ? 0 : 1; }      // diagnostic switch, read per launch (tests, A/B): 0 = re-linearize after every rejected step
    return k;
}

struct DevWin;
static int pack_window(const uvs_window* w_in, const uvs_options& opts, std::vector<char>& out, DevWin& hdr, std::string& err, int chunk_grid = 0, PackCache* cache = nullptr, PackDst* dst = nullptr);

extern "C" {

int uvs_abi_version(void) { return UVS_ABI_VERSION; }

void uvs_default_options(uvs_options* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->max_num_iterations = 10;          // euroc_config.yaml:56
    o->estimate_extrinsic = 0;           // :26
    o->estimate_td = 0;                  // :73
    o->function_tol_keeps_candidate = 0;
    o->focal_length = 461.6;             // :20
    o->point_sqrt_info = 461.6 / 1.6;    // estimator.cpp:17
    o->line_factor = 300.0; o->vp_factor = 10.0;   // :86-87
    o->loss_point = 1.0; o->loss_line = 0.1; o->loss_vp = 1.0;   // estimator.cpp:765-772
    o->gravity[0] = 0.0; o->gravity[1] = 0.0; o->gravity[2] = 9.81007;   // :64
    o->initial_trust_region_radius = 1e4; o->max_trust_region_radius = 1e16; o->min_trust_region_radius = 1e-32;
    o->min_relative_decrease = 1e-3; o->min_lm_diagonal = 1e-6; o->max_lm_diagonal = 1e32;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->max_consecutive_invalid_steps = 5; o->jacobi_scaling = 1;
    o->max_solver_time_in_seconds = 0.0;      // no wall-clock cap (the reference sets 0.1 s / 0.08 s, estimator.cpp:987-991: two orders of magnitude above a solve here)
}

const char* uvs_status_string(int st) {
    switch (st) {
        case UVS_OK: return "ok";
        case UVS_ERR_INVALID_ARG: return "invalid argument";
        case UVS_ERR_UNSUPPORTED: return "unsupported configuration";
        case UVS_ERR_NO_DEVICE: return "no HIP device (this library has no CPU path)";
        case UVS_ERR_HIP: return "HIP runtime error";
        case UVS_ERR_CAPACITY: return "capacity exceeded";
        case UVS_ERR_NUMERIC: return "numeric failure";
    }
    return "unknown";
}

const char* uvs_last_error(const uvs_solver* s) { return s ? s->err.c_str() : "null solver"; }

// host-only: the packing of `w` as uvs_batch_upload() would do it, nothing touches a device (CPU tests of the chunk / list layout, timing)
int uvs_debug_pack_layout(const uvs_options* o, const uvs_window* w, int32_t* info) {
    if (!o || !w || !info) return UVS_ERR_INVALID_ARG;
    std::vector<char> blob; DevWin h; std::string err;
    const char* grid_env = std::getenv("UVS_DEBUG_CHUNK_GRID");      // CPU tests of the large-window chunking (uvs_large_begin passes the device's CU count)
    const int rc = pack_window(w, *o, blob, h, err, grid_env ? std::atoi(grid_env) : 0, nullptr);
    if (rc != UVS_OK) return rc;
    const int32_t v[12] = {h.blob_bytes, h.ws_doubles, h.n_chunks, h.n_pt_obs, h.n_relo, h.pt_rec, h.pt_xslots, h.max_chunk_doubles, UVS_S_DOUBLES, h.n_parts, h.n_cimg, h.n_pblk};
    std::memcpy(info, v, sizeof(v));
    return UVS_OK;
}

int uvs_reduced_dim(const uvs_options* o) { return 15 * UVS_NUM_FRAMES + ((o && o->estimate_extrinsic) ? 6 : 0); }

int uvs_create(const uvs_options* opts, int device, int max_batch, int max_points, int max_point_obs, int max_lines,
               int max_line_obs, uvs_solver** out) {
    if (!opts || !out || max_batch < 1 || max_points < 0 || max_point_obs < 0 || max_lines < 0 || max_line_obs < 0) return UVS_ERR_INVALID_ARG;
    if (opts->max_num_iterations < 0) return UVS_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return UVS_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return UVS_ERR_NO_DEVICE;
    uvs_solver* s = new uvs_solver();
    s->opts = *opts; s->device = device; s->max_batch = max_batch;
    s->max_points = max_points; s->max_point_obs = max_point_obs; s->max_lines = max_lines; s->max_line_obs = max_line_obs;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) { delete s; return UVS_ERR_HIP; }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) s->n_cus = cus; }
    // block table for the output-stationary gather
    unsigned char fa[UVS_NBLK], fb[UVS_NBLK];
    for (int i = 0, b = 0; i < UVS_NF; ++i) for (int j = 0; j <= i; ++j, ++b) { fa[b] = (unsigned char)i; fb[b] = (unsigned char)j; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(c_blk_fa), fa, sizeof(fa)) != hipSuccess || hipMemcpyToSymbol(HIP_SYMBOL(c_blk_fb), fb, sizeof(fb)) != hipSuccess) { delete s; return UVS_ERR_HIP; }
    if (uvs_k_solve512_arg_bytes(0) != sizeof(KOpts) || uvs_k_solve512_arg_bytes(1) != sizeof(DebugOut) || uvs_k_solve512_init(fa, fb, UVS_NBLK) != UVS_OK) { delete s; return UVS_ERR_HIP; }
    { const char* e = std::getenv("UVS_KSOLVE_NT"); s->ksolve_nt = (e && std::atoi(e) == 256) ? 256 : 512; }
    { const char* e = std::getenv("UVS_LARGE_CHUNKS_NT"); s->large_chunks_nt = (e && std::atoi(e) == 256) ? 256 : 512; }
    { const char* e = std::getenv("UVS_LARGE_SOLVE_NT"); s->large_solve_nt = (e && std::atoi(e) == 256) ? 256 : 512; }      // A/B switch: 256 = the one-wave-per-SIMD instantiation of the persistent kernel
    // the LDS opt-in is a per-device function attribute: every handle sets it for its own device (the current one since hipSetDevice above)
    for (const void* fn : {(const void*)k_solve, (const void*)k_evaluate, (const void*)k_large_chunks, (const void*)k_large_solve, (const void*)k_large_backsub, (const void*)k_marg_linearize, (const void*)k_marg_linearize_batch})
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES) != hipSuccess) { delete s; return UVS_ERR_HIP; }
    if (hipFuncSetAttribute((const void*)uvsmarg::k_marg_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)uvsmarg::MF_LDS_BYTES) != hipSuccess) { delete s; return UVS_ERR_HIP; }
    *out = s;
    return UVS_OK;
}

void uvs_destroy(uvs_solver* s) {
    if (s) { free_marg_worker(s->marg_worker); s->marg_worker = nullptr; }      // (waits for a marginalization begun and never waited for: it still uses the handle)
    if (!s) return;
    if (s->twin) { uvs_destroy(s->twin); s->twin = nullptr; }
    if (s->twin2) { uvs_destroy(s->twin2); s->twin2 = nullptr; }
    if (s->twin3) { uvs_destroy(s->twin3); s->twin3 = nullptr; }
    if (s->ev_done) { (void)hipEventDestroy(s->ev_done); s->ev_done = nullptr; }
    free_pack_cache(s->pack_cache); s->pack_cache = nullptr;
    if (!s->pool_borrowed) delete s->pool;
    s->pool = nullptr;
    free_marg_scratch(s->marg_dev); s->marg_dev = nullptr;
    free_marg_batch(s->marg_batch); s->marg_batch = nullptr;
    (void)hipSetDevice(s->device);      // teardown: nothing useful to do with an error
    if (s->d_blobs) (void)hipFree(s->d_blobs);
    if (s->d_ws) (void)hipFree(s->d_ws);
    if (s->h_up) (void)hipHostFree(s->h_up);
    if (s->h_out) (void)hipHostFree(s->h_out);
    uvs_large_comm_destroy(s);
    if (s->L.d_ctl) (void)hipFree(s->L.d_ctl);
    if (s->L.d_fimg) (void)hipFree(s->L.d_fimg);
    if (s->L.d_rep) (void)hipFree(s->L.d_rep);
    if (s->d_outpack) (void)hipFree(s->d_outpack);
    if (s->d_reports) (void)hipFree(s->d_reports);
    if (s->d_dbg) (void)hipFree(s->d_dbg);
    s->eval_scratch.release();
    (void)hipEventDestroy(s->ev0); (void)hipEventDestroy(s->ev1);
    (void)hipStreamDestroy(s->stream);
    delete s;
}

}  // extern "C"

// ------------------------------------------------------------------ host packing: uvs_window -> blob
static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

static int validate_window(const uvs_window* w, std::string& err) {
    if (!w) { err = "null window"; return UVS_ERR_INVALID_ARG; }
    if (w->n_points < 0 || w->n_point_obs < 0 || w->n_lines < 0 || w->n_line_obs < 0 || w->n_imu < 0 || w->n_imu > UVS_WINDOW_SIZE) { err = "bad counts"; return UVS_ERR_INVALID_ARG; }
    if ((w->n_point_obs && (!w->pt_lm || !w->pt_fi || !w->pt_fj || !w->pt_pi || !w->pt_pj || !w->inv_depth)) ||
        (w->n_line_obs && (!w->ln_lm || !w->ln_fj || !w->ln_sp || !w->ln_ep || !w->ln_has_vp || !w->ln_vp || !w->line_orth)) || (w->n_imu && !w->imu)) { err = "null array"; return UVS_ERR_INVALID_ARG; }
    int prev = -1, prev_fj = -1, anchor = -1;
    for (int k = 0; k < w->n_point_obs; ++k) {
        const int lm = w->pt_lm[k], fi = w->pt_fi[k], fj = w->pt_fj[k];
        if (lm < 0 || lm >= w->n_points || lm < prev) { err = "point observations must be grouped by non-decreasing landmark index"; return UVS_ERR_INVALID_ARG; }
        if (fi < 0 || fj <= fi || fj >= UVS_NUM_FRAMES) { err = "point observation needs 0 <= imu_i < imu_j <= WINDOW_SIZE"; return UVS_ERR_INVALID_ARG; }
        if (lm != prev) { anchor = fi; prev_fj = -1; }
        if (fi != anchor || fj <= prev_fj) { err = "point observations of one landmark must share imu_i and have increasing imu_j"; return UVS_ERR_INVALID_ARG; }
        prev = lm; prev_fj = fj;
    }
    prev = -1; prev_fj = -1;
    for (int k = 0; k < w->n_line_obs; ++k) {
        const int lm = w->ln_lm[k], fj = w->ln_fj[k];
        if (lm < 0 || lm >= w->n_lines || lm < prev) { err = "line observations must be grouped by non-decreasing landmark index"; return UVS_ERR_INVALID_ARG; }
        if (fj < 0 || fj >= UVS_NUM_FRAMES) { err = "line observation frame out of range"; return UVS_ERR_INVALID_ARG; }
        if (lm != prev) prev_fj = -1;
        if (fj <= prev_fj) { err = "line observations of one landmark must have increasing imu_j"; return UVS_ERR_INVALID_ARG; }
        prev = lm; prev_fj = fj;
    }
    for (int b = 0; b < w->n_imu; ++b) if (w->imu[b].frame_i < 0 || w->imu[b].frame_i >= UVS_WINDOW_SIZE) { err = "imu frame_i out of range"; return UVS_ERR_INVALID_ARG; }
    if (w->prior && w->prior->n > 0) {
        const uvs_prior& p = *w->prior;
        if (p.n > UVS_MAX_PRIOR_DIM || p.n_blocks < 1 || p.n_blocks > UVS_MAX_PRIOR_BLOCKS) { err = "prior too large"; return UVS_ERR_CAPACITY; }
        if (!p.linearized_jacobians || !p.linearized_residuals || !p.x0) { err = "null array"; return UVS_ERR_INVALID_ARG; }
        for (int b = 0; b < p.n_blocks; ++b) {
            // kind <-> global size: pose / extrinsic 7, speed-bias 9, time offset 1; x0_off addresses x0[UVS_PRIOR_X0_LEN]
            const int kind = p.block_kind[b], want = kind == UVS_BLOCK_SPEEDBIAS ? 9 : kind == UVS_BLOCK_TD ? 1 : 7;
            if (kind < UVS_BLOCK_POSE || kind > UVS_BLOCK_TD || p.block_size[b] != want) { err = "prior block kind / size mismatch"; return UVS_ERR_INVALID_ARG; }
            if (p.x0_off[b] < 0 || p.x0_off[b] > UVS_PRIOR_X0_LEN - p.block_size[b]) { err = "prior x0 offset out of range"; return UVS_ERR_INVALID_ARG; }
            const int loc = p.block_size[b] == 7 ? 6 : p.block_size[b];
            if (p.block_idx[b] < 0 || p.block_idx[b] + loc > p.n) { err = "prior block index out of range"; return UVS_ERR_INVALID_ARG; }
            if ((p.block_kind[b] == UVS_BLOCK_POSE || p.block_kind[b] == UVS_BLOCK_SPEEDBIAS) && (p.block_frame[b] < 0 || p.block_frame[b] >= UVS_NUM_FRAMES)) { err = "prior frame out of range"; return UVS_ERR_INVALID_ARG; }
            // every kept block once, every prior column once: two blocks on the same parameter block would map two prior columns to one index of the reduced system, and the
            // device's (H0 entry, S offset) table -- generated with one slot per pair of S indices -- would be overrun (setup_window)
            for (int a = 0; a < b; ++a) {
                const int loca = p.block_size[a] == 7 ? 6 : p.block_size[a];
                const bool same_block = p.block_kind[a] == kind && (kind == UVS_BLOCK_EX_POSE || kind == UVS_BLOCK_TD || p.block_frame[a] == p.block_frame[b]);
                const bool overlap = p.block_idx[a] < p.block_idx[b] + loc && p.block_idx[b] < p.block_idx[a] + loca;
                if (same_block || overlap) { err = "prior keeps a parameter block twice / its blocks overlap"; return UVS_ERR_INVALID_ARG; }
            }
        }
    }
    return UVS_OK;
}

// appends the blob of `w` to `out` (8-byte aligned) and returns its header
// chunk_grid > 0 (large-window path): the landmark chunks are made SMALLER than the LDS staging area allows so that their number is a
// multiple of chunk_grid (the persistent workgroups of k_large_chunks / k_large_backsub then all carry the same number of chunks), or
// -- a shard with few landmarks -- so that every compute unit gets one
// [0, n) split into `nt` contiguous ranges, one host thread each (nt <= 1: the caller's thread).  Used INSIDE the packing of one large window
// (configs[3]: 510 chunks, 135 000 observations); batches of small windows are threaded across windows instead (upload_windows).
template <class F> static void pack_parallel(int n, int nt, F&& fn) {
    if (nt <= 1 || n < 2) { fn(0, n, 0); return; }
    nt = std::min(nt, n);
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back([&, t] { fn((int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt), t); });
    fn(0, (int)((long long)n / nt), 0);
    for (auto& th : pool) th.join();
}
static int pack_inner_threads(int n_obs) {
    if (n_obs < 20000) return 1;
    const char* env = std::getenv("UVS_PACK_THREADS");
    const int nt = env ? std::atoi(env) : (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
    return std::max(1, nt);
}

// The STRUCTURE of the last large window packed through a handle (index arrays, IMU links, prior block table, options): when the next window has the same
// structure -- the same landmarks observed from the same frames, only states and measurements moved on: repeated solves of one map, a benchmark loop --
// chunking, work split and gather lists (3/4 of the packing time) are reused and only the value sections of the blob are rewritten.  Compared exactly
// (memcmp of the arrays), no hashing.  Small windows do not use it: their structure changes with every frame of a live sequence.
struct PackCache {
    bool valid = false, device_holds_tables = false;
    int chunk_grid = 0, td_on = 0, ex_on = 0;
    int n_points = 0, n_pt_obs = 0, n_lines = 0, n_ln_obs = 0, n_imu = 0;
    std::vector<int32_t> pt_lm, pt_fi, pt_fj, ln_lm, ln_fj, ln_has_vp;
    int imu_fs[UVS_WINDOW_SIZE][2];
    bool have_prior = false; int prior_n = 0, prior_nb = 0; int prior_tab[5][UVS_MAX_PRIOR_BLOCKS];
    DevWin hdr;
    bool matches(const uvs_window* w, const uvs_options& o, int grid) const {
        if (!valid || !w || grid != chunk_grid || (o.estimate_td != 0) != (td_on != 0) || (o.estimate_extrinsic != 0) != (ex_on != 0) || w->n_relo_obs > 0) return false;
        if (w->n_points != n_points || w->n_point_obs != n_pt_obs || w->n_lines != n_lines || w->n_line_obs != n_ln_obs || w->n_imu != n_imu) return false;
        const bool hp = w->prior && w->prior->n > 0;
        if (hp != have_prior) return false;
        if (hp) {
            const uvs_prior& p = *w->prior;
            if (p.n != prior_n || p.n_blocks != prior_nb || p.n_blocks > UVS_MAX_PRIOR_BLOCKS) return false;
            for (int b = 0; b < p.n_blocks; ++b) if (p.block_kind[b] != prior_tab[0][b] || p.block_frame[b] != prior_tab[1][b] || p.block_size[b] != prior_tab[2][b] || p.block_idx[b] != prior_tab[3][b] || p.x0_off[b] != prior_tab[4][b]) return false;
            if (!p.linearized_jacobians || !p.linearized_residuals || !p.x0) return false;
        }
        if ((n_pt_obs && (!w->pt_lm || !w->pt_fi || !w->pt_fj || !w->pt_pi || !w->pt_pj || !w->inv_depth)) || (n_ln_obs && (!w->ln_lm || !w->ln_fj || !w->ln_sp || !w->ln_ep || !w->ln_has_vp || !w->ln_vp || !w->line_orth)) || (n_imu && !w->imu)) return false;
        if (td_on && n_pt_obs && (!w->pt_vel_i || !w->pt_vel_j || !w->pt_td_i || !w->pt_td_j)) return false;
        for (int b = 0; b < n_imu; ++b) if (w->imu[b].frame_i != imu_fs[b][0] || (w->imu[b].skip ? 1 : 0) != imu_fs[b][1]) return false;
        const size_t np_ = (size_t)n_pt_obs * 4, nl_ = (size_t)n_ln_obs * 4;
        return (!np_ || (!std::memcmp(w->pt_lm, pt_lm.data(), np_) && !std::memcmp(w->pt_fi, pt_fi.data(), np_) && !std::memcmp(w->pt_fj, pt_fj.data(), np_))) &&
               (!nl_ || (!std::memcmp(w->ln_lm, ln_lm.data(), nl_) && !std::memcmp(w->ln_fj, ln_fj.data(), nl_) && !std::memcmp(w->ln_has_vp, ln_has_vp.data(), nl_)));
    }
    void store(const uvs_window* w, const uvs_options& o, int grid, const DevWin& h) {
        valid = true; device_holds_tables = false; chunk_grid = grid; td_on = o.estimate_td != 0; ex_on = o.estimate_extrinsic != 0;
        n_points = w->n_points; n_pt_obs = w->n_point_obs; n_lines = w->n_lines; n_ln_obs = w->n_line_obs; n_imu = w->n_imu;
        pt_lm.assign(w->pt_lm, w->pt_lm + n_pt_obs); pt_fi.assign(w->pt_fi, w->pt_fi + n_pt_obs); pt_fj.assign(w->pt_fj, w->pt_fj + n_pt_obs);
        ln_lm.assign(w->ln_lm, w->ln_lm + n_ln_obs); ln_fj.assign(w->ln_fj, w->ln_fj + n_ln_obs); ln_has_vp.assign(w->ln_has_vp, w->ln_has_vp + n_ln_obs);
        for (int b = 0; b < n_imu; ++b) { imu_fs[b][0] = w->imu[b].frame_i; imu_fs[b][1] = w->imu[b].skip ? 1 : 0; }
        have_prior = w->prior && w->prior->n > 0;
        if (have_prior) { const uvs_prior& p = *w->prior; prior_n = p.n; prior_nb = p.n_blocks; for (int b = 0; b < p.n_blocks; ++b) { prior_tab[0][b] = p.block_kind[b]; prior_tab[1][b] = p.block_frame[b]; prior_tab[2][b] = p.block_size[b]; prior_tab[3][b] = p.block_idx[b]; prior_tab[4][b] = p.x0_off[b]; } }
        hdr = h;
    }
};
static void free_pack_cache(PackCache* c) { delete c; }
static constexpr int kPackCacheMinObs = 20000;      // windows at least this large use the structure cache (and the inner packing threads)

// the VALUE sections of a blob (everything that is not index bookkeeping): header, frame states, landmark parameters, measurements, IMU blocks, prior
static void fill_values(char* B, const DevWin& h, const uvs_window* w, bool td_on, int threads) {
    double* D = (double*)B;
    std::memcpy(B, &h, sizeof(h));
    std::memcpy(D + h.d_frames, w->pose, sizeof(double) * 77);
    std::memcpy(D + h.d_frames + 77, w->speedbias, sizeof(double) * 99);
    std::memcpy(D + h.d_frames + 176, w->ex_pose, sizeof(double) * 7);
    D[h.d_frames + 183] = w->td;
    std::memcpy(D + h.d_frames + 184, w->relo_pose, sizeof(double) * 7);
    for (int k = 0; k < h.n_points; ++k) D[h.d_invd + k] = w->inv_depth[k];
    pack_parallel(h.n_pt_obs, threads, [&](int k0_, int k1_, int) {
        for (int k = k0_; k < k1_; ++k) {
            for (int q = 0; q < 3; ++q) { D[h.d_ptmeas + q * h.pt_stride + k] = w->pt_pi[3 * k + q]; D[h.d_ptmeas + (3 + q) * h.pt_stride + k] = w->pt_pj[3 * k + q]; }
            if (td_on) {
                for (int q = 0; q < 2; ++q) { D[h.d_ptvel + q * h.pt_stride + k] = w->pt_vel_i[2 * k + q]; D[h.d_ptvel + (2 + q) * h.pt_stride + k] = w->pt_vel_j[2 * k + q]; }
                D[h.d_ptvel + 4 * h.pt_stride + k] = w->pt_td_i[k]; D[h.d_ptvel + 5 * h.pt_stride + k] = w->pt_td_j[k];
            }
        }
    });
    for (int k = 0; k < 4 * h.n_lines; ++k) D[h.d_line + k] = w->line_orth[k];
    pack_parallel(h.n_ln_obs, threads, [&](int k0_, int k1_, int) {
        for (int k = k0_; k < k1_; ++k)
            for (int q = 0; q < 3; ++q) {
                D[h.d_lnmeas + q * h.ln_stride + k] = w->ln_sp[3 * k + q]; D[h.d_lnmeas + (3 + q) * h.ln_stride + k] = w->ln_ep[3 * k + q];
                D[h.d_lnmeas + (6 + q) * h.ln_stride + k] = w->ln_vp[3 * k + q];
            }
    });
    for (int b = 0; b < h.n_imu; ++b) {
        const uvs_imu_block& ib = w->imu[b];
        double* blk = D + h.d_imu + (size_t)b * UVS_IMU_STRIDE;
        blk[0] = ib.sum_dt;
        for (int q = 0; q < 3; ++q) { blk[1 + q] = ib.delta_p[q]; blk[8 + q] = ib.delta_v[q]; blk[11 + q] = ib.linearized_ba[q]; blk[14 + q] = ib.linearized_bg[q]; }
        for (int q = 0; q < 4; ++q) blk[4 + q] = ib.delta_q[q];
        {   // only the five 3x3 blocks the factor reads, packed (UVS_IMU_JIDX)
            const int RC[5][2] = {{0, 9}, {0, 12}, {3, 12}, {6, 9}, {6, 12}};
            for (int q = 0; q < 5; ++q) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) blk[UVS_IMU_JAC + 9 * q + 3 * i + j] = ib.jacobian[(RC[q][0] + i) * 15 + RC[q][1] + j];
        }
        std::memcpy(blk + UVS_IMU_COV, ib.covariance, sizeof(double) * 225);
    }
    if (h.prior_n > 0) {
        const uvs_prior& p = *w->prior;
        const int n = p.n;
        std::memcpy(D + h.d_prior, p.linearized_jacobians, sizeof(double) * (size_t)n * n);      // row-major, read once per solve (setup_window builds J0^T J0, J0^T r0 from it)
        for (int r = 0; r < n; ++r) D[h.d_prior + n * n + r] = p.linearized_residuals[r];
        // linearization point of block b at stride 9 (not at x0_off[b]): the kernel's loads of it then do not depend on a table load
        for (int b = 0; b < p.n_blocks && b < 16; ++b) for (int k = 0; k < p.block_size[b] && k < 9; ++k) D[h.d_prior + n * n + 2 * n + 9 * b + k] = p.x0[p.x0_off[b] + k];
    }
}

static int pack_window(const uvs_window* w_in, const uvs_options& opts, std::vector<char>& out, DevWin& hdr, std::string& err, int chunk_grid, PackCache* cache, PackDst* dst) {
    if (cache && cache->matches(w_in, opts, chunk_grid) && out.size() == (size_t)cache->hdr.blob_bytes) {      // same structure as the blob still sitting in `out`: values only
        fill_values(out.data(), cache->hdr, w_in, opts.estimate_td != 0, pack_inner_threads(w_in->n_point_obs + w_in->n_line_obs));
        hdr = cache->hdr;
        return UVS_OK;
    }
    if (cache) { cache->valid = false; cache->device_holds_tables = false; out.clear(); }
    const bool prof_ = std::getenv("UVS_PACK_PROFILE") != nullptr;
    auto t_prev_ = std::chrono::steady_clock::now();
    auto lap_ = [&](const char* what) { if (prof_) { const auto n_ = std::chrono::steady_clock::now(); fprintf(stderr, "pack %-10s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n_ - t_prev_).count()); t_prev_ = n_; } };
    int rc = validate_window(w_in, err);
    if (rc != UVS_OK) return rc;
    lap_("validate");
    const bool td_on = opts.estimate_td != 0;
    // Relocalization blocks (estimator.cpp:944-978) become ordinary point observations whose second frame is the pseudo frame 12 = relo_Pose,
    // placed right after their landmark's last observation (the kernel wants a landmark's blocks together).  eidx maps a merged observation
    // back to the caller's index (-1 for a relocalization block): uvs_evaluate / uvs_marginalize keep the caller's numbering and skip them.
    const uvs_window* w = w_in;
    uvs_window wm;
    std::vector<int32_t> m_lm, m_fi, m_fj, eidx; std::vector<double> m_pi, m_pj, m_vi, m_vj, m_tdi, m_tdj;
    const int n_relo = w_in->n_relo_obs;
    if (n_relo < 0) { err = "bad counts"; return UVS_ERR_INVALID_ARG; }
    if (td_on && w_in->n_point_obs > 0 && (!w_in->pt_vel_i || !w_in->pt_vel_j || !w_in->pt_td_i || !w_in->pt_td_j)) { err = "estimate_td needs pt_vel_i / pt_vel_j / pt_td_i / pt_td_j"; return UVS_ERR_INVALID_ARG; }
    if (n_relo > 0) {
        // relo_Pose takes the six spare slots of the reduced system that a free extrinsic would take; the time offset has its own (index 175), so
        // ESTIMATE_TD and relocalization blocks coexist (estimator.cpp:784-797 + :944-978)
        // with a free extrinsic the spare slots are taken: relo_Pose becomes a second-level block (uvs_layout.h: UVS_RELO2_BLOCKROW) -- in the persistent kernel and, since
        // round 6, in the landmark-sharded forms of one rank (chunk_grid > 0; uvs_large_kernel.h: LG_R2)
        if (!w_in->relo_lm || !w_in->relo_pi || !w_in->relo_pj) { err = "null array"; return UVS_ERR_INVALID_ARG; }
        const int npo = w_in->n_point_obs;
        int q = 0;
        for (int k = 0; k < npo; ++k) {
            const int lm = w_in->pt_lm[k];
            m_lm.push_back(lm); m_fi.push_back(w_in->pt_fi[k]); m_fj.push_back(w_in->pt_fj[k]); eidx.push_back(k);
            for (int c = 0; c < 3; ++c) { m_pi.push_back(w_in->pt_pi[3 * k + c]); m_pj.push_back(w_in->pt_pj[3 * k + c]); }
            if (td_on) { for (int c = 0; c < 2; ++c) { m_vi.push_back(w_in->pt_vel_i[2 * k + c]); m_vj.push_back(w_in->pt_vel_j[2 * k + c]); } m_tdi.push_back(w_in->pt_td_i[k]); m_tdj.push_back(w_in->pt_td_j[k]); }
            if (k + 1 < npo && w_in->pt_lm[k + 1] == lm) continue;
            if (q < n_relo && w_in->relo_lm[q] < lm) { err = "relo_lm must be strictly increasing and name landmarks that have observations"; return UVS_ERR_INVALID_ARG; }
            if (q < n_relo && w_in->relo_lm[q] == lm) {
                m_lm.push_back(lm); m_fi.push_back(w_in->pt_fi[k]); m_fj.push_back(UVS_RELO_FRAME); eidx.push_back(-1);
                for (int c = 0; c < 3; ++c) { m_pi.push_back(w_in->relo_pi[3 * q + c]); m_pj.push_back(w_in->relo_pj[3 * q + c]); }
                // a relocalization block is the plain ProjectionFactor also under ESTIMATE_TD (estimator.cpp:967-970): zero image velocities
                // switch the time-offset terms of its record off (no shift of pts_i / pts_j, d r / d td = 0)
                if (td_on) { for (int c = 0; c < 2; ++c) { m_vi.push_back(0.0); m_vj.push_back(0.0); } m_tdi.push_back(w_in->td); m_tdj.push_back(w_in->td); }
                ++q;
            }
        }
        if (q != n_relo) { err = "relo_lm must be strictly increasing and name landmarks that have observations"; return UVS_ERR_INVALID_ARG; }
        wm = *w_in;
        wm.n_point_obs = (int)m_lm.size(); wm.pt_lm = m_lm.data(); wm.pt_fi = m_fi.data(); wm.pt_fj = m_fj.data(); wm.pt_pi = m_pi.data(); wm.pt_pj = m_pj.data();
        if (td_on) { wm.pt_vel_i = m_vi.data(); wm.pt_vel_j = m_vj.data(); wm.pt_td_i = m_tdi.data(); wm.pt_td_j = m_tdj.data(); }
        w = &wm;
    }
    DevWin h; std::memset(&h, 0, sizeof(h));
    h.n_points = w->n_points; h.n_pt_obs = w->n_point_obs; h.n_lines = w->n_lines; h.n_ln_obs = w->n_line_obs; h.n_imu = w->n_imu;
    const bool have_prior = w->prior && w->prior->n > 0;
    h.prior_n = have_prior ? w->prior->n : 0; h.prior_nb = have_prior ? w->prior->n_blocks : 0;
    h.pt_stride = rup(std::max(h.n_pt_obs, 1), 8); h.ln_stride = rup(std::max(h.n_ln_obs, 1), 8);
    const bool ex_on = opts.estimate_extrinsic != 0;
    h.td_on = td_on ? 1 : 0; h.ex_on = ex_on ? 1 : 0;
    const bool relo_on = n_relo > 0;
    h.relo_on = relo_on ? 1 : 0; h.n_relo = n_relo;
    const bool relo2 = relo_on && opts.estimate_extrinsic != 0;      // relo_Pose as a second-level block (block row 13 of the gather)
    h.relo2 = relo2 ? 1 : 0;
    h.pt_rec = ex_on ? UVS_PT_REC_EX : td_on ? UVS_PT_REC_TD : UVS_PT_REC; h.pt_xslots = 1 + (td_on ? 1 : 0) + (ex_on ? 1 : 0);
    const int PREC = h.pt_rec, XS = h.pt_xslots;
    const long stage_cap = (long)UVS_S_DOUBLES;
    constexpr int NGMAX = UVS_NGRP, NG = UVS_NGRP, GPW = GRP_PER_WAVE;      // gather groups (two lanes each)
    // CSR by landmark
    std::vector<int> pbeg(h.n_points + 1, 0), lbeg(h.n_lines + 1, 0);
    for (int k = 0; k < h.n_pt_obs; ++k) pbeg[w->pt_lm[k] + 1]++;
    for (int k = 0; k < h.n_points; ++k) pbeg[k + 1] += pbeg[k];
    for (int k = 0; k < h.n_ln_obs; ++k) lbeg[w->ln_lm[k] + 1]++;
    for (int k = 0; k < h.n_lines; ++k) lbeg[k + 1] += lbeg[k];
    // chunks: greedy packing of whole landmarks into the LDS staging area (UVS_S_DOUBLES doubles).  A chunk holds the
    // observation records, the per-landmark Schur factors AND its gather lists (ints, 2 per double).
    std::vector<int> chunks;     // UVS_CHUNK_INTS ints per chunk (uvs_layout.h: i_chunks)
    {
        const long list_hdr = 2 * (NG + 1);
        // LDS doubles a chunk of landmarks [k0, k1) needs (records + Schur factors + gather lists), -1 if an index field overflows
        auto need_pt = [&](int k0, int k1) -> long {
            long nob = pbeg[k1] - pbeg[k0], nlm = k1 - k0, nli = list_hdr;
            // Schur entries: all slot pairs of the landmark; direct entries per observation: 3, + 3 with td, + 3 with ex (+ 1 more with both: (ex, td))
            const long dper = 3 + (td_on ? 3 : 0) + (ex_on ? 3 + (td_on ? 1 : 0) : 0);
            for (int k = k0; k < k1; ++k) { const long no = pbeg[k + 1] - pbeg[k]; nli += no ? (no + XS) * (no + XS + 1) / 2 + dper * no : 0; }
            if (nlm > 1023 || nob + XS * nlm > 16383) return -1;
            return (long)PREC * nob + 12 * (nob + XS * nlm) + (nli + 1) / 2;
        };
        auto need_ln = [&](int k0, int k1) -> long {
            long nob = lbeg[k1] - lbeg[k0], nlm = k1 - k0, nli = list_hdr;
            for (int k = k0; k < k1; ++k) { const long no = lbeg[k + 1] - lbeg[k]; nli += no * (no + 1) / 2 + no; }
            if (nlm > 1023 || nob > 16383) return -1;
            return (long)(UVS_LN_REC + 2 * UVS_LN_EY) * nob + 20 * nlm + (nli + 1) / 2;
        };
        // smallest number of chunks whose EVEN split (by observation count) fits; the kernel pays a fixed cost per chunk, so a
        // greedy fill that leaves a nearly empty last chunk would waste a whole pass
        // even split (by observation count) of a landmark family into n chunks; empty vector if some chunk does not fit the staging area
        auto cuts_for = [&](int n, int n_lm, const std::vector<int>& beg, auto&& need) -> std::vector<int> {
            std::vector<int> cut(1, 0);
            const long tot = beg[n_lm];
            for (int j = 1; j < n; ++j) {
                int k = cut.back() + 1;
                while (k < n_lm && (long)beg[k] * n < tot * j) ++k;
                cut.push_back(std::min(k, n_lm - (n - j)));
            }
            cut.push_back(n_lm);
            for (int j = 0; j < n; ++j) { const long nd = need(cut[j], cut[j + 1]); if (!(cut[j + 1] > cut[j] && nd >= 0 && nd <= stage_cap)) return {}; }
            return cut;
        };
        // smallest number of chunks >= n_from whose even split fits; the kernel pays a fixed cost per chunk, so a greedy fill that leaves a
        // nearly empty last chunk would waste a whole pass
        auto split = [&](int type, int n_lm, const std::vector<int>& beg, auto&& need, int n_from, std::vector<int>& cut) -> int {
            cut.clear();
            if (n_lm == 0) return UVS_OK;
            // start at the capacity lower bound (records + Schur factors alone; the lists come on top): walking n = 1, 2, ... costs
            // O(n * landmarks) per attempt, milliseconds for the 340 chunks of configs[3]
            const long mine = type == 0 ? (long)PREC * h.n_pt_obs + 12L * (h.n_pt_obs + XS * h.n_points) : (long)(UVS_LN_REC + 2 * UVS_LN_EY) * h.n_ln_obs + 20L * h.n_lines;
            const int n_first = (int)std::min<long>(n_lm, std::max<long>(std::max(1, n_from), mine / stage_cap));
            for (int n = n_first; n <= n_lm; ++n) { cut = cuts_for(n, n_lm, beg, need); if (!cut.empty()) return UVS_OK; }
            return UVS_ERR_CAPACITY;
        };
        std::vector<int> cut_pt, cut_ln;
        if (split(0, h.n_points, pbeg, need_pt, 1, cut_pt) != UVS_OK) { err = "single point landmark exceeds LDS staging"; return UVS_ERR_CAPACITY; }
        if (split(1, h.n_lines, lbeg, need_ln, 1, cut_ln) != UVS_OK) { err = "single line landmark exceeds LDS staging"; return UVS_ERR_CAPACITY; }
        if (chunk_grid > 0) {
            const int n_pt = cut_pt.empty() ? 0 : (int)cut_pt.size() - 1, n_ln = cut_ln.empty() ? 0 : (int)cut_ln.size() - 1, n_min = n_pt + n_ln;
            // work per family ~ its staging volume; a chunk should keep at least ~64 observations (its fixed cost is a few microseconds)
            const double w_pt = (double)PREC * h.n_pt_obs + 12.0 * (h.n_pt_obs + XS * h.n_points), w_ln = (double)(UVS_LN_REC + 2 * UVS_LN_EY) * h.n_ln_obs + 20.0 * h.n_lines;
            static const long min_obs = std::getenv("UVS_CHUNK_MIN_OBS") ? std::max(1, std::atoi(std::getenv("UVS_CHUNK_MIN_OBS"))) : 64;
            const long by_size = (long)(h.n_pt_obs + h.n_ln_obs) / min_obs;
            long target = n_min >= chunk_grid ? (long)((n_min + chunk_grid - 1) / chunk_grid) * chunk_grid : std::min<long>(chunk_grid, std::max<long>(n_min, by_size));
            if (target > n_min && w_pt + w_ln > 0.0) {
                int t_pt = (int)std::lround(target * w_pt / (w_pt + w_ln));
                t_pt = std::max(n_pt, std::min(t_pt, (int)target - n_ln));
                int t_ln = (int)target - t_pt;
                t_pt = std::min(t_pt, h.n_points); t_ln = std::min(t_ln, h.n_lines);
                std::vector<int> c2;
                if (t_pt > n_pt && split(0, h.n_points, pbeg, need_pt, t_pt, c2) == UVS_OK) cut_pt = c2;
                if (t_ln > n_ln && split(1, h.n_lines, lbeg, need_ln, t_ln, c2) == UVS_OK) cut_ln = c2;
            }
        }
        for (size_t j = 0; j + 1 < cut_pt.size(); ++j) chunks.insert(chunks.end(), {0, cut_pt[j], cut_pt[j + 1], 0, 0, 0, pbeg[cut_pt[j]], pbeg[cut_pt[j + 1]] - pbeg[cut_pt[j]]});
        for (size_t j = 0; j + 1 < cut_ln.size(); ++j) chunks.insert(chunks.end(), {1, cut_ln[j], cut_ln[j + 1], 0, 0, 0, lbeg[cut_ln[j]], lbeg[cut_ln[j + 1]] - lbeg[cut_ln[j]]});
    }
    lap_("split");
    // gather lists per chunk and per lower 6x6 pose block (see uvs_solve_kernel.h: gather_points / gather_lines),
    // pre-expanded into LDS offsets (doubles from the staging base; the chunk layout below mirrors lin_chunk()):
    //   points: rec[nob][30] | E[(nob+nlm)][6] | EI[(nob+nlm)][6] | lists      lines: rec[nob][34] | E[nob][24] | Y[nob][24] | X[nlm][20] | lists
    //   Schur entry : offset(E row of frame a) | offset(EI / Y row of frame b) << 16
    //   direct entry: points: offset(first Jacobian block) | offset(second) << 16   (A^T A, B^T B, B^T A) ; lines: record offset
    const int inner_threads = pack_inner_threads(h.n_pt_obs + h.n_ln_obs);
    std::vector<int> lists;
    int wblk[NGMAX];
    for (int g = 0; g < NGMAX; ++g) wblk[g] = -1;
    h.n_parts = 1;
    {
    const int n_ch = (int)chunks.size() / UVS_CHUNK_INTS;
    // Two passes over the same generator: the first only COUNTS the entries per pose block (what the work split below needs), the second
    // regenerates them chunk by chunk into one reused set of vectors while the lists are written.  (Keeping every chunk's entries
    // alive between the passes cost 80 k small vectors on a configs[3]-sized window: two thirds of the packing time.)
    std::vector<long> blk_work(UVS_NBLKX2, 0), blk_s(UVS_NBLKX2, 0), blk_d(UVS_NBLKX2, 0), blk_wp(UVS_NBLKX2, 0), blk_wl(UVS_NBLKX2, 0);
    auto blk_of = [](int fa, int fb) { return fa * (fa + 1) / 2 + fb; };   // fa >= fb ; fa == 11 is the time-offset pseudo frame: 66 + fb
    auto chunk_entries = [&](int qc, auto&& addS, auto&& addD) {
        const int type = chunks[UVS_CHUNK_INTS * qc], k0 = chunks[UVS_CHUNK_INTS * qc + 1], k1 = chunks[UVS_CHUNK_INTS * qc + 2];
        if (type == 0) {
            const int o0 = pbeg[k0], nob = pbeg[k1] - o0, nlm = k1 - k0;
            const int oE = nob * PREC, oEI = oE + 6 * (nob + XS * nlm);
            for (int k = k0; k < k1; ++k) {
                const int li = k - k0, b0 = pbeg[k] - o0, b1 = pbeg[k + 1] - o0;
                if (b1 == b0) continue;
                const int first_slot = b0 + XS * li;
                int fr[UVS_NUM_FRAMES + 4], nf = 0;      // block rows of the landmark's Schur slots
                fr[nf++] = w->pt_fi[o0 + b0];
                for (int o = b0; o < b1; ++o) fr[nf++] = (relo2 && w->pt_fj[o0 + o] == UVS_RELO_FRAME) ? UVS_RELO2_BLOCKROW : w->pt_fj[o0 + o];
                if (td_on) fr[nf++] = UVS_NUM_FRAMES;                                  // then the td slot of this landmark (pseudo frame 11)
                if (ex_on) fr[nf++] = UVS_NUM_FRAMES + 1;                              // then its extrinsic slot (pseudo frame 12)
                for (int sa = 0; sa < nf; ++sa) for (int sb = 0; sb <= sa; ++sb) {    // frames increase with the slot, except a relocalization block (pseudo frame 12) ahead of the td slot (11)
                    const bool up = fr[sa] >= fr[sb];
                    const int ra = up ? sa : sb, rb = up ? sb : sa;
                    addS(blk_of(fr[ra], fr[rb]), (oE + 6 * (first_slot + ra)) | ((oEI + 6 * (first_slot + rb)) << 16));
                }
                for (int o = b0; o < b1; ++o) {
                    const bool is_relo = w->pt_fj[o0 + o] == UVS_RELO_FRAME;
                    const int fi = w->pt_fi[o0 + o], fj = (relo2 && is_relo) ? UVS_RELO2_BLOCKROW : w->pt_fj[o0 + o], ro = o * PREC;
                    addD(blk_of(fi, fi), (ro + UVS_PT_A) | UVS_PT_ENTRY_A | ((ro + UVS_PT_A) << 16));      // (flag: this entry's corrected residual sits 26, not 12, doubles behind its first operand)
                    addD(blk_of(fj, fj), (ro + UVS_PT_B) | ((ro + UVS_PT_B) << 16));
                    addD(blk_of(fj, fi), (ro + UVS_PT_B) | ((ro + UVS_PT_A) << 16));
                    if (td_on && !is_relo) {                                           // J_td^T [A | B | J_td]  (a relocalization block does not depend on td)
                        addD(blk_of(UVS_NUM_FRAMES, fi), (ro + UVS_PT_TD) | ((ro + UVS_PT_A) << 16));
                        addD(blk_of(UVS_NUM_FRAMES, fj), (ro + UVS_PT_TD) | ((ro + UVS_PT_B) << 16));
                        addD(blk_of(UVS_NUM_FRAMES, UVS_NUM_FRAMES), (ro + UVS_PT_TD) | ((ro + UVS_PT_TD) << 16));
                    }
                    if (ex_on) {                                                       // J_ex^T [A | B | J_td | J_ex]
                        const int X = UVS_NUM_FRAMES + 1;
                        addD(blk_of(X, fi), (ro + UVS_PT_EX) | ((ro + UVS_PT_A) << 16));
                        if (fj > X) addD(blk_of(fj, X), (ro + UVS_PT_B) | ((ro + UVS_PT_EX) << 16));      // (relo_Pose, ex): the rows are relo_Pose's
                        else addD(blk_of(X, fj), (ro + UVS_PT_EX) | ((ro + UVS_PT_B) << 16));
                        if (td_on && !is_relo) addD(blk_of(X, UVS_NUM_FRAMES), (ro + UVS_PT_EX) | ((ro + UVS_PT_TD) << 16));
                        addD(blk_of(X, X), (ro + UVS_PT_EX) | ((ro + UVS_PT_EX) << 16));
                    }
                }
            }
        } else {
            const int o0 = lbeg[k0], nob = lbeg[k1] - o0;
            const int oE = nob * UVS_LN_REC, oY = oE + UVS_LN_EY * nob;
            for (int k = k0; k < k1; ++k) {
                const int b0 = lbeg[k] - o0, b1 = lbeg[k + 1] - o0;
                for (int sa = 0; sa < b1 - b0; ++sa) for (int sb = 0; sb <= sa; ++sb)
                    addS(blk_of(w->ln_fj[o0 + b0 + sa], w->ln_fj[o0 + b0 + sb]), (oE + UVS_LN_EY * (b0 + sa)) | ((oY + UVS_LN_EY * (b0 + sb)) << 16));
                for (int o = b0; o < b1; ++o) addD(blk_of(w->ln_fj[o0 + o], w->ln_fj[o0 + o]), o * UVS_LN_REC);
            }
        }
    };
    std::vector<int> cnt_s((size_t)n_ch * UVS_NBLKX2), cnt_d((size_t)n_ch * UVS_NBLKX2);      // entries per chunk and pose block (first pass), reused when the lists are written
    {
        struct Cnt { long s[UVS_NBLKX2], d[UVS_NBLKX2], wp[UVS_NBLKX2], wl[UVS_NBLKX2]; };
        std::vector<Cnt> part((size_t)std::max(inner_threads, 1));
        for (auto& c : part) std::memset(&c, 0, sizeof(c));
        pack_parallel(n_ch, inner_threads, [&](int q0, int q1, int t) {
            Cnt& c = part[t];
            for (int qc = q0; qc < q1; ++qc) {
                long cs[UVS_NBLKX2] = {0}, cd[UVS_NBLKX2] = {0};
                chunk_entries(qc, [&](int b, int) { ++cs[b]; }, [&](int b, int) { ++cd[b]; });
                for (int b = 0; b < UVS_NBLKX2; ++b) { cnt_s[(size_t)qc * UVS_NBLKX2 + b] = (int)cs[b]; cnt_d[(size_t)qc * UVS_NBLKX2 + b] = (int)cd[b]; }
                const int type = chunks[UVS_CHUNK_INTS * qc];
                // work units ~ cycles per entry of the rows-per-lane gather
                for (int b = 0; b < UVS_NBLKX2; ++b) {
                    // measured per entry on MI355X (per-wave timers, UVS_DEBUG_GATHER_TIMERS): point Schur 350 cycles, point direct 675 cycles
                    const long ws_ = (type == 0 ? 18 : 72) * cs[b], wd_ = (type == 0 ? 35 : 63) * cd[b];
                    c.s[b] += ws_; c.d[b] += wd_;
                    (type == 0 ? c.wp : c.wl)[b] += ws_ + wd_;      // per landmark family: the chunks of a family are separated by barriers
                }
            }
        });
        for (const auto& c : part) for (int b = 0; b < UVS_NBLKX2; ++b) { blk_s[b] += c.s[b]; blk_d[b] += c.d[b]; blk_wp[b] += c.wp[b]; blk_wl[b] += c.wl[b]; }
    }
    lap_("entries");
    // gather groups: 256 two-lane groups, at least one per pose block; the spare ones split the heaviest blocks further.  Groups are dealt to
    // the waves heaviest first (similar list lengths inside a wave => little divergence); the wave order pairs heavy with light
    // waves on a SIMD (waves w and w+4 share one).
    int g_blk[NGMAX], g_part[NGMAX], g_np[NGMAX];
    {
        for (int b = 0; b < UVS_NBLKX2; ++b) blk_work[b] = blk_s[b] + blk_d[b];
        struct Item { int b, part, np; long work; double shape; };
        // water-filling: hand the spare groups, one at a time, to the block whose per-group share is largest (at most 16 parts)
        int np[UVS_NBLKX2]; int used = 0;
        for (int b = 0; b < UVS_NBLKX2; ++b) {      // the pseudo-frame blocks only exist with their option
            const bool tdb = b >= UVS_NBLK && b < UVS_NBLK + UVS_NF + 1, exb = b >= UVS_NBLK + UVS_NF + 1 && b < UVS_NBLKX;
            // a block nothing contributes to (frames further apart than the longest track, pseudo-frame blocks of an option that is off) gets
            // no group at all: S is zeroed anyway, and its group goes to a heavy block instead (15 of 128 groups for the canonical window)
            const bool r2b = b >= UVS_NBLKX;      // block row 13 (relo_Pose beside a free extrinsic)
            np[b] = ((b < UVS_NBLK || (tdb && td_on) || (!r2b && exb && (ex_on || relo_on) && (td_on || b != UVS_NBLK + UVS_NF + 1 + UVS_NF)) || (r2b && relo2 && (td_on || b != UVS_NBLKX + UVS_NF))) && blk_work[b] > 0) ? 1 : 0;
            used += np[b];
        }
        // The waves run in lock step inside a chunk and the chunks of the two landmark families are separated by barriers, so what counts
        // is the LARGEST per-group share within each family, not the per-group total: a block that is heavy in the point chunks only (the
        // diagonal blocks: all the J^T J terms) must be split until its point share matches the others', even if its total looks average.
        // Greedy: the next spare group goes to the family whose current maximum weighs more, and there to the block that holds it.
        int act[UVS_NBLKX2], na = 0;      // the blocks that take part (ascending: the scans below keep the tie-breaking order of a scan over all blocks)
        for (int b = 0; b < UVS_NBLKX2; ++b) if (np[b] > 0) act[na++] = b;
        while (used < NG) {
            int bp = -1, bl = -1;
            double mp = 0.0, ml = 0.0;      // the true maxima include the blocks that cannot be split any further
            for (int q = 0; q < na; ++q) {
                const int b = act[q];
                mp = std::max(mp, (double)blk_wp[b] / np[b]); ml = std::max(ml, (double)blk_wl[b] / np[b]);
                if (np[b] >= 16) continue;
                if (blk_wp[b] > 0 && (bp < 0 || blk_wp[b] * np[bp] > blk_wp[bp] * np[b])) bp = b;
                if (blk_wl[b] > 0 && (bl < 0 || blk_wl[b] * np[bl] > blk_wl[bl] * np[b])) bl = b;
            }
            int best = -1;
            const double sp_ = bp >= 0 ? (double)blk_wp[bp] / np[bp] : -1.0, sl_ = bl >= 0 ? (double)blk_wl[bl] / np[bl] : -1.0;
            if (bp >= 0 && sp_ >= mp && (mp >= ml || bl < 0 || sl_ < ml)) best = bp;
            else if (bl >= 0 && sl_ >= ml) best = bl;
            else if (bp >= 0 && (bl < 0 || sp_ >= sl_)) best = bp;
            else best = bl;
            if (best < 0) break;
            ++np[best]; ++used;
        }
        std::vector<Item> items;
        for (int b = 0; b < UVS_NBLKX2; ++b) for (int q = 0; q < np[b]; ++q) items.push_back({b, q, np[b], blk_work[b] / np[b], blk_work[b] ? (double)blk_d[b] / (double)blk_work[b] : -1.0});
        // a wave runs max(Schur count) + max(direct count) iterations over its 32 groups: deal groups of similar SHAPE (share of
        // direct work) to the same wave, idle groups last
        std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b2) { return a.shape != b2.shape ? a.shape > b2.shape : a.work > b2.work; });
        int wave_of_rank[NW];
        // (identity: the parts of a split block must sit in CONSECUTIVE groups -- gacc_gather_parts addresses part p at lane + p * UVS_GLANES -- and a block may
        // straddle two ranks; the round-2 order for 8 waves, heavy ranks paired with light ones on a SIMD, broke exactly that: the wrong pose blocks of the 512-thread builds)
        for (int r = 0; r < NW; ++r) wave_of_rank[r] = r;
        h.n_parts = 1;
        for (int b = 0; b < UVS_NBLKX2; ++b) h.n_parts = std::max(h.n_parts, np[b]);
        for (int g = 0; g < NG; ++g) { wblk[g] = -1; g_blk[g] = -1; g_part[g] = 0; g_np[g] = 1; }
        for (size_t q = 0; q < items.size(); ++q) {
            const int g = wave_of_rank[q / GPW] * GPW + (int)(q % GPW);
            const int b = items[q].b;
            const int bfa = b >= UVS_NBLKX ? UVS_RELO2_BLOCKROW : b >= UVS_NBLK + UVS_NF + 1 ? UVS_NUM_FRAMES + 1 : b >= UVS_NBLK ? UVS_NUM_FRAMES : (int)((std::sqrt(8.0 * b + 1.0) - 1.0) * 0.5 + 1e-9), bfb = b - bfa * (bfa + 1) / 2;     // b = fa(fa+1)/2 + fb
            wblk[g] = b | (bfa == bfb ? 256 : 0) | (items[q].part << 9) | (bfa << 13) | (bfb << 17) | ((items[q].np - 1) << 21);      // parts of a block sit in consecutive groups
            g_blk[g] = b; g_part[g] = items[q].part; g_np[g] = items[q].np;
        }
    }
    lap_("groups");
    {
        // every thread builds the lists of a contiguous range of chunks into its own vector (offsets relative to it); the ranges are concatenated
        // in chunk order afterwards, so the result does not depend on the thread count
        struct Part { std::vector<int> lists; int max_used = 0; bool overflow = false; int q0 = 0, q1 = 0; };
        std::vector<Part> part((size_t)std::max(inner_threads, 1));
        const bool dbg_lists = std::getenv("UVS_DEBUG_LISTS") != nullptr;
        pack_parallel(n_ch, dbg_lists ? 1 : inner_threads, [&](int q0, int q1, int t) {
            Part& P = part[t]; P.q0 = q0; P.q1 = q1;
            // A group's list is a contiguous slice [n p / np, n (p + 1) / np) of its block's entries in generation order; with the counts of the first pass the
            // destination of every entry is known before it is generated, so the entries go straight to their place (no per-block vectors: they were half of
            // the packing time of a canonical window).
            int first_grp[UVS_NBLKX2], blk_np[UVS_NBLKX2];
            for (int b = 0; b < UVS_NBLKX2; ++b) { first_grp[b] = -1; blk_np[b] = 0; }
            for (int g = 0; g < NG; ++g) if (g_blk[g] >= 0) { blk_np[g_blk[g]] = g_np[g]; if (g_part[g] == 0) first_grp[g_blk[g]] = g; }
            int dst[2][NGMAX], lo_of[2][NGMAX], fill[2][UVS_NBLKX2], cur_part[2][UVS_NBLKX2], cur_hi[2][UVS_NBLKX2];
            for (int qc = q0; qc < q1; ++qc) {
                const int* cS = cnt_s.data() + (size_t)qc * UVS_NBLKX2; const int* cD = cnt_d.data() + (size_t)qc * UVS_NBLKX2;
                chunks[UVS_CHUNK_INTS * qc + 3] = (int)P.lists.size();      // relative to this part for now
                const size_t base = P.lists.size();
                int at = 0;
                for (int pass = 0; pass < 2; ++pass) {
                    const int* cnt = pass == 0 ? cS : cD;
                    for (int g = 0; g < NG; ++g) {
                        dst[pass][g] = at; lo_of[pass][g] = 0;
                        if (g_blk[g] < 0) continue;
                        const long n = cnt[g_blk[g]], lo = n * g_part[g] / g_np[g], hi = n * (g_part[g] + 1) / g_np[g];
                        lo_of[pass][g] = (int)lo; at += (int)(hi - lo);
                    }
                }
                const int n_ent = at;
                P.lists.resize(base + 2 * (NG + 1) + n_ent);
                int* hdrp = P.lists.data() + base; int* ent = hdrp + 2 * (NG + 1);
                for (int pass = 0; pass < 2; ++pass) {
                    for (int g = 0; g < NG; ++g) hdrp[pass * (NG + 1) + g] = dst[pass][g];
                    hdrp[pass * (NG + 1) + NG] = pass == 0 ? dst[1][0] : n_ent;
                }
                // (the entries of a block arrive in order, so its current part and that part's end are carried along: no division per entry)
                for (int b = 0; b < UVS_NBLKX2; ++b) for (int pass = 0; pass < 2; ++pass) {
                    fill[pass][b] = 0; cur_part[pass][b] = 0;
                    const long n = (pass == 0 ? cS : cD)[b];
                    cur_hi[pass][b] = blk_np[b] > 0 ? (int)(n / blk_np[b]) : 0;
                    while (blk_np[b] > 0 && cur_part[pass][b] + 1 < blk_np[b] && cur_hi[pass][b] == 0) { ++cur_part[pass][b]; cur_hi[pass][b] = (int)(n * (cur_part[pass][b] + 1) / blk_np[b]); }      // leading empty parts
                }
                const auto put = [&](int pass, int b, int v) {
                    const int np_ = blk_np[b];
                    if (np_ <= 0) return;      // (a block without a group has no work by construction)
                    const int e = fill[pass][b]++;
                    while (e >= cur_hi[pass][b] && cur_part[pass][b] + 1 < np_) { ++cur_part[pass][b]; cur_hi[pass][b] = (int)((long)(pass == 0 ? cS : cD)[b] * (cur_part[pass][b] + 1) / np_); }
                    const int g = first_grp[b] + cur_part[pass][b];
                    ent[dst[pass][g] + (e - lo_of[pass][g])] = v;
                };
                chunk_entries(qc, [&](int b, int v) { put(0, b, v); }, [&](int b, int v) { put(1, b, v); });
                chunks[UVS_CHUNK_INTS * qc + 4] = (int)(P.lists.size() - base);
                {   // the chunk as the kernel lays it out must fit the staging area: records + Schur factors + the lists just built (an estimate that
                    // is too small would let the lists run over the LM state that follows S in LDS)
                    const int type = chunks[UVS_CHUNK_INTS * qc], k0 = chunks[UVS_CHUNK_INTS * qc + 1], k1 = chunks[UVS_CHUNK_INTS * qc + 2];
                    const long nlist = (long)(P.lists.size() - base);
                    long used;
                    if (type == 0) { const long nob = pbeg[k1] - pbeg[k0], nlm = k1 - k0; used = (long)PREC * nob + 12 * (nob + XS * nlm) + (nlist + 1) / 2; }
                    else { const long nob = lbeg[k1] - lbeg[k0], nlm = k1 - k0; used = (long)(UVS_LN_REC + 2 * UVS_LN_EY) * nob + 20 * nlm + (nlist + 1) / 2; }
                    if (used > stage_cap) P.overflow = true;
                    P.max_used = std::max(P.max_used, (int)used);
                }
                if (dbg_lists) {
                    fprintf(stderr, "chunk %d type %d lm [%d,%d):\n", qc, chunks[UVS_CHUNK_INTS * qc], chunks[UVS_CHUNK_INTS * qc + 1], chunks[UVS_CHUNK_INTS * qc + 2]);
                    for (int wv = 0; wv < NW; ++wv) {
                        fprintf(stderr, "  wave %d:", wv);
                        for (int q = 0; q < GPW; ++q) { const int g = wv * GPW + q; fprintf(stderr, " b%d.%d(%d,%d)", g_blk[g], g_part[g], P.lists[base + g + 1] - P.lists[base + g], P.lists[base + NG + 2 + g] - P.lists[base + NG + 1 + g]); }
                        fprintf(stderr, "\n");
                    }
                }
            }
        });
        const int nparts = (int)part.size();
        size_t total = 0;
        for (const auto& P : part) total += P.lists.size();
        lists.resize(total);
        size_t at = 0;
        for (int t = 0; t < nparts; ++t) {
            const Part& P = part[t];
            if (P.overflow) { err = "internal: chunk layout exceeds the LDS staging area"; return UVS_ERR_CAPACITY; }
            h.max_chunk_doubles = std::max(h.max_chunk_doubles, P.max_used);
            if (!P.lists.empty()) std::memcpy(lists.data() + at, P.lists.data(), P.lists.size() * sizeof(int));
            for (int qc = P.q0; qc < P.q1; ++qc) chunks[UVS_CHUNK_INTS * qc + 3] += (int)at;      // part-relative -> absolute (parts are in chunk order: thread t took the t-th range)
            at += P.lists.size();
        }
    }
    }
    h.n_chunks = (int)chunks.size() / UVS_CHUNK_INTS;
    // re-damping (uvs_solve_kernel.h: redamp_chunk) keeps its per-line table and gradient rows in the record area of a line chunk
    h.chol_half_ok = 1;
    if (have_prior) for (int b = 0; b < w->prior->n_blocks; ++b) if (w->prior->block_kind[b] == UVS_BLOCK_SPEEDBIAS && w->prior->block_frame[b] >= 2) h.chol_half_ok = 0;
    if (std::getenv("UVS_CHOL_FULL_ROWS")) h.chol_half_ok = 0;
    h.redamp_ok = (!td_on && !ex_on && !relo_on) ? 1 : 0;
    for (int qc = 0; qc < h.n_chunks && h.redamp_ok; ++qc)
        if (chunks[UVS_CHUNK_INTS * qc] == 1) { const long nob = lbeg[chunks[UVS_CHUNK_INTS * qc + 2]] - lbeg[chunks[UVS_CHUNK_INTS * qc + 1]], nlm = chunks[UVS_CHUNK_INTS * qc + 2] - chunks[UVS_CHUNK_INTS * qc + 1]; if (34 * nlm + 6 * nob > (long)UVS_LN_REC * nob) h.redamp_ok = 0; }
        else {      // point chunk: redamp_chunk's gradient rows Gb[(nob + nlm)][6] sit in front of the E rows at rec + nob * pt_rec -- a chunk made mostly of landmarks WITHOUT observations would run into them
            const long nob = pbeg[chunks[UVS_CHUNK_INTS * qc + 2]] - pbeg[chunks[UVS_CHUNK_INTS * qc + 1]], nlm = chunks[UVS_CHUNK_INTS * qc + 2] - chunks[UVS_CHUNK_INTS * qc + 1];
            if (6 * (nob + nlm) > (long)h.pt_rec * nob) h.redamp_ok = 0;
        }
    lap_("lists");
    // layout
    int d = (int)((sizeof(DevWin) + 7) / 8);
    h.d_frames = d; d += UVS_XDIM;
    h.d_invd = d; d += rup(std::max(h.n_points, 1), 2);
    h.d_ptmeas = d; d += 6 * h.pt_stride;
    h.d_ptvel = d; d += td_on ? 6 * h.pt_stride : 0;
    h.d_line = d; d += 4 * std::max(h.n_lines, 1);
    h.d_lnmeas = d; d += 9 * h.ln_stride;
    // S blocks touched by the prior (all pairs of frames that own a kept pose / speed-bias block)
    std::vector<int> pblk;
    if (have_prior) {
        bool in[UVS_NUM_FRAMES] = {false};
        for (int b = 0; b < w->prior->n_blocks; ++b)
            if (w->prior->block_kind[b] == UVS_BLOCK_POSE || w->prior->block_kind[b] == UVS_BLOCK_SPEEDBIAS) in[w->prior->block_frame[b]] = true;
            else if (w->prior->block_kind[b] == UVS_BLOCK_TD && td_on) in[UVS_NUM_FRAMES - 1] = true;       // td lives in the last frame's block row
            else if (w->prior->block_kind[b] == UVS_BLOCK_EX_POSE && ex_on) for (int q = 0; q < 6; ++q) in[q] = true;      // ex dofs live in frames 0..5
        for (int fa = 0; fa < UVS_NUM_FRAMES; ++fa) for (int fb = 0; fb <= fa; ++fb) if (in[fa] && in[fb]) pblk.push_back((fa * (fa + 1) / 2 + fb) | (fa << 8) | (fb << 12));      // block | fa << 8 | fb << 12
    }
    // compact image: only the entries whose row AND column are prior columns (a pose block owns 6 of the 16 rows of its S block, so 36 of 272
    // entries of a pose-pose block): what the per-linearization add reads (value + S offset) shrinks from 15 k to ~2.6 k entries
    int n_cimg = 0;      // (the table itself is generated on the device: setup_window)
    if (have_prior) {
        int inv_s[UVS_RD]; for (int q = 0; q < UVS_RD; ++q) inv_s[q] = -1;
        const uvs_prior& p = *w->prior;
        if (!p.linearized_jacobians || !p.linearized_residuals || !p.x0) { err = "null array"; return UVS_ERR_INVALID_ARG; }
        for (int b = 0; b < p.n_blocks; ++b) {
            // kind <-> global size: pose / extrinsic 7, speed-bias 9, time offset 1; x0_off addresses x0[UVS_PRIOR_X0_LEN]
            const int kind = p.block_kind[b], want = kind == UVS_BLOCK_SPEEDBIAS ? 9 : kind == UVS_BLOCK_TD ? 1 : 7;
            if (kind < UVS_BLOCK_POSE || kind > UVS_BLOCK_TD || p.block_size[b] != want) { err = "prior block kind / size mismatch"; return UVS_ERR_INVALID_ARG; }
            if (p.x0_off[b] < 0 || p.x0_off[b] > UVS_PRIOR_X0_LEN - p.block_size[b]) { err = "prior x0 offset out of range"; return UVS_ERR_INVALID_ARG; }
            const int loc = p.block_size[b] == 7 ? 6 : p.block_size[b];
            int basecol = -1;
            if (p.block_kind[b] == UVS_BLOCK_POSE) basecol = 16 * p.block_frame[b];
            else if (p.block_kind[b] == UVS_BLOCK_SPEEDBIAS) basecol = 16 * p.block_frame[b] + 6;
            else if (p.block_kind[b] == UVS_BLOCK_TD && td_on) basecol = UVS_TD_INDEX;
            const bool exb = p.block_kind[b] == UVS_BLOCK_EX_POSE && ex_on;
            for (int q = 0; q < loc; ++q) { const int si = exb ? UVS_EX_INDEX(q) : (basecol < 0 ? -1 : basecol + q); if (si >= 0) inv_s[si] = p.block_idx[b] + q; }
        }
        { int mapped = 0; for (int q = 0; q < UVS_RD; ++q) mapped += inv_s[q] >= 0; n_cimg = mapped * (mapped + 1) / 2; }      // pairs of mapped S indices i >= j
    }
    h.d_imu = d; d += std::max(h.n_imu, 1) * UVS_IMU_STRIDE;
    h.d_prior = d; d += h.prior_n * h.prior_n + 2 * h.prior_n + 144;
    int i = 2 * d;
    h.i_pt_lm = i; i += h.pt_stride; h.i_pt_fi = i; i += h.pt_stride; h.i_pt_fj = i; i += h.pt_stride; h.i_pt_beg = i; i += rup(h.n_points + 1, 2);
    h.i_pt_eidx = i; i += relo_on ? h.pt_stride : 0;
    h.i_ln_lm = i; i += h.ln_stride; h.i_ln_fj = i; i += h.ln_stride; h.i_ln_vp = i; i += h.ln_stride; h.i_ln_beg = i; i += rup(h.n_lines + 1, 2);
    h.i_imu = i; i += 2 * std::max(h.n_imu, 1);
    h.i_prior = i; i += 80 + UVS_MAX_PRIOR_DIM + UVS_RD + UVS_NBLK;
    h.i_chunks = i; i += UVS_CHUNK_INTS * std::max(h.n_chunks, 1);
    h.i_wblk = i; i += NG;
    h.i_lists = i; i += (int)lists.size() + 2;
    h.blob_bytes = rup(4 * i, 256);
    // workspace layout
    int wsz = 0;
    h.w_invd0 = wsz; wsz += rup(std::max(h.n_points, 1), 2); h.w_invd1 = wsz; wsz += rup(std::max(h.n_points, 1), 2);
    h.w_line0 = wsz; wsz += 4 * std::max(h.n_lines, 1); h.w_line1 = wsz; wsz += 4 * std::max(h.n_lines, 1);
    h.w_ltrig0 = wsz; wsz += 8 * std::max(h.n_lines, 1); h.w_ltrig1 = wsz; wsz += 8 * std::max(h.n_lines, 1);
    h.w_scale_pt = wsz; wsz += rup(std::max(h.n_points, 1), 2); h.w_scale_ln = wsz; wsz += 4 * std::max(h.n_lines, 1);
    h.w_pt_E = wsz; wsz += 6 * (h.n_pt_obs + XS * h.n_points) + 6; h.w_pt_x = wsz; wsz += 4 * std::max(h.n_points, 1);
    h.w_ln_Y = wsz; wsz += 24 * std::max(h.n_ln_obs, 1); h.w_ln_x = wsz; wsz += UVS_LN_X * std::max(h.n_lines, 1);
    h.w_imu = wsz; wsz += std::max(h.n_imu, 1) * UVS_WIMU_STRIDE;
    h.w_imu_w = wsz; wsz += std::max(h.n_imu, 1) * UVS_IMU_WS;
    h.w_out = wsz; wsz += UVS_XDIM + std::max(h.n_points, 0) + 4 * std::max(h.n_lines, 0);
    h.n_pblk = (int)pblk.size();
    h.w_prior_h0 = wsz; wsz += UVS_PH_DOUBLES(h.prior_n);      // H0 = J0^T J0, g0, c0, diag(H0) per S index: written once per solve by setup_window
    h.n_cimg = n_cimg;
    h.w_cimg = wsz; wsz += n_cimg + 2;      // 2 x n_cimg ints
    h.w_relo2 = wsz; if (relo2) wsz += UVS_RELO2_DOUBLES;
    h.w_gacc = wsz; wsz += 8 * UVS_GROWS * UVS_GT;      // (the 512-thread k_solve: gather accumulators of the last linearization)
    h.ws_doubles = rup(wsz, 32);
    // fill
    char* B = nullptr;
    if (dst && dst->base) {      // straight into the pinned staging buffer when it has room (blob offsets are multiples of 256 bytes either way)
        const size_t at = dst->bump->fetch_add((size_t)h.blob_bytes);
        if (at + (size_t)h.blob_bytes <= dst->cap) { B = dst->base + at; dst->off = (long long)at; std::memset(B, 0, (size_t)h.blob_bytes); }
    }
    if (!B) {
        const size_t base = out.size();
        out.resize(base + h.blob_bytes, 0);
        B = out.data() + base;
    }
    double* D = (double*)B; int* I = (int*)B;
    fill_values(B, h, w, td_on, inner_threads);
    // ---- the tables (index bookkeeping)
    if (relo_on) for (int k = 0; k < h.n_pt_obs; ++k) I[h.i_pt_eidx + k] = eidx[k];
    for (int k = 0; k < h.n_pt_obs; ++k) { I[h.i_pt_lm + k] = w->pt_lm[k]; I[h.i_pt_fi + k] = w->pt_fi[k]; I[h.i_pt_fj + k] = w->pt_fj[k]; }
    for (int k = 0; k <= h.n_points; ++k) I[h.i_pt_beg + k] = pbeg[k];
    for (int k = 0; k < h.n_ln_obs; ++k) { I[h.i_ln_lm + k] = w->ln_lm[k]; I[h.i_ln_fj + k] = w->ln_fj[k]; I[h.i_ln_vp + k] = w->ln_has_vp[k] ? 1 : 0; }
    for (int k = 0; k <= h.n_lines; ++k) I[h.i_ln_beg + k] = lbeg[k];
    for (int b = 0; b < h.n_imu; ++b) { I[h.i_imu + 2 * b] = w->imu[b].frame_i; I[h.i_imu + 2 * b + 1] = w->imu[b].skip ? 1 : 0; }
    if (have_prior) {
        const uvs_prior& p = *w->prior;
        int* pt = I + h.i_prior;
        for (int q = 0; q < 80 + UVS_MAX_PRIOR_DIM + UVS_RD + UVS_NBLK; ++q) pt[q] = -1;
        for (int b = 0; b < p.n_blocks; ++b) {
            pt[b] = p.block_kind[b]; pt[16 + b] = p.block_frame[b]; pt[32 + b] = p.block_size[b]; pt[48 + b] = p.block_idx[b]; pt[64 + b] = p.x0_off[b];
            const int loc = p.block_size[b] == 7 ? 6 : p.block_size[b];
            int basecol = -1;
            if (p.block_kind[b] == UVS_BLOCK_POSE) basecol = 16 * p.block_frame[b];
            else if (p.block_kind[b] == UVS_BLOCK_SPEEDBIAS) basecol = 16 * p.block_frame[b] + 6;
            else if (p.block_kind[b] == UVS_BLOCK_TD && td_on) basecol = UVS_TD_INDEX;
            const bool exb = p.block_kind[b] == UVS_BLOCK_EX_POSE && ex_on;
            // a constant Ex_Pose (ESTIMATE_EXTRINSIC == 0) drops its columns; a free one maps dof q to the spare slot of frame q
            for (int q = 0; q < loc; ++q) {
                const int si = exb ? UVS_EX_INDEX(q) : (basecol < 0 ? -1 : basecol + q);
                pt[80 + p.block_idx[b] + q] = si;
                if (si >= 0) pt[80 + UVS_MAX_PRIOR_DIM + si] = p.block_idx[b] + q;      // S index -> prior column
            }
        }
        for (size_t q = 0; q < pblk.size(); ++q) pt[80 + UVS_MAX_PRIOR_DIM + UVS_RD + q] = pblk[q];

    }
    for (size_t q = 0; q < chunks.size(); ++q) I[h.i_chunks + q] = chunks[q];
    if (!lists.empty()) std::memcpy(I + h.i_lists, lists.data(), lists.size() * sizeof(int));
    for (int q = 0; q < NG; ++q) I[h.i_wblk + q] = wblk[q];
    lap_("blob");
    hdr = h;
    if (cache && !relo_on && h.n_pt_obs + h.n_ln_obs >= kPackCacheMinObs) cache->store(w_in, opts, chunk_grid, h);
    return UVS_OK;
}

static int ensure(uvs_solver* s, void** p, size_t* cap, size_t need) {
    if (*cap >= need) return UVS_OK;
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    HIPCHK(s, hipMalloc(p, need));
    *cap = need;
    return UVS_OK;
}

static int ensure_pinned(uvs_solver* s, char** p, size_t* cap, size_t need) {      // grow-only (pinning costs milliseconds: never per call)
    if (*cap >= need) return UVS_OK;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = need + need / 2 + 4096;
    HIPCHK(s, hipHostMalloc((void**)p, want, hipHostMallocDefault));
    *cap = want;
    return UVS_OK;
}

// gathers the per-window outputs into one contiguous buffer: tab[3 b] = {source offset in ws, doubles, destination offset}; the reports
// follow the states (rep_dst = offset of the report array in `out`, in doubles; sizeof(uvs_report) is a multiple of 8)
static_assert(sizeof(uvs_report) % 8 == 0, "uvs_report is copied as doubles");
__global__ void k_pack_outputs(const double* ws, const long long* tab, double* out, const uvs_report* reps, long long rep_dst) {
    const long long src = tab[3 * blockIdx.x], cnt = tab[3 * blockIdx.x + 1], dst = tab[3 * blockIdx.x + 2];
    for (long long t = threadIdx.x; t < cnt; t += blockDim.x) out[dst + t] = ws[src + t];
    constexpr int RD = (int)(sizeof(uvs_report) / 8);
    const double* r = (const double*)(reps + blockIdx.x);
    for (int t = threadIdx.x; t < RD; t += blockDim.x) out[rep_dst + (long long)blockIdx.x * RD + t] = r[t];
}

// out_direct: (uvs_batch_stream) every staged header gets the address of its window's slot in the pinned result buffer (DevWin::out_host): k_solve then writes the final state there itself
static int upload_windows(uvs_solver* s, int n, const uvs_window* const* ws, bool wait, int chunk_grid = 0, bool out_direct = false) {
    if (!s || n < 1 || !ws) return UVS_ERR_INVALID_ARG;
    if (n > s->max_batch) { s->err = "batch larger than max_batch"; return UVS_ERR_CAPACITY; }
    HIPCHK(s, hipSetDevice(s->device));
    s->hdrs.resize(n); s->blob_off.resize(n); s->ws_off.resize(n);
    long long wtot = 0;
    for (int b = 0; b < n; ++b)
        if (ws[b] && (ws[b]->n_points > s->max_points || ws[b]->n_point_obs + std::max(ws[b]->n_relo_obs, 0) > s->max_point_obs || ws[b]->n_lines > s->max_lines || ws[b]->n_line_obs > s->max_line_obs)) {
            s->err = "window exceeds the capacity given to uvs_create (max_points / max_point_obs / max_lines / max_line_obs)"; s->n_loaded = 0; return UVS_ERR_CAPACITY;
        }
    // Packing (index bookkeeping of the gather lists: the analogue of Ceres' problem construction) is independent per window: a batch is
    // packed by several host threads into per-window buffers and concatenated -- 0.3 ms per window on one core was 83 ms for the 256-window
    // batch, 45 x the solve it feeds.  UVS_PACK_THREADS overrides the thread count (1 = the serial path, also taken for small batches).
    int nthreads = 1;
    static const bool sprof_ = std::getenv("UVS_STREAM_PROFILE") != nullptr;      // stage times of an upload on stderr (where the end-to-end rate of uvs_batch_stream goes)
    const auto tp0_ = std::chrono::steady_clock::now();
    auto tp1_ = tp0_, tp2_ = tp0_, tp3_ = tp0_;
    size_t packed_total = 0;      // > 0: the windows sit in s->slot_blobs (threaded path) and go straight into the pinned staging buffer below
    bool packed_direct = false;   // ... or are there already (packed in place)
    if (n >= 8) {
        const char* env = std::getenv("UVS_PACK_THREADS");
        nthreads = env ? std::atoi(env) : (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency() / 2));      // (half the hardware threads at most: SMT siblings share a core)
        nthreads = std::max(1, std::min(nthreads, n));
    }
    bool values_only = false;      // structure-cache hit AND the device still holds this window's tables: only the value sections travel
    if (n == 1 && ws[0] && ws[0]->n_point_obs + ws[0]->n_line_obs >= kPackCacheMinObs && !std::getenv("UVS_NO_PACK_CACHE")) {
        if (!s->pack_cache) s->pack_cache = new PackCache();
        const bool was_valid = s->pack_cache->valid, dev = s->pack_cache->device_holds_tables;
        s->blob_off[0] = 0;
        int rc = pack_window(ws[0], s->opts, s->host_blobs, s->hdrs[0], s->err, chunk_grid, s->pack_cache);
        if (rc != UVS_OK) { s->n_loaded = 0; return rc; }
        values_only = !out_direct && was_valid && dev && s->pack_cache->valid && s->pack_cache->device_holds_tables;      // (a miss resets both flags; the stream patches every staged header -- DevWin::out_host -- so the whole blob travels)
    } else if (nthreads == 1) {
        s->host_blobs.clear();
        if (s->pack_cache) { s->pack_cache->valid = false; s->pack_cache->device_holds_tables = false; }
        for (int b = 0; b < n; ++b) {
            s->blob_off[b] = (long long)s->host_blobs.size();
            int rc = pack_window(ws[b], s->opts, s->host_blobs, s->hdrs[b], s->err, chunk_grid, nullptr);
            if (rc != UVS_OK) { s->n_loaded = 0; return rc; }
        }
    } else {
        if (s->pack_cache) { s->pack_cache->valid = false; s->pack_cache->device_holds_tables = false; }
        // per-slot buffers that live in the handle: a fresh 260 KB vector per window was a page fault per 4 KB of it, every batch (0.9 ms per window
        // on a cold buffer against 0.12 ms on a warm one)
        if ((int)s->slot_blobs.size() < n) s->slot_blobs.resize(n);
        std::vector<int> rcs(n, UVS_OK); std::vector<std::string> errs(n); std::vector<long long> placed(n, -1);
        // The windows go STRAIGHT into the pinned staging buffer when it is large enough (it is from the second batch of a size on: the first one takes the vectors and
        // sizes the buffer): every worker claims room with an atomic bump, so the blobs sit in completion order -- the kernel finds them through blob_off.  The buffer
        // may still feed the previous upload's copy: drain first.
        HIPCHK(s, hipStreamSynchronize(s->stream));
        std::atomic<size_t> bump{0};
        const size_t direct_cap = s->h_up_cap > (size_t)n * 40 + 64 ? s->h_up_cap - (size_t)n * 40 - 64 : 0;
        const auto job = [&](int t) {
            for (int b = t; b < n; b += nthreads) {
                PackDst d{&bump, direct_cap ? s->h_up : nullptr, direct_cap, -1};
                s->slot_blobs[b].clear();
                rcs[b] = pack_window(ws[b], s->opts, s->slot_blobs[b], s->hdrs[b], errs[b], chunk_grid, nullptr, &d);
                placed[b] = d.off;
            }
        };
        if (!s->pool) s->pool = new PackPool();
        if (s->pool->ensure(nthreads)) s->pool->run(nthreads, job);
        else { const int nt_ = nthreads; nthreads = 1; job(0); nthreads = nt_; }      // (no worker threads: this thread packs everything)
        bool all_placed = direct_cap > 0;
        for (int b = 0; b < n; ++b) {
            if (rcs[b] != UVS_OK) { s->err = errs[b]; s->n_loaded = 0; return rcs[b]; }      // the first failing window in batch order, as the serial path reports it
            if (placed[b] < 0) all_placed = false;
        }
        size_t total = 0;
        if (all_placed) { for (int b = 0; b < n; ++b) s->blob_off[b] = placed[b]; total = bump.load(); packed_direct = true; }
        else {
            // (a window that found no room has its blob in its vector; one that did is copied back out: this path runs when the batch outgrew the buffer)
            for (int b = 0; b < n; ++b) if (placed[b] >= 0) s->slot_blobs[b].assign(s->h_up + placed[b], s->h_up + placed[b] + s->hdrs[b].blob_bytes);
            for (int b = 0; b < n; ++b) { s->blob_off[b] = (long long)total; total += s->slot_blobs[b].size(); }
        }
        packed_total = total;
    }
    tp1_ = std::chrono::steady_clock::now();      // packing ends here; the offset tables and the drain of the stream follow
    for (int b = 0; b < n; ++b) { s->ws_off[b] = wtot; wtot += s->hdrs[b].ws_doubles; }
    s->out_tab.resize(3 * (size_t)n); s->out_total = 0;
    for (int b = 0; b < n; ++b) {
        const DevWin& h = s->hdrs[b];
        const long long cnt = UVS_XDIM + (long long)h.n_points + 4 * (long long)h.n_lines;
        s->out_tab[3 * b] = s->ws_off[b] + h.w_out; s->out_tab[3 * b + 1] = cnt; s->out_tab[3 * b + 2] = s->out_total;
        s->out_total += cnt;
    }
    int rc;
    const size_t raw_bytes = packed_total ? packed_total : s->host_blobs.size();
    const size_t blob_bytes = (raw_bytes + 7) & ~(size_t)7, up_bytes = blob_bytes + (size_t)n * 40;
    // the staging buffer may still feed the previous upload's copy (the single-window path does not wait for it): drain before reuse
    HIPCHK(s, hipStreamSynchronize(s->stream));
    tp2_ = std::chrono::steady_clock::now();
    // every upload is staged in pinned memory together with its tables: ONE DMA copy that the host need not wait for (a copy from the pageable vector
    // is staged by the runtime anyway, synchronously and on one thread); a large blob (configs[3]: 13 MB) is moved there by several threads
    const bool staged = true;
    if ((rc = ensure_pinned(s, &s->h_up, &s->h_up_cap, staged ? up_bytes + (packed_total && !packed_direct ? up_bytes / 8 + 4096 : 0) : (size_t)n * 40)) != UVS_OK) return rc;      // (slack: the next batch of this size packs in place)
    { char* before = s->d_blobs; if ((rc = ensure(s, (void**)&s->d_blobs, &s->d_blobs_cap, up_bytes)) != UVS_OK) return rc; if (s->d_blobs != before) values_only = false; }
    // all doubles of a blob precede its int tables (pack_window: i = 2 d), so the value sections are ONE prefix
    const size_t value_bytes = values_only ? (size_t)4 * (size_t)s->hdrs[0].i_pt_lm : 0;
    if ((rc = ensure(s, (void**)&s->d_outpack, &s->d_outpack_cap, (size_t)s->out_total * 8 + (size_t)n * sizeof(uvs_report))) != UVS_OK) return rc;
    if ((rc = ensure(s, (void**)&s->d_ws, &s->d_ws_cap, (size_t)wtot * 8)) != UVS_OK) return rc;
    if ((rc = ensure(s, (void**)&s->d_reports, &s->d_rep_cap, (size_t)n * sizeof(uvs_report))) != UVS_OK) return rc;
    if (packed_direct) { /* the blobs are in the staging buffer already */ }
    else if (packed_total) {      // every packing thread moves its own windows (67 MB for 256 canonical windows: one core would need ~10 ms)
        const auto cp = [&](int t) { for (int b = t; b < n; b += nthreads) std::memcpy(s->h_up + s->blob_off[b], s->slot_blobs[b].data(), s->slot_blobs[b].size()); };
        if (s->pool && s->pool->ensure(nthreads)) s->pool->run(nthreads, cp); else for (int t = 0; t < nthreads; ++t) cp(t);
    } else if (s->host_blobs.size() > ((size_t)4 << 20)) {
        const size_t nb_ = values_only ? value_bytes : s->host_blobs.size(); const int ct = pack_inner_threads(1 << 30);
        pack_parallel((int)((nb_ + 65535) >> 16), ct, [&](int c0, int c1, int) { const size_t a0 = (size_t)c0 << 16, a1 = std::min(nb_, (size_t)c1 << 16); if (a1 > a0) std::memcpy(s->h_up + a0, s->host_blobs.data() + a0, a1 - a0); });
    } else std::memcpy(s->h_up, s->host_blobs.data(), s->host_blobs.size());
    tp3_ = std::chrono::steady_clock::now();
    if (out_direct) {
        if ((rc = ensure_pinned(s, &s->h_out, &s->h_out_cap, (size_t)s->out_total * 8 + (size_t)n * sizeof(uvs_report))) != UVS_OK) return rc;
        void* dp = nullptr; HIPCHK(s, hipHostGetDevicePointer(&dp, s->h_out, 0));
        for (int b = 0; b < n; ++b) ((DevWin*)(s->h_up + s->blob_off[b]))->out_host = (int64_t)(uintptr_t)((double*)dp + s->out_tab[3 * (size_t)b + 2]);
    }
    long long* tabs = (long long*)(s->h_up + (staged ? blob_bytes : 0));
    std::memcpy(tabs, s->blob_off.data(), (size_t)n * 8);
    std::memcpy(tabs + n, s->ws_off.data(), (size_t)n * 8);
    std::memcpy(tabs + 2 * (size_t)n, s->out_tab.data(), (size_t)n * 24);
    s->d_blob_off = (long long*)(s->d_blobs + blob_bytes); s->d_ws_off = s->d_blob_off + n; s->d_out_tab = s->d_blob_off + 2 * (size_t)n;
    if (values_only) {      // the tables of this window are on the device already (structure cache): the value prefix and the three small offset tables
        HIPCHK(s, hipMemcpyAsync(s->d_blobs, s->h_up, value_bytes, hipMemcpyHostToDevice, s->stream));
        HIPCHK(s, hipMemcpyAsync(s->d_blobs + blob_bytes, s->h_up + blob_bytes, (size_t)n * 40, hipMemcpyHostToDevice, s->stream));
    } else if (staged) HIPCHK(s, hipMemcpyAsync(s->d_blobs, s->h_up, up_bytes, hipMemcpyHostToDevice, s->stream));
    else {
        HIPCHK(s, hipMemcpyAsync(s->d_blobs, s->host_blobs.data(), s->host_blobs.size(), hipMemcpyHostToDevice, s->stream));
        HIPCHK(s, hipMemcpyAsync(s->d_blobs + blob_bytes, s->h_up, (size_t)n * 40, hipMemcpyHostToDevice, s->stream));
        wait = true;      // host_blobs is reused by the next upload
    }
    if (sprof_ && n > 1) {
        const auto tp4_ = std::chrono::steady_clock::now();
        auto ms_ = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "upload n=%d threads=%d bytes=%zu: pack %.3f ms, tables + drain of the stream %.3f, copy to pinned %.3f, enqueue %.3f\n", n, nthreads, up_bytes, ms_(tp0_, tp1_), ms_(tp1_, tp2_), ms_(tp2_, tp3_), ms_(tp3_, tp4_));
    }
    if (wait) HIPCHK(s, hipStreamSynchronize(s->stream));
    if (s->pack_cache && s->pack_cache->valid && n == 1) s->pack_cache->device_holds_tables = true;
    s->n_loaded = n;
    return UVS_OK;
}

// ---------------------------------------------------------------- MARGIN_OLD on the device (round 3)
// The factors the reference marginalizes (estimator.cpp:1002-1135) form a small window of their own; ONE linearization of it by the solver's own kernels
// (k_marg_linearize) delivers the assembled and landmark-eliminated system, and only the elimination of frame 0's 15 dofs and the n x n factorization stay
// on the host (uvs_marg.h: marg_finish).  Returns UVS_OK, an error, or kMargFallback when the host path must take the call (no such factors, a landmark
// block that is not safely regular, relocalization blocks in the way).
namespace { constexpr int kMargFallback = 1000; }
struct MargDevScratch {
    std::vector<int32_t> pt_lm, pt_fi, pt_fj, ln_lm, ln_fj, ln_vpf; std::vector<double> pt_pi, pt_pj, pt_vi, pt_vj, pt_tdi, pt_tdj, invd, ln_sp, ln_ep, ln_vp, lorth;
    std::vector<uvs_imu_block> imu; std::vector<int> pmap, lmap, lstart;
    std::vector<char> blob;
    char* d_blob = nullptr; size_t d_blob_cap = 0; double* d_ws = nullptr; size_t d_ws_cap = 0; double* d_out = nullptr; char* h_out = nullptr; size_t h_out_cap = 0;
    char* h_up = nullptr; size_t h_up_cap = 0;
};
static void free_marg_scratch(MargDevScratch* m) {
    if (!m) return;
    if (m->d_blob) (void)hipFree(m->d_blob); if (m->d_ws) (void)hipFree(m->d_ws); if (m->d_out) (void)hipFree(m->d_out);
    if (m->h_out) (void)hipHostFree(m->h_out); if (m->h_up) (void)hipHostFree(m->h_up);
    delete m;
}
// The sub-window of the factors MARGIN_OLD reads (estimator.cpp:1002-1135): the prior, the IMU link of frame 0, the observations of the points anchored in frame 0 and of the lines
// that start there (without their anchor observation).  Its arrays live in M; used[] = the frame blocks (ids: pose f -> f ; speedbias f -> 11 + f ; ex -> 22 ; td -> 23) it touches.
static int marg_build_sub(uvs_solver* s, const uvs_window* w, MargDevScratch& M, bool used[24], uvs_window& sub, std::string& err_) {
    std::string& serr = err_;
    for (int k = 0; k < 24; ++k) used[k] = false;
    const bool td_on = s->opts.estimate_td != 0;
    const int NFR = UVS_NF;
    // the sub-window below is cut out of the caller's arrays BEFORE pack_window sees them: same checks first
    { const int rv = validate_window(w, serr); if (rv != UVS_OK) return rv; }
    if (td_on && w->n_point_obs > 0 && (!w->pt_vel_i || !w->pt_vel_j || !w->pt_td_i || !w->pt_td_j)) { serr = "estimate_td needs pt_vel_i / pt_vel_j / pt_td_i / pt_td_j"; return UVS_ERR_INVALID_ARG; }
    // ---- the sub-window: which blocks it touches (ids: pose f -> f ; speedbias f -> 11 + f ; ex -> 22 ; td -> 23)
    const bool have_prior = w->prior && w->prior->n > 0;
    if (have_prior) for (int b = 0; b < w->prior->n_blocks; ++b) {
        const uvs_prior& p = *w->prior;
        used[p.block_kind[b] == UVS_BLOCK_POSE ? p.block_frame[b] : p.block_kind[b] == UVS_BLOCK_SPEEDBIAS ? NFR + p.block_frame[b] : p.block_kind[b] == UVS_BLOCK_TD ? 23 : 22] = true;
    }
    M.imu.clear();
    for (int b = 0; b < w->n_imu; ++b) {
        if (w->imu[b].frame_i != 0 || !(w->imu[b].sum_dt < 10.0)) continue;
        uvs_imu_block ib = w->imu[b]; ib.skip = 0; M.imu.push_back(ib);
        used[0] = used[NFR] = used[1] = used[NFR + 1] = true;
    }
    M.pmap.assign(std::max(w->n_points, 1), -1); M.lmap.assign(std::max(w->n_lines, 1), -1); M.lstart.assign(std::max(w->n_lines, 1), -1);
    M.pt_lm.clear(); M.pt_fi.clear(); M.pt_fj.clear(); M.pt_pi.clear(); M.pt_pj.clear(); M.pt_vi.clear(); M.pt_vj.clear(); M.pt_tdi.clear(); M.pt_tdj.clear(); M.invd.clear();
    for (int k = 0; k < w->n_point_obs; ++k) {
        if (w->pt_fi[k] != 0) continue;
        const int lm = w->pt_lm[k];
        if (M.pmap[lm] < 0) { M.pmap[lm] = (int)M.invd.size(); M.invd.push_back(w->inv_depth[lm]); }
        M.pt_lm.push_back(M.pmap[lm]); M.pt_fi.push_back(0); M.pt_fj.push_back(w->pt_fj[k]);
        for (int q = 0; q < 3; ++q) { M.pt_pi.push_back(w->pt_pi[3 * k + q]); M.pt_pj.push_back(w->pt_pj[3 * k + q]); }
        if (td_on) { for (int q = 0; q < 2; ++q) { M.pt_vi.push_back(w->pt_vel_i[2 * k + q]); M.pt_vj.push_back(w->pt_vel_j[2 * k + q]); } M.pt_tdi.push_back(w->pt_td_i[k]); M.pt_tdj.push_back(w->pt_td_j[k]); }
        used[0] = used[w->pt_fj[k]] = used[22] = true; if (td_on) used[23] = true;
    }
    M.ln_lm.clear(); M.ln_fj.clear(); M.ln_vpf.clear(); M.ln_sp.clear(); M.ln_ep.clear(); M.ln_vp.clear(); M.lorth.clear();
    for (int k = 0; k < w->n_line_obs; ++k) if (M.lstart[w->ln_lm[k]] < 0) M.lstart[w->ln_lm[k]] = w->ln_fj[k];
    for (int k = 0; k < w->n_line_obs; ++k) {
        const int lm = w->ln_lm[k], fj = w->ln_fj[k];
        if (M.lstart[lm] != 0 || fj == 0) continue;      // lines that start in frame 0, without the anchor observation (estimator.cpp:1102-1104)
        if (M.lmap[lm] < 0) { M.lmap[lm] = (int)(M.lorth.size() / 4); for (int q = 0; q < 4; ++q) M.lorth.push_back(w->line_orth[4 * lm + q]); }
        M.ln_lm.push_back(M.lmap[lm]); M.ln_fj.push_back(fj); M.ln_vpf.push_back(w->ln_has_vp[k] ? 1 : 0);
        for (int q = 0; q < 3; ++q) { M.ln_sp.push_back(w->ln_sp[3 * k + q]); M.ln_ep.push_back(w->ln_ep[3 * k + q]); M.ln_vp.push_back(w->ln_vp[3 * k + q]); }
        used[fj] = true;
    }
    if (M.imu.empty() && M.pt_lm.empty() && M.ln_lm.empty() && !have_prior) return kMargFallback;
    std::memset(&sub, 0, sizeof(sub));
    std::memcpy(sub.pose, w->pose, sizeof(sub.pose)); std::memcpy(sub.speedbias, w->speedbias, sizeof(sub.speedbias)); std::memcpy(sub.ex_pose, w->ex_pose, sizeof(sub.ex_pose));
    sub.td = w->td; for (int q = 0; q < 7; ++q) sub.relo_pose[q] = q == 6 ? 1.0 : 0.0;
    sub.n_points = (int)M.invd.size(); sub.n_point_obs = (int)M.pt_lm.size(); sub.inv_depth = M.invd.data();
    sub.pt_lm = M.pt_lm.data(); sub.pt_fi = M.pt_fi.data(); sub.pt_fj = M.pt_fj.data(); sub.pt_pi = M.pt_pi.data(); sub.pt_pj = M.pt_pj.data();
    if (td_on) { sub.pt_vel_i = M.pt_vi.data(); sub.pt_vel_j = M.pt_vj.data(); sub.pt_td_i = M.pt_tdi.data(); sub.pt_td_j = M.pt_tdj.data(); }
    sub.n_lines = (int)(M.lorth.size() / 4); sub.n_line_obs = (int)M.ln_lm.size(); sub.line_orth = M.lorth.data();
    sub.ln_lm = M.ln_lm.data(); sub.ln_fj = M.ln_fj.data(); sub.ln_has_vp = M.ln_vpf.data(); sub.ln_sp = M.ln_sp.data(); sub.ln_ep = M.ln_ep.data(); sub.ln_vp = M.ln_vp.data();
    sub.n_imu = (int)M.imu.size(); sub.imu = M.imu.data(); sub.prior = have_prior ? w->prior : nullptr;
    return UVS_OK;
}
// Ordering of the frame blocks of a device-linearized sub-window: the dropped ones (Pose[0], SpeedBias[0]) first, then the kept ones in id order.  map[i] = index of row i in
// k_marg_linearize's padded reduced system (16 x frame + dof, the extrinsic / time-offset slots).
static void marg_frame_order(const bool used[24], std::vector<int>& pos, std::vector<int>& keep_ids, int& md, int& n, std::vector<int>& map) {
    const int NFR = UVS_NF;
    auto lsize = [&](int id) { return id < NFR ? 6 : id < 2 * NFR ? 9 : id == 22 ? 6 : 1; };
    auto pad = [&](int id, int q) { return id < NFR ? 16 * id + q : id < 2 * NFR ? 16 * (id - NFR) + 6 + q : id == 22 ? UVS_EX_INDEX(q) : UVS_TD_INDEX; };
    pos.assign(24, -1); keep_ids.clear(); map.clear();
    md = 0;
    for (int id : {0, NFR}) if (used[id]) { pos[id] = md; md += lsize(id); for (int q = 0; q < lsize(id); ++q) map.push_back(pad(id, q)); }
    int N = md;
    for (int id = 0; id < 24; ++id) if (used[id] && id != 0 && id != NFR) { pos[id] = N; N += lsize(id); keep_ids.push_back(id); for (int q = 0; q < lsize(id); ++q) map.push_back(pad(id, q)); }
    n = N - md;
}
static int marginalize_old_device(uvs_solver* s, const uvs_window* w, uvs_prior* out) {
    if (std::getenv("UVS_MARG_HOST")) return kMargFallback;      // (relocalization blocks are not marginalized, estimator.cpp:1002-1228: the sub-window simply leaves them out)
    const bool prof = std::getenv("UVS_MARG_PROFILE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (!s->marg_dev) s->marg_dev = new MargDevScratch();
    MargDevScratch& M = *s->marg_dev;
    bool used[24]; uvs_window sub;
    { const int rb = marg_build_sub(s, w, M, used, sub, s->err); if (rb != UVS_OK) return rb; }
    // ---- pack with a FREE extrinsic (the prior keeps para_Ex_Pose), upload, one linearization
    uvs_options o = s->opts; o.estimate_extrinsic = 1; o.initial_trust_region_radius = 1e300;
    DevWin h; M.blob.clear();
    const auto tp0 = std::chrono::steady_clock::now();
    int rc = pack_window(&sub, o, M.blob, h, s->err);
    const auto tp1 = std::chrono::steady_clock::now();
    if (rc == UVS_ERR_UNSUPPORTED || rc == UVS_ERR_CAPACITY) return kMargFallback;
    if (rc != UVS_OK) return rc;
    HIPCHK(s, hipSetDevice(s->device));
    const auto ens = [&](void** p, size_t* cap, size_t need) -> int { if (*cap >= need) return UVS_OK; if (*p) (void)hipFree(*p); *p = nullptr; *cap = 0; HIPCHK(s, hipMalloc(p, need + need / 2)); *cap = need + need / 2; return UVS_OK; };
    if ((rc = ens((void**)&M.d_blob, &M.d_blob_cap, M.blob.size())) != UVS_OK) return rc;
    if ((rc = ens((void**)&M.d_ws, &M.d_ws_cap, (size_t)h.ws_doubles * 8)) != UVS_OK) return rc;
    if (!M.d_out) HIPCHK(s, hipMalloc((void**)&M.d_out, MARG_OUT * 8));
    if (!M.h_out) { HIPCHK(s, hipHostMalloc((void**)&M.h_out, MARG_OUT * 8, hipHostMallocDefault)); M.h_out_cap = MARG_OUT * 8; }
    if ((rc = ensure_pinned(s, &M.h_up, &M.h_up_cap, M.blob.size())) != UVS_OK) return rc;
    HIPCHK(s, hipStreamSynchronize(s->stream));      // the staging buffer may still feed the previous call's copy
    const auto tq0 = std::chrono::steady_clock::now();
    std::memcpy(M.h_up, M.blob.data(), M.blob.size());
    const auto tq1 = std::chrono::steady_clock::now();
    HIPCHK(s, hipMemcpyAsync(M.d_blob, M.h_up, M.blob.size(), hipMemcpyHostToDevice, s->stream));
    const KOpts ko = make_kopts(o, 0);
    hipLaunchKernelGGL(k_marg_linearize, dim3(1), dim3(NT), LDS_BYTES, s->stream, M.d_blob, M.d_ws, ko, M.d_out);
    HIPCHK(s, hipGetLastError());
    HIPCHK(s, hipMemcpyAsync(M.h_out, M.d_out, MARG_OUT * 8, hipMemcpyDeviceToHost, s->stream));
    const auto tq2 = std::chrono::steady_clock::now();
    HIPCHK(s, hipStreamSynchronize(s->stream));
    const auto t1 = std::chrono::steady_clock::now();
    if (prof) { auto us = [](auto a_, auto b_) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b_ - a_).count() * 1e-3; };
                std::fprintf(stderr, "[uvs_marginalize] device path: sub-window %.0f us, pack %.0f us (%zu bytes), upload + kernel + download %.0f us (allocations + drain %.0f, copy into pinned %.0f, three enqueues %.0f, wait %.0f)\n",
                             us(t0, tp0), us(tp0, tp1), M.blob.size(), us(tp1, t1), us(tp1, tq0), us(tq0, tq1), us(tq1, tq2), us(tq2, t1)); }
    const double* S = (const double*)M.h_out; const double* g = S + UVS_RD * (UVS_RD + 1) / 2; const double* scal = g + UVS_RD;
    if (scal[1] != 0.0 || !std::isfinite(scal[0])) return kMargFallback;      // a landmark block the reference's eps cut would touch: the host path applies that cut
    // ---- ordering: the dropped frame blocks (Pose[0], SpeedBias[0]) first, then the kept ones in id order
    std::vector<int> pos, keep_ids, map; int md = 0, n = 0;
    marg_frame_order(used, pos, keep_ids, md, n, map);
    const int N = md + n;
    if (n > UVS_MAX_PRIOR_DIM || (int)keep_ids.size() > UVS_MAX_PRIOR_BLOCKS) { s->err = "prior capacity"; return UVS_ERR_CAPACITY; }
    if (md == 0 || n == 0) return kMargFallback;
    std::vector<double>&A = s->eval_scratch.work[0], &bv = s->eval_scratch.work[1];
    A.assign((size_t)N * N, 0.0); bv.assign(N, 0.0);
    for (int i = 0; i < N; ++i) {
        const int ia = map[i];
        bv[i] = g[ia];
        for (int j = 0; j < N; ++j) { const int ib = map[j]; const int hi = ia >= ib ? ia : ib, lo = ia >= ib ? ib : ia; A[(size_t)i * N + j] = S[(size_t)hi * (hi + 1) / 2 + lo]; }
    }
    double us_pre[3] = {(double)std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count() * 1e-3, 0.0, 0.0};
    const int rf = marg_finish(N, md, md, n, A, bv, pos, keep_ids, w, 0, out, s->eval_scratch, prof, us_pre);
    if (rf != UVS_OK) s->err = "marginalization: the linearized system is not finite";
    return rf;
}

extern "C" {

int uvs_batch_upload(uvs_solver* s, int n, const uvs_window* const* ws) { return upload_windows(s, n, ws, true); }

// rep_direct: (uvs_batch_stream after upload_windows(out_direct)) the reports go into the pinned result buffer as well
static int launch_solve(uvs_solver* s, int debug, float* elapsed_ms, bool wait = true, bool rep_direct = false) {
    if (s->n_loaded < 1) { s->err = "no batch uploaded"; return UVS_ERR_INVALID_ARG; }
    HIPCHK(s, hipSetDevice(s->device));
    KOpts ko = make_kopts(s->opts, debug);
    uvs_report* d_reports = s->d_reports;
    if (rep_direct) {
        void* dp = nullptr; HIPCHK(s, hipHostGetDevicePointer(&dp, s->h_out, 0));
        d_reports = (uvs_report*)((double*)dp + s->out_total);
    }
    DebugOut dbg; std::memset(&dbg, 0, sizeof(dbg));
    if (debug) {
        if (!s->d_dbg) HIPCHK(s, hipMalloc((void**)&s->d_dbg, sizeof(double) * (UVS_RD * UVS_RD + 5 * UVS_RD + 40)));
        dbg.S = s->d_dbg; dbg.g = dbg.S + UVS_RD * UVS_RD; dbg.hd = dbg.g + UVS_RD; dbg.dd = dbg.hd + UVS_RD; dbg.step = dbg.dd + UVS_RD; dbg.scal = dbg.step + UVS_RD;
    }
    HIPCHK(s, hipEventRecord(s->ev0, s->stream));
    if (s->ksolve_nt == 512) { if (uvs_k_solve512_launch(s->n_loaded, s->stream, s->d_blobs, s->d_blob_off, s->d_ws, s->d_ws_off, &ko, sizeof(ko), d_reports, &dbg, sizeof(dbg)) != UVS_OK) { s->err = "k_solve (512 threads): argument layout mismatch between the translation units"; return UVS_ERR_HIP; } }
    else hipLaunchKernelGGL(k_solve, dim3(s->n_loaded), dim3(NT), LDS_BYTES, s->stream, s->d_blobs, s->d_blob_off, s->d_ws, s->d_ws_off, ko, d_reports, dbg);
    HIPCHK(s, hipGetLastError());
    HIPCHK(s, hipEventRecord(s->ev1, s->stream));
    if (!wait) return UVS_OK;
    HIPCHK(s, hipStreamSynchronize(s->stream));
    if (elapsed_ms) HIPCHK(s, hipEventElapsedTime(elapsed_ms, s->ev0, s->ev1));
    return UVS_OK;
}

int uvs_batch_solve(uvs_solver* s, float* elapsed_ms) {
    if (!s) return UVS_ERR_INVALID_ARG;
    return launch_solve(s, 0, elapsed_ms);
}

// the two halves of a download: enqueue (gather kernel + ONE copy into pinned memory, nothing waits) and finish (wait, unpack)
// `direct`: the gather kernel writes into the pinned host buffer itself and no copy is enqueued (uvs_batch_stream with UVS_STREAM_D2H_COPY=2; its default lets k_solve write the results)
static int download_enqueue(uvs_solver* s, int n, bool direct = false) {
    // every window's final state (frames | inv_depth | line_orth, written by k_solve into its workspace) and its report are gathered on the
    // device and fetched with ONE copy into pinned memory (256 windows were 256 synchronous round trips once)
    const size_t nst = (size_t)(s->out_tab[3 * (size_t)(n - 1) + 2] + s->out_tab[3 * (size_t)(n - 1) + 1]);      // doubles of the first n states
    const size_t tot = nst * 8 + (size_t)n * sizeof(uvs_report);
    int rc;
    if ((rc = ensure_pinned(s, &s->h_out, &s->h_out_cap, tot)) != UVS_OK) return rc;
    double* out = s->d_outpack;
    if (direct) { void* dp = nullptr; HIPCHK(s, hipHostGetDevicePointer(&dp, s->h_out, 0)); out = (double*)dp; }
    hipLaunchKernelGGL(k_pack_outputs, dim3(n), dim3(256), 0, s->stream, s->d_ws, s->d_out_tab, out, s->d_reports, (long long)nst);
    HIPCHK(s, hipGetLastError());
    if (!direct) HIPCHK(s, hipMemcpyAsync(s->h_out, s->d_outpack, tot, hipMemcpyDeviceToHost, s->stream));
    return UVS_OK;
}
static int download_finish(uvs_solver* s, int n, uvs_state* states, uvs_report* reps) {
    HIPCHK(s, hipStreamSynchronize(s->stream));
    const size_t nst = (size_t)(s->out_tab[3 * (size_t)(n - 1) + 2] + s->out_tab[3 * (size_t)(n - 1) + 1]);
    int worst = UVS_OK;
    const uvs_report* hr = (const uvs_report*)(s->h_out + nst * 8);
    for (int b = 0; b < n; ++b) if (hr[b].status != UVS_OK) worst = hr[b].status;
    if (reps) std::memcpy(reps, hr, sizeof(uvs_report) * (size_t)n);
    for (int b = 0; states && b < n; ++b) {
        const DevWin& h = s->hdrs[b];
        const double* buf = (const double*)s->h_out + s->out_tab[3 * (size_t)b + 2];
        uvs_state& st = states[b];
        std::memcpy(st.pose, buf, sizeof(double) * 77);
        std::memcpy(st.speedbias, buf + 77, sizeof(double) * 99);
        std::memcpy(st.ex_pose, buf + 176, sizeof(double) * 7);
        st.td = buf[183];
        std::memcpy(st.relo_pose, buf + 184, sizeof(double) * 7);
        if (st.inv_depth) std::memcpy(st.inv_depth, buf + UVS_XDIM, sizeof(double) * h.n_points);
        if (st.line_orth) std::memcpy(st.line_orth, buf + UVS_XDIM + h.n_points, sizeof(double) * 4 * h.n_lines);
    }
    return worst;
}

int uvs_batch_download(uvs_solver* s, int n, uvs_state* states, uvs_report* reps) {
    if (!s || n < 1 || n > s->n_loaded) return UVS_ERR_INVALID_ARG;
    HIPCHK(s, hipSetDevice(s->device));
    const int rc = download_enqueue(s, n);      // (written by the gather kernel instead -- as the stream does -- the call takes as long: 1.370 ms either way for one window, 0.17 ms for 256 states)
    if (rc != UVS_OK) return rc;
    return download_finish(s, n, states, reps);
}

// A STREAM of batches, end to end: packing (host threads), upload, solve and download of consecutive batches overlap.  Three buffer sets -- this handle and two
// twins created on first use with the same options and capacities, each with its own stream, pinned staging and device buffers -- take turns: while the GPU runs
// k_solve -> gather of batch k on one set, the copy engine moves batch k + 1 into the second and the host packs batch k + 2 into the third; a set is drained (wait +
// unpack) right before it is reused.  Results equal uvs_batch_upload / solve / download of each batch (same packing, same kernel).
// What it took to make copy and kernel overlap (round 5; tools/micro_overlap.hip, tools/stream_trace.py, profiles/r05_stream_timeline*.txt, r05_stream_ab.txt):
//   * With TWO sets and the results fetched by a device-to-host copy enqueued behind k_solve, the upload of batch k + 1 -- enqueued on the other stream while k_solve of batch k
//     ran -- did not start until that download had been done, i.e. after the kernel: copy and kernel strictly alternated (95 - 105 k solves/s).  The device itself overlaps
//     them completely (micro_overlap).  Nothing is enqueued behind k_solve any more: k_solve writes every window's final state and report into the pinned result buffer
//     itself (DevWin::out_host, patched into the staged headers by upload_windows; the report pointer of the launch) -- UVS_STREAM_D2H_COPY=2: a gather kernel does, =1: gather
//     kernel + copy, the old form.
//   * three sets, and the kernels of consecutive batches chained by events, see below.
// With these the stream runs at 155 - 170 k solves/s: 1.50 - 1.65 ms per batch beside a kernel of 1.45 ms.
// (An in-kernel prefetch of the next batch -- k_solve's idle wave reading the pinned buffer -- was built before the first point was understood and is slower than the copy
// engine: tools/experiments/r05_stream_prefetch.patch.)
int uvs_batch_stream(uvs_solver* s, int n_batches, int per_batch, const uvs_window* const* ws, uvs_state* states, uvs_report* reps, double* wall_ms) {
    if (!s || n_batches < 1 || per_batch < 1 || !ws) return UVS_ERR_INVALID_ARG;
    if (per_batch > s->max_batch) { s->err = "batch larger than max_batch"; return UVS_ERR_CAPACITY; }
    // THREE buffer sets by default (UVS_STREAM_SETS=2: two): with two, the host can pack batch k only after batch k - 2 has been solved, and pack + copy (1.0 + 0.85 ms) then sit on
    // the critical path of every second kernel (1.72 ms per batch measured); with three the GPU always has a copied batch waiting
    // (the three knobs are read per CALL, not once per process: tools/stream_ab.py alternates the configurations inside one process, on the same windows)
    // UVS_STREAM_SETS=4: a fourth set with at most two kernels in flight (below) -- a batch of slack for a host whose packing threads get descheduled.  Measured against the three-set default
    // in four alternating A/Bs on busy and quiet hosts (profiles/r06_stream_ab_four_sets.txt): medians 172.5 / 173.6 / 150.8 k against 173.5 / 175.8 / 174.8 k, tighter quartiles in one of
    // them, wider in another -- the host's noise decides, not the set count; the default stays three.
    const int NS = [] { const char* e = std::getenv("UVS_STREAM_SETS"); const int v = e ? std::atoi(e) : 3; return v == 2 || v == 4 ? v : 3; }();
    for (uvs_solver** t : {&s->twin, &s->twin2, &s->twin3}) {
        if ((t == &s->twin2 && NS < 3) || (t == &s->twin3 && NS < 4)) break;
        if (!*t) { const int rc = uvs_create(&s->opts, s->device, s->max_batch, s->max_points, s->max_point_obs, s->max_lines, s->max_line_obs, t); if (rc != UVS_OK) { s->err = "uvs_batch_stream: could not create a buffer set"; return rc; } }
        if (!s->pool) s->pool = new PackPool();
        if (!(*t)->pool) { (*t)->pool = s->pool; (*t)->pool_borrowed = true; }      // one pool of packing threads for all sets (they pack one after the other)
    }
    const auto t0 = std::chrono::steady_clock::now();
    uvs_solver* set[4] = {s, s->twin, s->twin2, s->twin3};
    // UVS_STREAM_CHAIN=1: the kernels of consecutive batches chained by events (round 5's default).  Round 6 measured both forms alternately in one process, ten runs of 32 batches each
    // (tools/stream_ab.py, profiles/r06_stream_ab.txt): un-chained 175.4 k solves/s median (quartiles 171.3 - 175.9 k), chained 167.5 k (167.3 - 167.8 k) -- the chain is steadier and
    // 4.5 % slower (a batch's kernel then never starts under the tail of the previous one, whose last workgroups leave compute units idle), so the default is un-chained.
    const bool chain_ = [] { const char* e = std::getenv("UVS_STREAM_CHAIN"); return e && e[0] == '1'; }();
    const int d2h_ = [] { const char* e = std::getenv("UVS_STREAM_D2H_COPY"); return e ? std::atoi(e) : 0; }();      // 0: k_solve writes the results into the pinned buffer; 1: gather kernel + device-to-host copy; 2: the gather kernel writes them
    for (int j = 0; j < NS; ++j) if (!set[j]->ev_done) HIPCHK(s, hipEventCreateWithFlags(&set[j]->ev_done, hipEventDisableTiming));
    int pending[4] = {-1, -1, -1, -1};      // batch index in flight on each set
    // the resident blobs of this call carry addresses into its pinned result buffers (DevWin::out_host): whatever way the call ends, a later uvs_batch_solve needs its own upload
    struct Invalidate { uvs_solver** set; int n; bool on; ~Invalidate() { if (on) for (int j = 0; j < n; ++j) set[j]->n_loaded = 0; } } invalidate_{set, NS, d2h_ == 0};
    int worst = UVS_OK;
    const auto drain = [&](int q) -> int {
        if (pending[q] < 0) return UVS_OK;
        const size_t off = (size_t)pending[q] * per_batch;
        const int rc = download_finish(set[q], per_batch, states ? states + off : nullptr, reps ? reps + off : nullptr);
        pending[q] = -1;
        if (rc != UVS_OK && rc != UVS_ERR_NUMERIC) { if (set[q] != s) s->err = set[q]->err; return rc; }
        if (rc != UVS_OK) worst = rc;
        return UVS_OK;
    };
    static const bool sprof_ = std::getenv("UVS_STREAM_PROFILE") != nullptr;
    for (int k = 0; k < n_batches; ++k) {
        const int q = k % NS;
        const auto td0_ = std::chrono::steady_clock::now();
        int rc = drain(q);
        if (sprof_) fprintf(stderr, "stream batch %d: drain (wait + unpack of batch %d) %.3f ms\n", k, k - NS, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td0_).count());
        if (rc == UVS_OK) rc = upload_windows(set[q], per_batch, ws + (size_t)k * per_batch, false, 0, d2h_ == 0);
        // (UVS_STREAM_CHAIN=1) the kernels run one after the other (an event chain through the sets): two k_solve launches on two streams otherwise share the compute units workgroup by
        // workgroup, both finish late and together, and the host -- which packs batch k + 1 into the set of the batch that finishes first -- stalls and then has two batches to pack in a row
        if (rc == UVS_OK && chain_ && k > 0 && hipStreamWaitEvent(set[q]->stream, set[(k - 1) % NS]->ev_done, 0) != hipSuccess) { s->err = "hipStreamWaitEvent failed"; rc = UVS_ERR_HIP; }
        // four sets, un-chained: at most TWO kernels in flight (batch k waits for batch k - 2), so that the fourth set is slack for the host and not a third kernel sharing the compute units
        const bool chain2_ = !chain_ && NS == 4;
        if (rc == UVS_OK && chain2_ && k > 1 && hipStreamWaitEvent(set[q]->stream, set[(k - 2) % NS]->ev_done, 0) != hipSuccess) { s->err = "hipStreamWaitEvent failed"; rc = UVS_ERR_HIP; }
        if (rc == UVS_OK) rc = launch_solve(set[q], 0, nullptr, false, d2h_ == 0);
        if (rc == UVS_OK && (chain_ || chain2_) && hipEventRecord(set[q]->ev_done, set[q]->stream) != hipSuccess) { s->err = "hipEventRecord failed"; rc = UVS_ERR_HIP; }
        if (rc == UVS_OK && d2h_ != 0) rc = download_enqueue(set[q], per_batch, d2h_ == 2);
        if (rc != UVS_OK) {
            // batch k failed before it was enqueued: the batch still in flight on the OTHER buffer set (k - 1) is delivered like the ones before it, so that on
            // return every batch < k holds results and nothing from k on does; the first error code is the one returned
            if (set[q] != s) s->err = set[q]->err;
            const std::string first_err = s->err;
            for (int j = 1; j < NS; ++j) (void)drain((q + j) % NS);      // (oldest first)
            for (int j = 0; j < NS; ++j) (void)hipStreamSynchronize(set[j]->stream);
            s->err = first_err;
            return rc;
        }
        pending[q] = k;
    }
    for (int j = 0; j < NS; ++j) { const int rc = drain((n_batches + j) % NS); if (rc != UVS_OK) return rc; }      // (oldest first)
    if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return worst;
}

int uvs_solve_window(uvs_solver* s, const uvs_window* w, uvs_state* out, uvs_report* rep) {
    if (!s || !w || !out || !rep) return UVS_ERR_INVALID_ARG;
    const uvs_window* arr[1] = {w};
    // one stream, one wait: pinned upload -> k_solve -> k_pack_outputs -> pinned download (uvs_batch_download synchronizes)
    int rc = upload_windows(s, 1, arr, false);
    if (rc != UVS_OK) return rc;
    rc = launch_solve(s, 0, nullptr, false);
    if (rc != UVS_OK) return rc;
    return uvs_batch_download(s, 1, out, rep);
}

// Diagnostic entry (parity tests): reduced system of the FIRST linearization of window 0 of the uploaded batch.
// S_lower[176*176] row-major (damped, landmark-Schur-reduced, padded index 16*frame+dof), g/hd/dd/step[176], scal[UVS_DEBUG_SCAL_LEN].
int uvs_debug_first_iteration(uvs_solver* s, const uvs_window* w, double* S_lower, double* g, double* hd, double* dd, double* step, double* scal) {
    if (!s || !w) return UVS_ERR_INVALID_ARG;
    const uvs_window* arr[1] = {w};
    int rc = uvs_batch_upload(s, 1, arr);
    if (rc != UVS_OK) return rc;
    const char* tl_path = std::getenv("UVS_DEBUG_LIN_TIMELINE");      // debug: (stamp, clock) log of every wave's steps through the linearizations of this solve, written to this file
    rc = launch_solve(s, tl_path ? 5 : std::getenv("UVS_DEBUG_GATHER_TIMERS") ? 2 : std::getenv("UVS_DEBUG_ASM_TIMERS") ? 3 : std::getenv("UVS_DEBUG_CHOL_TIMELINE") ? 4 : 1, nullptr);      // 2 / 3: the four per-wave timer slots carry the gather / the assembly's sub-steps instead of the Cholesky column phase
    if (rc != UVS_OK) return rc;
    if (tl_path) {
        std::vector<long long> tl(8 * TL_PER_WAVE * 2);
        if ((s->ksolve_nt == 512 ? uvs_k_solve512_timeline(tl.data(), tl.size()) == UVS_OK : hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_lin_tl), tl.size() * 8) == hipSuccess)) { if (FILE* f = std::fopen(tl_path, "wb")) { std::fwrite(tl.data(), 8, tl.size(), f); std::fclose(f); } }
    }
    const size_t nS = (size_t)UVS_RD * UVS_RD;
    if (S_lower) HIPCHK(s, hipMemcpy(S_lower, s->d_dbg, nS * 8, hipMemcpyDeviceToHost));
    if (g) HIPCHK(s, hipMemcpy(g, s->d_dbg + nS, UVS_RD * 8, hipMemcpyDeviceToHost));
    if (hd) HIPCHK(s, hipMemcpy(hd, s->d_dbg + nS + UVS_RD, UVS_RD * 8, hipMemcpyDeviceToHost));
    if (dd) HIPCHK(s, hipMemcpy(dd, s->d_dbg + nS + 2 * UVS_RD, UVS_RD * 8, hipMemcpyDeviceToHost));
    if (step) HIPCHK(s, hipMemcpy(step, s->d_dbg + nS + 3 * UVS_RD, UVS_RD * 8, hipMemcpyDeviceToHost));
    if (scal) HIPCHK(s, hipMemcpy(scal, s->d_dbg + nS + 4 * UVS_RD, UVS_DEBUG_SCAL_LEN * 8, hipMemcpyDeviceToHost));
    return UVS_OK;
}

int uvs_evaluate(uvs_solver* s, const uvs_window* w, int robust, uvs_eval* out) {
    if (!s || !w || !out) return UVS_ERR_INVALID_ARG;
    const uvs_window* arr[1] = {w};
    int rc = uvs_batch_upload(s, 1, arr);
    if (rc != UVS_OK) return rc;
    return run_evaluate(s->device, s->stream, s->d_blobs, s->d_ws, s->hdrs[0], make_kopts(s->opts, 0), robust, out, s->err, s->eval_scratch);
}

// The handle's marginalization worker (uvs_marginalize_resident_begin / uvs_marginalize_wait): ONE thread per handle, created on the first begin and parked on a condition
// variable between jobs.
struct MargWorker {
    std::thread th; std::mutex m; std::condition_variable cv_job, cv_done;
    uvs_solver* s = nullptr; const uvs_window* w = nullptr; int flag = 0, rc = UVS_OK;
    bool has_job = false, done = false, stop = false, in_flight = false;
    void loop() {
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            cv_job.wait(lk, [&] { return stop || has_job; });
            if (stop) return;
            has_job = false;
            const uvs_window* w_ = w; const int f_ = flag;
            lk.unlock();
            const int r = uvs_marginalize_resident(s, w_, f_, &s->marg_job_out);
            lk.lock();
            rc = r; done = true;
            cv_done.notify_all();
        }
    }
};
static bool marg_in_flight(const uvs_solver* s) { return s->marg_worker && s->marg_worker->in_flight; }      // (only the caller's thread reads / writes in_flight)
static int marg_worker_begin(uvs_solver* s, const uvs_window* w, int flag) {
    if (!s->marg_worker) {
        MargWorker* mw = new MargWorker(); mw->s = s;
        try { mw->th = std::thread([mw] { mw->loop(); }); }
        catch (const std::exception& e) {      // (std::system_error when no thread can be created: nothing may cross the C boundary)
            delete mw; s->err = std::string("uvs_marginalize_resident_begin: could not start the worker thread: ") + e.what();
            return UVS_ERR_HIP;
        }
        s->marg_worker = mw;
    }
    MargWorker& mw = *s->marg_worker;
    { std::lock_guard<std::mutex> lk(mw.m); mw.w = w; mw.flag = flag; mw.has_job = true; mw.done = false; }
    mw.in_flight = true;
    mw.cv_job.notify_one();
    return UVS_OK;
}
static int marg_worker_wait(uvs_solver* s) {
    MargWorker& mw = *s->marg_worker;
    std::unique_lock<std::mutex> lk(mw.m);
    mw.cv_done.wait(lk, [&] { return mw.done; });
    mw.done = false; mw.in_flight = false;
    return mw.rc;
}
static void free_marg_worker(MargWorker* mw) {
    if (!mw) return;
    if (mw->in_flight) { std::unique_lock<std::mutex> lk(mw->m); mw->cv_done.wait(lk, [&] { return mw->done; }); }
    { std::lock_guard<std::mutex> lk(mw->m); mw->stop = true; }
    mw->cv_job.notify_one();
    if (mw->th.joinable()) mw->th.join();
    delete mw;
}

// MARGIN_SECOND_NEW (estimator.cpp:1159-1228) marginalizes Pose[WINDOW_SIZE - 1] out of the OLD PRIOR and reads nothing else: no factor is evaluated, so no kernel runs and nothing is
// copied -- r = r0 + J0 dx, A = J0^T J0, b = J0^T r, the 6 x 6 elimination and the n x n factorization are host work (uvs_marg.h).  Round 5 packed and uploaded the window and
// evaluated it on the device to obtain that one vector r (0.2 ms of the 0.5 ms a call took).
static int marginalize_second_new_host(uvs_solver* s, const uvs_window* w, uvs_prior* out) {
    { const int rv = validate_window(w, s->err); if (rv != UVS_OK) return rv; }
    // (the same complaint the packing of the window made when this path still uploaded it)
    if (s->opts.estimate_td != 0 && w->n_point_obs > 0 && (!w->pt_vel_i || !w->pt_vel_j || !w->pt_td_i || !w->pt_td_j)) { s->err = "estimate_td needs pt_vel_i / pt_vel_j / pt_td_i / pt_td_j"; return UVS_ERR_INVALID_ARG; }
    DevWin h; std::memset(&h, 0, sizeof(h)); h.td_on = s->opts.estimate_td != 0;
    return run_marginalize(s->device, s->stream, nullptr, nullptr, h, w, make_kopts(s->opts, 0), 1, out, s->err, s->eval_scratch);
}

int uvs_marginalize(uvs_solver* s, const uvs_window* w, int flag, uvs_prior* out) {
    if (!s || !w || !out || (flag != 0 && flag != 1)) return UVS_ERR_INVALID_ARG;
    if (flag == 1) return marginalize_second_new_host(s, w, out);
    if (flag == 0) { const int rd = marginalize_old_device(s, w, out); if (rd != kMargFallback) return rd; }
    const uvs_window* arr[1] = {w};
    const auto tu0 = std::chrono::steady_clock::now();
    int rc = uvs_batch_upload(s, 1, arr);
    if (rc != UVS_OK) return rc;
    if (std::getenv("UVS_MARG_PROFILE")) std::fprintf(stderr, "[uvs_marginalize] upload %.0f us\n", (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tu0).count() * 1e-3);
    return run_marginalize(s->device, s->stream, s->d_blobs, s->d_ws, s->hdrs[0], w, make_kopts(s->opts, 0), flag, out, s->err, s->eval_scratch);
}


int uvs_marginalize_resident(uvs_solver* s, const uvs_window* w, int flag, uvs_prior* out) {
    if (!s || !w || !out || (flag != 0 && flag != 1)) return UVS_ERR_INVALID_ARG;
    if (s->n_loaded != 1) { s->err = "uvs_marginalize_resident: no single resident window"; return UVS_ERR_INVALID_ARG; }
    const DevWin& h = s->hdrs[0];
    const int pn = (w->prior && w->prior->n > 0) ? w->prior->n : 0;
    if (h.n_points != w->n_points || h.n_pt_obs - h.n_relo != w->n_point_obs || h.n_lines != w->n_lines || h.n_ln_obs != w->n_line_obs || h.n_imu != w->n_imu || h.prior_n != pn) {
        s->err = "uvs_marginalize_resident: the window does not match the resident one"; return UVS_ERR_INVALID_ARG;
    }
    if (flag == 1) return marginalize_second_new_host(s, w, out);      // (reads the old prior only: host work, no device round trip)
    if (flag == 0) { const int rd = marginalize_old_device(s, w, out); if (rd != kMargFallback) return rd; }      // (needs nothing of the resident blob: the factors of frame 0 travel as a window of their own)
    HIPCHK(s, hipSetDevice(s->device));
    // state sections of the resident blob: frames[184] = pose | speedbias | ex_pose | td, inverse depths, line parameters
    // staged in the pinned upload buffer (copies from the caller's pageable arrays would each be a synchronous staging round trip)
    const size_t nst = 184 + (size_t)h.n_points + 4 * (size_t)h.n_lines;
    HIPCHK(s, hipStreamSynchronize(s->stream));      // the staging buffer may still feed an earlier copy
    int rcp;
    if ((rcp = ensure_pinned(s, &s->h_up, &s->h_up_cap, nst * 8)) != UVS_OK) return rcp;
    double* fr = (double*)s->h_up;
    std::memcpy(fr, w->pose, 77 * 8); std::memcpy(fr + 77, w->speedbias, 99 * 8); std::memcpy(fr + 176, w->ex_pose, 7 * 8); fr[183] = w->td;
    if (h.n_points) std::memcpy(fr + 184, w->inv_depth, (size_t)h.n_points * 8);
    if (h.n_lines) std::memcpy(fr + 184 + h.n_points, w->line_orth, (size_t)h.n_lines * 32);
    char* blob = s->d_blobs + s->blob_off[0];
    HIPCHK(s, hipMemcpyAsync(blob + (size_t)h.d_frames * 8, fr, 184 * 8, hipMemcpyHostToDevice, s->stream));
    if (h.n_points) HIPCHK(s, hipMemcpyAsync(blob + (size_t)h.d_invd * 8, fr + 184, (size_t)h.n_points * 8, hipMemcpyHostToDevice, s->stream));
    if (h.n_lines) HIPCHK(s, hipMemcpyAsync(blob + (size_t)h.d_line * 8, fr + 184 + h.n_points, (size_t)h.n_lines * 32, hipMemcpyHostToDevice, s->stream));
    return run_marginalize(s->device, s->stream, s->d_blobs, s->d_ws, s->hdrs[0], w, make_kopts(s->opts, 0), flag, out, s->err, s->eval_scratch);
}

int uvs_marginalize_resident_begin(uvs_solver* s, const uvs_window* w, int flag) {
    if (!s || !w || (flag != 0 && flag != 1)) return UVS_ERR_INVALID_ARG;
    if (marg_in_flight(s)) { s->err = "uvs_marginalize_resident_begin: the previous marginalization has not been waited for"; return UVS_ERR_INVALID_ARG; }
    // the worker owns the handle until uvs_marginalize_wait(): device selection is per thread, everything else (stream, pinned buffers, scratch) is the handle's own
    return marg_worker_begin(s, w, flag);
}
int uvs_marginalize_wait(uvs_solver* s, uvs_prior* out) {
    if (!s || !out) return UVS_ERR_INVALID_ARG;
    if (!marg_in_flight(s)) { s->err = "uvs_marginalize_wait: no marginalization in flight"; return UVS_ERR_INVALID_ARG; }
    const int rc = marg_worker_wait(s);
    if (rc == UVS_OK) *out = s->marg_job_out;
    return rc;
}

}  // extern "C"


// ---------------------------------------------------------------- marginalization of a BATCH of windows (round 6, ABI v7)
// Per window the same result as uvs_marginalize(), with everything that is O(n^3) on the device for all windows at once: the sub-windows of the MARGIN_OLD windows are packed by the
// handle's packing threads and linearized by ONE launch (k_marg_linearize_batch: assembly + elimination of the dropped landmarks), the dropped frame block, the Schur complement and the
// n x n eigen-decomposition of every window run in ONE launch of k_marg_finish (uvs_marg_kernel.h: parallel cyclic Jacobi).  MARGIN_SECOND_NEW windows send their prior-only system
// (assembled on the packing threads) to the same kernel.  A window the device path does not take (a landmark or frame block the reference's eps cut would touch, a system larger than
// the kernel's LDS layout, no factors at all) goes through uvs_marginalize() on the calling thread.
struct MargBatchBuf {
    char* h_stage = nullptr; size_t h_stage_cap = 0;      // pinned: blobs | tables | descriptors | dense systems
    char* h_out = nullptr; size_t h_out_cap = 0;          // pinned: finish outputs | linearization scalars
    char* d_blobs = nullptr; size_t d_blobs_cap = 0; double* d_ws = nullptr; size_t d_ws_cap = 0; double* d_lin = nullptr; size_t d_lin_cap = 0;
    char* d_tab = nullptr; size_t d_tab_cap = 0; double* d_in = nullptr; size_t d_in_cap = 0; double* d_out = nullptr; size_t d_out_cap = 0;
    std::vector<MargDevScratch> thread_sub; std::vector<EvalScratch> thread_eval;
};
static void free_marg_batch(MargBatchBuf* m) {
    if (!m) return;
    if (m->h_stage) (void)hipHostFree(m->h_stage); if (m->h_out) (void)hipHostFree(m->h_out);
    for (void* p : {(void*)m->d_blobs, (void*)m->d_ws, (void*)m->d_lin, (void*)m->d_tab, (void*)m->d_in, (void*)m->d_out}) if (p) (void)hipFree(p);
    delete m;
}
namespace {
struct MargBatchItem {
    int path = 3;      // 0: *out is final already; 1: device linearization + device finish (MARGIN_OLD); 2: device finish of a host-assembled system (MARGIN_SECOND_NEW); 3: uvs_marginalize()
    int rc = UVS_OK; std::string err;
    bool used[24]; std::vector<int> pos, keep_ids, map; int md = 0, n = 0;
    std::vector<char> blob; DevWin h;
    std::vector<double> dense;      // path 2: A [N][N] | b [N]
};
}
extern "C" int uvs_marginalize_batch(uvs_solver* s, int n_win, const uvs_window* const* ws, const int* flags, uvs_prior* out, int* status) {
    using namespace uvsmarg;
    if (!s || n_win < 0 || (n_win > 0 && (!ws || !flags || !out))) return UVS_ERR_INVALID_ARG;
    for (int b = 0; b < n_win; ++b) if (!ws[b] || (flags[b] != 0 && flags[b] != 1)) { s->err = "uvs_marginalize_batch: null window or flag outside {0, 1}"; return UVS_ERR_INVALID_ARG; }
    if (marg_in_flight(s)) { s->err = "uvs_marginalize_batch: a marginalization begun with uvs_marginalize_resident_begin has not been waited for"; return UVS_ERR_INVALID_ARG; }
    if (n_win == 0) return UVS_OK;
    HIPCHK(s, hipSetDevice(s->device));
    if (!s->marg_batch) s->marg_batch = new MargBatchBuf();
    MargBatchBuf& B = *s->marg_batch;
    const bool prof = std::getenv("UVS_MARG_PROFILE") != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    auto tb1 = tb0, tb2 = tb0, tb3 = tb0;
    int nthreads = 1;
    if (n_win >= 4) {
        const char* env = std::getenv("UVS_PACK_THREADS");
        nthreads = env ? std::atoi(env) : (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency() / 2));
        nthreads = std::max(1, std::min(nthreads, n_win));
    }
    if ((int)B.thread_sub.size() < nthreads) { B.thread_sub.resize(nthreads); B.thread_eval.resize(nthreads); }
    std::vector<MargBatchItem> items((size_t)n_win);
    uvs_options o_sub = s->opts; o_sub.estimate_extrinsic = 1; o_sub.initial_trust_region_radius = 1e300;      // (as marginalize_old_device: the prior keeps para_Ex_Pose)
    const bool host_only = std::getenv("UVS_MARG_HOST") != nullptr;
    // ---- host stage, per window, on the packing threads
    const auto job = [&](int t) {
        for (int b = t; b < n_win; b += nthreads) {
            MargBatchItem& it = items[b]; const uvs_window* w = ws[b];
            it.path = 3;
            if (host_only) continue;
            if (flags[b] == 0) {
                uvs_window sub;
                const int rb = marg_build_sub(s, w, B.thread_sub[t], it.used, sub, it.err);
                if (rb == kMargFallback) continue;
                if (rb != UVS_OK) { it.rc = rb; it.path = 0; continue; }
                const int rp = pack_window(&sub, o_sub, it.blob, it.h, it.err);
                if (rp == UVS_ERR_UNSUPPORTED || rp == UVS_ERR_CAPACITY) continue;
                if (rp != UVS_OK) { it.rc = rp; it.path = 0; continue; }
                marg_frame_order(it.used, it.pos, it.keep_ids, it.md, it.n, it.map);
                if (it.n > UVS_MAX_PRIOR_DIM || (int)it.keep_ids.size() > UVS_MAX_PRIOR_BLOCKS) { it.err = "prior capacity"; it.rc = UVS_ERR_CAPACITY; it.path = 0; continue; }
                if (it.md == 0 || it.n == 0 || it.md > MF_MD || it.n > MF_NKEEP || it.md + it.n > MF_NMAX) continue;
                it.path = 1;
            } else {
                { const int rv = validate_window(w, it.err); if (rv != UVS_OK) { it.rc = rv; it.path = 0; continue; } }
                if (s->opts.estimate_td != 0 && w->n_point_obs > 0 && (!w->pt_vel_i || !w->pt_vel_j || !w->pt_td_i || !w->pt_td_j)) { it.err = "estimate_td needs pt_vel_i / pt_vel_j / pt_td_i / pt_td_j"; it.rc = UVS_ERR_INVALID_ARG; it.path = 0; continue; }
                DevWin h; std::memset(&h, 0, sizeof(h)); h.td_on = s->opts.estimate_td != 0;
                MargSystem ms; bool done = false;
                EvalScratch& sc = B.thread_eval[t];
                const int ra = marg_assemble_host(s->device, s->stream, nullptr, nullptr, h, w, make_kopts(s->opts, 0), 1, &out[b], it.err, sc, ms, done);
                if (ra != UVS_OK || done) { it.rc = ra; it.path = 0; continue; }
                if (ms.m != ms.md || ms.md > MF_MD || ms.n > MF_NKEEP || ms.N > MF_NMAX || ms.md == 0 || ms.n == 0) continue;      // (never for MARGIN_SECOND_NEW: it drops one pose and no landmark)
                it.md = ms.md; it.n = ms.n; it.pos = ms.pos; it.keep_ids = ms.keep_ids;
                it.dense.assign(sc.work[0].begin(), sc.work[0].begin() + (size_t)ms.N * ms.N);
                it.dense.insert(it.dense.end(), sc.work[1].begin(), sc.work[1].begin() + ms.N);
                it.path = 2;
            }
        }
    };
    if (nthreads > 1) { if (!s->pool) s->pool = new PackPool(); if (s->pool->ensure(nthreads)) s->pool->run(nthreads, job); else { const int nt_ = nthreads; nthreads = 1; job(0); nthreads = nt_; } }
    else job(0);
    tb1 = std::chrono::steady_clock::now();
    // ---- device stage: finish slots = the path-1 windows (their linearization slots), then the path-2 windows
    std::vector<int> slot_win; slot_win.reserve(n_win);
    for (int b = 0; b < n_win; ++b) if (items[b].path == 1) slot_win.push_back(b);
    const int n1 = (int)slot_win.size();
    for (int b = 0; b < n_win; ++b) if (items[b].path == 2) slot_win.push_back(b);
    const int nfin = (int)slot_win.size();
    const int n1_prof = n1, nfin_prof = nfin;
    if (nfin > 0) {
        // staging layout: [blobs (8-byte aligned each)] [blob_off n1][ws_off n1] [desc nfin x MF_DESC ints] [dense (nfin - n1) x MF_IN doubles]
        std::vector<long long> blob_off(std::max(n1, 1)), ws_off(std::max(n1, 1));
        size_t blob_total = 0; long long ws_total = 0;
        for (int q = 0; q < n1; ++q) { const MargBatchItem& it = items[slot_win[q]]; blob_off[q] = (long long)blob_total; blob_total += (it.blob.size() + 255) & ~(size_t)255; ws_off[q] = ws_total; ws_total += it.h.ws_doubles; }
        const size_t tab_bytes = (size_t)n1 * 16 + (size_t)nfin * MF_DESC * 4, dense_bytes = (size_t)(nfin - n1) * MF_IN * 8;
        int rc;
        if ((rc = ensure_pinned(s, &B.h_stage, &B.h_stage_cap, blob_total + tab_bytes + dense_bytes + 64)) != UVS_OK) return rc;
        if ((rc = ensure_pinned(s, &B.h_out, &B.h_out_cap, (size_t)nfin * MF_OUT * 8 + (size_t)std::max(n1, 1) * 64)) != UVS_OK) return rc;
        if ((rc = ensure(s, (void**)&B.d_blobs, &B.d_blobs_cap, std::max<size_t>(blob_total, 256))) != UVS_OK) return rc;
        if ((rc = ensure(s, (void**)&B.d_ws, &B.d_ws_cap, std::max<size_t>((size_t)ws_total * 8, 256))) != UVS_OK) return rc;
        if ((rc = ensure(s, (void**)&B.d_lin, &B.d_lin_cap, (size_t)std::max(n1, 1) * MARG_OUT * 8)) != UVS_OK) return rc;
        if ((rc = ensure(s, (void**)&B.d_tab, &B.d_tab_cap, tab_bytes + 64)) != UVS_OK) return rc;
        if ((rc = ensure(s, (void**)&B.d_in, &B.d_in_cap, std::max<size_t>(dense_bytes, 256))) != UVS_OK) return rc;
        if ((rc = ensure(s, (void**)&B.d_out, &B.d_out_cap, (size_t)nfin * MF_OUT * 8)) != UVS_OK) return rc;
        HIPCHK(s, hipStreamSynchronize(s->stream));      // the staging buffer may still feed an earlier call's copies
        char* hb = B.h_stage; char* ht = hb + blob_total; char* hd = ht + ((tab_bytes + 7) & ~(size_t)7);
        for (int q = 0; q < n1; ++q) { const MargBatchItem& it = items[slot_win[q]]; std::memcpy(hb + blob_off[q], it.blob.data(), it.blob.size()); }
        long long* t_off = (long long*)ht; int* t_desc = (int*)(ht + (size_t)n1 * 16);
        for (int q = 0; q < n1; ++q) { t_off[q] = blob_off[q]; t_off[n1 + q] = ws_off[q]; }
        for (int q = 0; q < nfin; ++q) {
            const MargBatchItem& it = items[slot_win[q]]; int* d = t_desc + (size_t)q * MF_DESC;
            std::memset(d, 0, MF_DESC * 4);
            d[0] = it.md + it.n; d[1] = it.md; d[2] = it.n; d[3] = q < n1 ? 0 : 1;
            if (q < n1) for (int i = 0; i < it.md + it.n; ++i) d[4 + i] = it.map[i];
            else std::memcpy(hd + (size_t)(q - n1) * MF_IN * 8, it.dense.data(), it.dense.size() * 8);
        }
        if (n1 > 0) HIPCHK(s, hipMemcpyAsync(B.d_blobs, hb, blob_total, hipMemcpyHostToDevice, s->stream));
        HIPCHK(s, hipMemcpyAsync(B.d_tab, ht, tab_bytes, hipMemcpyHostToDevice, s->stream));
        if (nfin > n1) {      // (likewise only the used head N^2 + N of every dense input slot)
            int N_max = 1; for (int q = n1; q < nfin; ++q) N_max = std::max(N_max, items[slot_win[q]].md + items[slot_win[q]].n);
            HIPCHK(s, hipMemcpy2DAsync(B.d_in, (size_t)MF_IN * 8, hd, (size_t)MF_IN * 8, (size_t)(N_max * N_max + N_max) * 8, (size_t)(nfin - n1), hipMemcpyHostToDevice, s->stream));
        }
        const KOpts ko = make_kopts(o_sub, 0);
        if (n1 > 0) {
            hipLaunchKernelGGL(k_marg_linearize_batch, dim3(n1), dim3(NT), LDS_BYTES, s->stream, B.d_blobs, (const long long*)B.d_tab, B.d_ws, (const long long*)B.d_tab + n1, ko, B.d_lin);
            HIPCHK(s, hipGetLastError());
        }
        // (path-2 slots read their dense system at slot - n1: the pointer is shifted so that the kernel's `in_all + MF_IN * blockIdx.x` lands there)
        hipLaunchKernelGGL(k_marg_finish, dim3(nfin), dim3(MF_NT), MF_LDS_BYTES, s->stream, (const int*)(B.d_tab + (size_t)n1 * 16), (const double*)B.d_in - (size_t)n1 * MF_IN, (const double*)B.d_lin, (int)MARG_OUT,
                           (int)UVS_RD, B.d_out, 1e-8);
        HIPCHK(s, hipGetLastError());
        int n_max = 1; for (int q = 0; q < nfin; ++q) n_max = std::max(n_max, items[slot_win[q]].n);
        // (only the used head of every output slot travels: status | r0 | J0 [n][n])
        HIPCHK(s, hipMemcpy2DAsync(B.h_out, (size_t)MF_OUT * 8, B.d_out, (size_t)MF_OUT * 8, (size_t)(MF_OUT_J + n_max * n_max) * 8, (size_t)nfin, hipMemcpyDeviceToHost, s->stream));
        double* h_scal = (double*)(B.h_out + (size_t)nfin * MF_OUT * 8);
        if (n1 > 0) HIPCHK(s, hipMemcpy2DAsync(h_scal, 64, B.d_lin + (MARG_OUT - 8), (size_t)MARG_OUT * 8, 64, (size_t)n1, hipMemcpyDeviceToHost, s->stream));
        tb2 = std::chrono::steady_clock::now();
        HIPCHK(s, hipStreamSynchronize(s->stream));
        tb3 = std::chrono::steady_clock::now();
        if (prof) { double sw = 0, swmax = 0, rot = 0, cut = 0, cy[3] = {0, 0, 0}; for (int q = 0; q < nfin; ++q) { const double* fo = (const double*)B.h_out + (size_t)q * MF_OUT; sw += fo[MF_OUT_S + 1]; swmax = std::max(swmax, fo[MF_OUT_S + 1]); rot += fo[MF_OUT_S + 2]; cut += fo[MF_OUT_S + 3]; for (int k = 0; k < 3; ++k) cy[k] += fo[MF_OUT_S + 4 + k]; }
                    std::fprintf(stderr, "[uvs_marginalize_batch] k_marg_finish: %.1f Jacobi sweeps (most: %.0f), %.0f rotations, %.1f eigenvalues cut per window (mean over %d); shader-clock cycles per window: first rotation parameters of the sweeps %.0f k, A passes %.0f k, V passes (beside the next step's parameters) %.0f k\n",
                                 sw / nfin, swmax, rot / nfin, cut / nfin, nfin, cy[0] / nfin * 1e-3, cy[1] / nfin * 1e-3, cy[2] / nfin * 1e-3); }
        for (int q = 0; q < nfin; ++q) {
            const int b = slot_win[q]; MargBatchItem& it = items[b];
            const double* fo = (const double*)B.h_out + (size_t)q * MF_OUT;
            const int st = (int)fo[MF_OUT_S];
            if (q < n1 && (h_scal[8 * q + 1] != 0.0 || !std::isfinite(h_scal[8 * q]))) { it.path = 3; continue; }      // a landmark block the reference's eps cut would touch: the host path applies that cut
            if (st == MF_NONFINITE) { it.rc = UVS_ERR_NUMERIC; it.err = "marginalization: the linearized system is not finite"; it.path = 0; continue; }
            if (st != MF_OK) { it.path = 3; continue; }
            uvs_prior* po = &out[b];
            std::memset(po, 0, sizeof(*po));
            po->n = it.n;
            std::memcpy(po->linearized_jacobians, fo + MF_OUT_J, (size_t)it.n * it.n * 8);
            std::memcpy(po->linearized_residuals, fo + MF_OUT_R, (size_t)it.n * 8);
            marg_fill_blocks(po, it.pos, it.keep_ids, it.md, ws[b], flags[b]);
            it.path = 0;
        }
    }
    // ---- the windows the device path did not take
    int first_bad = UVS_OK;
    for (int b = 0; b < n_win; ++b) {
        MargBatchItem& it = items[b];
        if (it.path == 3) { it.rc = uvs_marginalize(s, ws[b], flags[b], &out[b]); if (it.rc != UVS_OK) it.err = s->err; }
        if (status) status[b] = it.rc;
        if (it.rc != UVS_OK && first_bad == UVS_OK) { first_bad = it.rc; s->err = it.err; }
    }
    if (prof) {
        auto us = [](auto a_, auto b_) { return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(b_ - a_).count() * 1e-3; };
        int n_fb = 0; for (int b = 0; b < n_win; ++b) n_fb += items[b].path == 3 ? 1 : 0;
        std::fprintf(stderr, "[uvs_marginalize_batch] %d windows on %d threads: host stage (sub-windows, packing, prior-only systems) %.0f us, staging + enqueue %.0f us, device (copies, k_marg_linearize_batch x %d, k_marg_finish x %d) %.0f us, priors + one-window fallbacks (%d) %.0f us\n",
                     n_win, nthreads, us(tb0, tb1), us(tb1, tb2), us(tb2, tb3), n1_prof, nfin_prof, n_fb, us(tb3, std::chrono::steady_clock::now()));
    }
    return first_bad;
}

// ------------------------------------------------------------------ large single window (configs[3]), optionally multi-GPU
// Step-wise so that the caller can all-reduce the two device vectors between steps (RCCL through torch.distributed in
// bench.py / api.py; nothing to reduce on one GPU):
//   uvs_large_begin -> loop { uvs_large_linearize -> [all-reduce SUM of uvs_large_reduced()] -> uvs_large_step
//                             -> [all-reduce SUM of uvs_large_scalars()] -> uvs_large_decide } -> uvs_large_finish
extern "C" {

int uvs_large_begin(uvs_solver* s, const uvs_window* w) {
    if (!s || !w) return UVS_ERR_INVALID_ARG;
    const uvs_window* arr[1] = {w};
    int rc = upload_windows(s, 1, arr, true, s->chunk_wgs());
    if (rc != UVS_OK) return rc;
    auto& L = s->L; const DevWin& h = s->hdrs[0];
    double* keep_ctl = L.d_ctl; uvs_report* keep_rep = L.d_rep; void* keep_comm = L.comm; const int keep_rank = L.rank, keep_nranks = L.nranks; double* keep_fimg = L.d_fimg;
    L = uvs_solver::Large{L.active, 0, 0, 0, 0, 0, 0, 0, 0, true, true, false, 0, 2, 0, 0, 0, 0, L.d_state, L.d_partials, L.d_reduced, L.d_bsums, L.d_out, L.d_sc5, L.cap_partials, L.cap_bsums, {}};
    L.active = true; L.n_chunks = h.n_chunks; L.radius = s->opts.initial_trust_region_radius;
    L.grid = std::min(h.n_chunks, s->chunk_wgs());
    L.d_ctl = keep_ctl; L.d_rep = keep_rep; L.comm = keep_comm; L.rank = keep_rank; L.nranks = keep_nranks; L.d_fimg = keep_fimg;
    if (!L.d_state) { HIPCHK(s, hipMalloc((void**)&L.d_state, LG_STATE * 8)); HIPCHK(s, hipMalloc((void**)&L.d_reduced, LG_XCH_ALL * 8)); HIPCHK(s, hipMemset(L.d_reduced, 0, LG_XCH_ALL * 8)); HIPCHK(s, hipMalloc((void**)&L.d_out, 64 * 8)); HIPCHK(s, hipMalloc((void**)&L.d_sc5, 8 * 8)); }
    if (!L.d_fimg) HIPCHK(s, hipMalloc((void**)&L.d_fimg, LG_FIMG * 8));
    int r2;
    if ((r2 = ensure(s, (void**)&L.d_partials, &L.cap_partials, (size_t)std::max(L.grid, 1) * LG_ROW * 8)) != UVS_OK) return r2;
    if ((r2 = ensure(s, (void**)&L.d_bsums, &L.cap_bsums, (size_t)std::max(L.n_chunks, 1) * 8 * 8)) != UVS_OK) return r2;
    HIPCHK(s, hipMemsetAsync(L.d_state, 0, LG_STATE * 8, s->stream));
    // frames -> state.X ; landmark parameters -> workspace buffer 0 (device-to-device from the blob)
    HIPCHK(s, hipMemcpyAsync(L.d_state + LS_X, s->d_blobs + (size_t)h.d_frames * 8, UVS_XDIM * 8, hipMemcpyDeviceToDevice, s->stream));
    if (h.n_points) HIPCHK(s, hipMemcpyAsync(s->d_ws + h.w_invd0, s->d_blobs + (size_t)h.d_invd * 8, (size_t)h.n_points * 8, hipMemcpyDeviceToDevice, s->stream));
    if (h.n_lines) HIPCHK(s, hipMemcpyAsync(s->d_ws + h.w_line0, s->d_blobs + (size_t)h.d_line * 8, (size_t)h.n_lines * 32, hipMemcpyDeviceToDevice, s->stream));
    double x2 = 0.0;
    for (int f = 0; f < UVS_NUM_FRAMES; ++f) { for (int k = 0; k < 7; ++k) x2 += w->pose[f][k] * w->pose[f][k]; for (int k = 0; k < 9; ++k) x2 += w->speedbias[f][k] * w->speedbias[f][k]; }
    if (s->opts.estimate_td) x2 += w->td * w->td;
    if (s->opts.estimate_extrinsic) for (int k = 0; k < 7; ++k) x2 += w->ex_pose[k] * w->ex_pose[k];
    if (w->n_relo_obs > 0) for (int k = 0; k < 7; ++k) x2 += w->relo_pose[k] * w->relo_pose[k];      // relo_Pose is a free block of the problem (estimator.cpp:947)
    double l2 = 0.0;
    for (int k = 0; k < w->n_points; ++k) l2 += w->inv_depth[k] * w->inv_depth[k];
    for (int k = 0; k < 4 * w->n_lines; ++k) l2 += w->line_orth[k] * w->line_orth[k];
    L.local_x2 = l2; L.x_norm = std::sqrt(x2 + l2); L.frame_x2 = x2;
    std::memset(&L.rep, 0, sizeof(L.rep));
    std::memcpy(L.relo_pose_in, w->relo_pose, sizeof(L.relo_pose_in));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    L.t_begin = std::chrono::steady_clock::now();
    return UVS_OK;
}

// landmark part of ||x||^2 of THIS rank (sum over ranks + frames gives Ceres' x_norm^2); set the global value with uvs_large_set_landmark_x2
double uvs_large_local_x2(const uvs_solver* s) { return s ? s->L.local_x2 : 0.0; }
void uvs_large_set_landmark_x2(uvs_solver* s, double all_ranks_x2) { if (s) { auto& L = s->L; L.x_norm = std::sqrt(L.x_norm * L.x_norm - L.local_x2 + all_ranks_x2); L.local_x2 = all_ranks_x2; } }

int uvs_large_need_linearize(const uvs_solver* s) { return s && s->L.active && !s->L.done && s->L.need_lin; }
int uvs_large_done(const uvs_solver* s) { return !s || !s->L.active || s->L.done; }
double* uvs_large_reduced(uvs_solver* s, int* n) { if (n) *n = LG_RED; return s ? s->L.d_reduced : nullptr; }     // DEVICE pointer; [LG_ACC+1] is a MAX entry
double* uvs_large_scalars(uvs_solver* s, int* n) { if (n) *n = 6; return s ? s->L.d_sc5 : nullptr; }             // DEVICE pointer; [5] = this rank's "time is up" vote (SUM over ranks > 0 ends the solve on every rank)

// host-staged access to the two exchange vectors (which = 0: reduced[LG_RED], 1: scalars[5]); set != 0 writes host -> device
int uvs_large_exchange_host(uvs_solver* s, int which, double* buf, int set) {
    if (!s || !s->L.active || !buf) return UVS_ERR_INVALID_ARG;
    double* d = which == 0 ? s->L.d_reduced : s->L.d_sc5; const size_t n = which == 0 ? LG_RED : 6;
    HIPCHK(s, hipSetDevice(s->device));
    if (set) HIPCHK(s, hipMemcpy(d, buf, n * 8, hipMemcpyHostToDevice)); else HIPCHK(s, hipMemcpy(buf, d, n * 8, hipMemcpyDeviceToHost));
    return UVS_OK;
}

int uvs_large_linearize(uvs_solver* s) {
    if (!s || !s->L.active) return UVS_ERR_INVALID_ARG;
    auto& L = s->L;
    HIPCHK(s, hipSetDevice(s->device));
    KOpts ko = make_kopts(s->opts, 0);
    if (s->large_chunks_nt == 512) { if (uvs_k_large_chunks512_launch(L.grid + 1, s->stream, s->d_blobs, s->d_ws, &ko, sizeof(ko), L.d_state, L.sel, L.first ? 1 : 0, L.radius, L.d_partials, nullptr, 0, 0, L.grid, L.d_fimg) != UVS_OK) { s->err = "k_large_chunks (512 threads): argument layout mismatch"; return UVS_ERR_HIP; } }
    else hipLaunchKernelGGL(k_large_chunks, dim3(L.grid + 1), dim3(NT), LDS_BYTES, s->stream, s->d_blobs, s->d_ws, ko, L.d_state, L.sel, L.first ? 1 : 0, L.radius, L.d_partials, LargeCtl{nullptr, 0, 0}, L.grid, L.d_fimg);
    { const int n_ent = s->hdrs[0].relo2 ? LG_ROW : LG_RED; hipLaunchKernelGGL(k_large_reduce, dim3((n_ent + 15) / 16), dim3(256), 0, s->stream, L.d_partials, L.grid, L.d_reduced, LargeCtl{nullptr, 0, 0}, n_ent); }
    HIPCHK(s, hipGetLastError());
    HIPCHK(s, hipStreamSynchronize(s->stream));
    return UVS_OK;
}

int uvs_large_step(uvs_solver* s) {
    if (!s || !s->L.active) return UVS_ERR_INVALID_ARG;
    auto& L = s->L;
    HIPCHK(s, hipSetDevice(s->device));
    KOpts ko = make_kopts(s->opts, 0);
    if (s->large_solve_nt == 512) { if (uvs_k_large_solve512_launch(s->stream, s->d_blobs, s->d_ws, &ko, sizeof(ko), L.d_state, L.d_reduced, L.first ? 1 : 0, L.radius, L.d_out, nullptr, 0, 0, L.d_fimg) != UVS_OK) { s->err = "k_large_solve (512 threads): argument layout mismatch"; return UVS_ERR_HIP; } }
    else hipLaunchKernelGGL(k_large_solve, dim3(1), dim3(NT), LDS_BYTES, s->stream, s->d_blobs, s->d_ws, ko, L.d_state, L.d_reduced, L.first ? 1 : 0, L.radius, L.d_out, LargeCtl{nullptr, 0, 0}, L.d_fimg);
    { const int bg = std::min(L.n_chunks, UVS_LARGE_OCC * s->chunk_wgs());      // (UVS_LARGE_OCC workgroups per compute unit: the kernel asks for little LDS and half the registers)
      hipLaunchKernelGGL(k_large_backsub, dim3(bg + 1), dim3(NT), LDS_BYTES_BACKSUB, s->stream, s->d_blobs, s->d_ws, ko, L.d_state, L.sel, L.d_bsums, LargeCtl{nullptr, 0, 0}, bg, L.d_out); }
    hipLaunchKernelGGL(k_large_sum_bsums, dim3(1), dim3(256), 0, s->stream, L.d_bsums, L.n_chunks, L.d_sc5, LargeCtl{nullptr, 0, 0}, 0LL);
    HIPCHK(s, hipGetLastError());
    HIPCHK(s, hipStreamSynchronize(s->stream));
    // options.max_solver_time_in_seconds on the host-driven loop: this process's vote travels as scalar [5], so that ranks which all-reduce the scalars decide together
    const uvs_options& o = s->opts;
    const double vote = (o.max_solver_time_in_seconds > 0.0 && L.it > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - L.t_begin).count() >= o.max_solver_time_in_seconds) ? 1.0 : 0.0;
    if (o.max_solver_time_in_seconds > 0.0) HIPCHK(s, hipMemcpy(L.d_sc5 + 5, &vote, 8, hipMemcpyHostToDevice));
    return UVS_OK;
}

// Host side of the trust-region loop (same order of tests as k_solve / SURVEY.md Appendix B).  Call after uvs_large_step (and after the
// caller all-reduced uvs_large_scalars()).  Note: on this path a (re)linearization is implied by need_lin BEFORE the next step.
int uvs_large_decide(uvs_solver* s) {
    if (!s || !s->L.active) return UVS_ERR_INVALID_ARG;
    auto& L = s->L; const uvs_options& o = s->opts;
    double out[LO_N + 8], sc[6];
    HIPCHK(s, hipMemcpy(out, L.d_out, sizeof(double) * (LO_N + 4), hipMemcpyDeviceToHost));
    HIPCHK(s, hipMemcpy(sc, L.d_sc5, sizeof(sc), hipMemcpyDeviceToHost));
    uvs_report& rep = L.rep;
    const double lc = out[LO_COST]; const double gm = out[LO_GMAX];
    if (L.first) {
        L.cost = lc; L.gmax = gm; L.first = false;
        rep.initial_cost = lc; rep.cost[0] = lc; rep.radius[0] = L.radius; rep.gradient_max_norm[0] = gm; rep.accepted[0] = 1;
        if (!std::isfinite(lc)) { L.term = UVS_TERM_NUMERIC_FAILURE; L.status = UVS_ERR_NUMERIC; L.done = true; return UVS_OK; }
    } else if (L.pending > 0) { L.cost = lc; L.gmax = gm; rep.cost[L.pending] = lc; rep.gradient_max_norm[L.pending] = gm; }
    L.pending = 0; L.need_lin = false;
    if (L.it >= o.max_num_iterations) { L.term = UVS_TERM_NO_CONVERGENCE; L.done = true; return UVS_OK; }
    if (o.max_solver_time_in_seconds > 0.0 && L.it > 0 && sc[5] > 0.0) {      // the host's clock, read in uvs_large_step; the vote is part of the scalars the ranks all-reduce, so every rank stops at the same iteration
        L.term = UVS_TERM_MAX_TIME; L.done = true; return UVS_OK;
    }
    if (L.gmax <= o.gradient_tolerance) { L.term = UVS_TERM_GRADIENT_TOL; L.done = true; return UVS_OK; }
    if (L.radius <= o.min_trust_region_radius) { L.term = UVS_TERM_MIN_RADIUS; L.done = true; return UVS_OK; }
    ++L.it;
    const int ti = L.it < UVS_MAX_ITER ? L.it : UVS_MAX_ITER;
    const double gd = out[LO_GD] + sc[0], dd2 = out[LO_DD2] + sc[1], step2 = out[LO_STEP2] + sc[2], xc2 = out[LO_XC2] + sc[3];
    const double mcc = 0.5 * (dd2 - gd);
    double cand = out[LO_FRAMECOST] + sc[4];
    bool ok = out[LO_CHOLOK] != 0.0 && std::isfinite(mcc) && std::isfinite(step2);
    rep.model_cost_change[ti] = mcc;
    if (!ok || !(mcc > 0.0)) {
        ++L.invalid; L.radius /= L.decr; L.decr *= 2.0; L.need_lin = true;
        rep.accepted[ti] = -1; rep.cost[ti] = L.cost; rep.candidate_cost[ti] = L.cost; rep.radius[ti] = L.radius; rep.gradient_max_norm[ti] = L.gmax;
        if (L.invalid >= o.max_consecutive_invalid_steps) { L.term = UVS_TERM_INVALID_STEPS; L.done = true; }
        return UVS_OK;
    }
    L.invalid = 0;
    if (!std::isfinite(cand)) cand = 1.7976931348623157e308;
    const double step_norm = std::sqrt(step2), rel = (L.cost - cand) / mcc;
    const bool successful = rel > o.min_relative_decrease;
    rep.candidate_cost[ti] = cand; rep.step_norm[ti] = step_norm; rep.relative_decrease[ti] = rel; rep.cost[ti] = L.cost; rep.radius[ti] = L.radius; rep.gradient_max_norm[ti] = L.gmax;
    bool stop = false;
    if (step_norm <= o.parameter_tolerance * (L.x_norm + o.parameter_tolerance)) { L.term = UVS_TERM_PARAMETER_TOL; stop = true; }
    else if (std::fabs(L.cost - cand) <= o.function_tolerance * L.cost) { L.term = UVS_TERM_FUNCTION_TOL; stop = true; }
    if (stop && !(o.function_tol_keeps_candidate && successful)) { L.done = true; return UVS_OK; }
    if (successful) {
        HIPCHK(s, hipMemcpyAsync(L.d_state + LS_X, L.d_state + LS_XC, UVS_XDIM * 8, hipMemcpyDeviceToDevice, s->stream));   // stream-ordered with the next launch (a plain D2D hipMemcpy
        // runs on the null stream, which this non-blocking stream does not wait for)
        L.sel ^= 1; ++L.nsucc; L.x_norm = std::sqrt(xc2);
        L.radius = L.radius / std::fmax(1.0 / 3.0, 1.0 - std::pow(2.0 * rel - 1.0, 3.0));
        L.radius = std::fmin(o.max_trust_region_radius, L.radius); L.decr = 2.0;
        L.cost = cand; L.need_lin = true; L.pending = ti;
        rep.accepted[ti] = 1; rep.cost[ti] = L.cost; rep.radius[ti] = L.radius;
        if (stop || L.it >= o.max_num_iterations) { if (!stop) L.term = UVS_TERM_NO_CONVERGENCE; L.done = true; }
    } else {
        L.radius /= L.decr; L.decr *= 2.0; L.need_lin = true;
        rep.accepted[ti] = 0; rep.radius[ti] = L.radius;
        if (L.it >= o.max_num_iterations) { L.term = UVS_TERM_NO_CONVERGENCE; L.done = true; }
    }
    return UVS_OK;
}

int uvs_large_finish(uvs_solver* s, uvs_state* out, uvs_report* rep) {
    if (!s || !s->L.active || !out || !rep) return UVS_ERR_INVALID_ARG;
    auto& L = s->L; const DevWin& h = s->hdrs[0];
    L.rep.status = L.status; L.rep.termination = L.term; L.rep.num_iterations = L.it; L.rep.num_successful = L.nsucc; L.rep.final_cost = L.cost;
    *rep = L.rep;
    double fr[UVS_XDIM];
    HIPCHK(s, hipMemcpy(fr, L.d_state + LS_X, sizeof(fr), hipMemcpyDeviceToHost));
    std::memcpy(out->pose, fr, 77 * 8); std::memcpy(out->speedbias, fr + 77, 99 * 8); std::memcpy(out->ex_pose, fr + 176, 7 * 8); out->td = fr[183];
    std::memcpy(out->relo_pose, fr + 184, sizeof(out->relo_pose));      // optimized when the window carries relocalization blocks, the input value otherwise
    if (out->inv_depth && h.n_points) HIPCHK(s, hipMemcpy(out->inv_depth, s->d_ws + (L.sel ? h.w_invd1 : h.w_invd0), (size_t)h.n_points * 8, hipMemcpyDeviceToHost));
    if (out->line_orth && h.n_lines) HIPCHK(s, hipMemcpy(out->line_orth, s->d_ws + (L.sel ? h.w_line1 : h.w_line0), (size_t)h.n_lines * 32, hipMemcpyDeviceToHost));
    L.active = false;
    return L.status;
}

// single-GPU convenience: the loop above with nothing to all-reduce; elapsed_ms (may be NULL) = wall time of the loop
int uvs_large_solve(uvs_solver* s, const uvs_window* w, uvs_state* out, uvs_report* rep) {
    int rc = uvs_large_begin(s, w);
    if (rc != UVS_OK) return rc;
    while (!uvs_large_done(s)) {
        if (uvs_large_need_linearize(s)) { if ((rc = uvs_large_linearize(s)) != UVS_OK) return rc; }
        if ((rc = uvs_large_step(s)) != UVS_OK) return rc;
        if ((rc = uvs_large_decide(s)) != UVS_OK) return rc;
    }
    return uvs_large_finish(s, out, rep);
}


// ---------------------------------------------------------------- fused loop: RCCL communicator owned by the handle, control on the device
// RCCL is resolved at run time (dlopen): the library itself carries no dependency on it, a process that already holds RCCL (PyTorch)
// shares that copy.  UVS_RCCL_LIB overrides the search.
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, uvs_rccl_id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string err;
};
RcclApi& rccl() {
    static RcclApi api;
    if (api.lib || !api.err.empty()) return api;
    const char* env = std::getenv("UVS_RCCL_LIB");
    if (env && *env) api.lib = dlopen(env, RTLD_NOW);            // an explicit library is taken as given, even when the process already holds another RCCL (PyTorch's)
    else {
        const char* names[2] = {"librccl.so.1", "librccl.so"};
        for (int pass = 0; pass < 2 && !api.lib; ++pass)          // first a copy that is already loaded, then a fresh one
            for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : 0)); if (api.lib) break; }
    }
    if (!api.lib) { api.err = "RCCL not found (librccl.so / librccl.so.1; set UVS_RCCL_LIB)"; return api; }
    api.GetUniqueId = (int (*)(void*))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (int (*)(void**, int, uvs_rccl_id, int))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
    api.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(api.lib, "ncclAllReduce");
    api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce) { api.err = "RCCL symbols missing"; api.lib = nullptr; }
    return api;
}
constexpr int kNcclDouble = 8, kNcclSum = 0;      // rccl.h: ncclFloat64 = 8, ncclSum = 0
}  // namespace

int uvs_large_comm_unique_id(uvs_rccl_id* id) {
    if (!id) return UVS_ERR_INVALID_ARG;
    RcclApi& r = rccl();
    if (!r.lib) return UVS_ERR_UNSUPPORTED;
    return r.GetUniqueId(id) == 0 ? UVS_OK : UVS_ERR_HIP;
}

int uvs_large_comm_init(uvs_solver* s, int nranks, int rank, const uvs_rccl_id* id) {
    if (!s || nranks < 1 || nranks > LG_MAXRANKS || rank < 0 || rank >= nranks || (nranks > 1 && !id)) return UVS_ERR_INVALID_ARG;
    auto& L = s->L;
    uvs_large_comm_destroy(s);
    L.rank = rank; L.nranks = nranks;
    if (nranks == 1 && !id) return UVS_OK;                        // nothing to exchange (with an id a one-rank communicator is built all the same: exercises the RCCL path on one GPU)
    RcclApi& r = rccl();
    if (!r.lib) { s->err = r.err; return UVS_ERR_UNSUPPORTED; }
    HIPCHK(s, hipSetDevice(s->device));
    const int rc = r.CommInitRank(&L.comm, nranks, *id, rank);
    if (rc != 0) { s->err = std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"); L.comm = nullptr; L.nranks = 1; L.rank = 0; return UVS_ERR_HIP; }
    return UVS_OK;
}

void uvs_large_comm_destroy(uvs_solver* s) {
    if (!s) return;
    auto& L = s->L;
    if (L.comm) { (void)hipSetDevice(s->device); rccl().CommDestroy(L.comm); L.comm = nullptr; }
    L.rank = 0; L.nranks = 1;
}

// Error inside the enqueue loop of the fused solve: drain what is already on the stream and leave the handle idle.  With several ranks the
// peers are still inside their collective -- the communicator must be considered broken afterwards (uvs_large_comm_destroy + re-init).
static int fused_abort(uvs_solver* s, const char* what) {
    (void)hipStreamSynchronize(s->stream);
    s->L.active = false;
    s->err = what;
    return UVS_ERR_HIP;
}

// ONE large window, landmark-sharded over the ranks of the handle's communicator (`w` = this rank's landmarks, frames / IMU / prior
// replicated), the whole Levenberg-Marquardt loop enqueued on the handle's stream without a host round trip: per iteration
//   k_large_chunks -> k_large_reduce -> ncclAllReduce(reduced, SUM, in place) -> k_large_solve -> k_large_backsub -> k_large_sum_bsums
//   -> ncclAllReduce(5 scalars) -> k_large_decide
// Every rank decides on identical numbers, so all ranks follow the same path; kernels of iterations after termination return at once.
int uvs_large_solve_fused(uvs_solver* s, const uvs_window* w, uvs_state* out, uvs_report* rep, float* loop_ms) {
    if (!s || !w || !out || !rep) return UVS_ERR_INVALID_ARG;
    // ONE stream, ONE wait: pinned upload -> k_large_init -> the passes -> k_large_pack -> pinned download.  (The step-wise API keeps
    // uvs_large_begin's host-side copies; here every small copy / memset is a line of k_large_init.)
    const uvs_window* arr[1] = {w};
    int rc = upload_windows(s, 1, arr, false, s->chunk_wgs());
    if (rc != UVS_OK) return rc;
    auto& L = s->L; const DevWin& h = s->hdrs[0]; const uvs_options& o = s->opts;
    // relocalization blocks are per-landmark, so a landmark shard may hold none of them while the all-reduced system carries the other ranks' relo_Pose rows: a rank
    // cannot tell from its own shard whether relo_Pose is a free block.  Not taken by a multi-rank solve (NO rank may pass n_relo_obs > 0; one rank takes them).
    if (L.nranks > 1 && w->n_relo_obs > 0) { s->err = "relocalization blocks are not taken by a landmark-sharded solve over several ranks"; return UVS_ERR_UNSUPPORTED; }
    {
        double* keep_ctl = L.d_ctl; uvs_report* keep_rep = L.d_rep; void* keep_comm = L.comm; const int keep_rank = L.rank, keep_nranks = L.nranks; double* keep_fimg = L.d_fimg;
        L = uvs_solver::Large{L.active, 0, 0, 0, 0, 0, 0, 0, 0, true, true, false, 0, 2, 0, 0, 0, 0, L.d_state, L.d_partials, L.d_reduced, L.d_bsums, L.d_out, L.d_sc5, L.cap_partials, L.cap_bsums, {}};
        L.active = false; L.n_chunks = h.n_chunks; L.radius = o.initial_trust_region_radius;      // active only while work is enqueued (set below): an allocation failure leaves the handle idle
        L.grid = std::min(h.n_chunks, s->chunk_wgs());
        L.d_ctl = keep_ctl; L.d_rep = keep_rep; L.comm = keep_comm; L.rank = keep_rank; L.nranks = keep_nranks; L.d_fimg = keep_fimg;
    }
    if (!L.d_state) { HIPCHK(s, hipMalloc((void**)&L.d_state, LG_STATE * 8)); HIPCHK(s, hipMalloc((void**)&L.d_reduced, LG_XCH_ALL * 8)); HIPCHK(s, hipMalloc((void**)&L.d_out, 64 * 8)); HIPCHK(s, hipMalloc((void**)&L.d_sc5, 8 * 8)); }
    if (!L.d_fimg) HIPCHK(s, hipMalloc((void**)&L.d_fimg, LG_FIMG * 8));
    if (!L.d_ctl) { HIPCHK(s, hipMalloc((void**)&L.d_ctl, 64 * 8)); HIPCHK(s, hipMalloc((void**)&L.d_rep, sizeof(uvs_report))); }
    if ((rc = ensure(s, (void**)&L.d_partials, &L.cap_partials, (size_t)std::max(L.grid, 1) * LG_ROW * 8)) != UVS_OK) return rc;
    if ((rc = ensure(s, (void**)&L.d_bsums, &L.cap_bsums, (size_t)std::max(L.n_chunks, 1) * 8 * 8)) != UVS_OK) return rc;
    constexpr int RD = (int)(sizeof(uvs_report) / 8);
    const size_t out_doubles = 64 + RD + UVS_XDIM + (size_t)h.n_points + 4 * (size_t)h.n_lines;
    if ((rc = ensure(s, (void**)&s->d_outpack, &s->d_outpack_cap, out_doubles * 8)) != UVS_OK) return rc;
    if ((rc = ensure_pinned(s, &s->h_out, &s->h_out_cap, out_doubles * 8)) != UVS_OK) return rc;
    double x2 = 0.0, l2 = 0.0;      // ||x||^2: frames (identical on every rank) and this rank's landmarks (summed over the ranks by the first all-reduce)
    for (int f = 0; f < UVS_NUM_FRAMES; ++f) { for (int k = 0; k < 7; ++k) x2 += w->pose[f][k] * w->pose[f][k]; for (int k = 0; k < 9; ++k) x2 += w->speedbias[f][k] * w->speedbias[f][k]; }
    if (o.estimate_td) x2 += w->td * w->td;
    if (o.estimate_extrinsic) for (int k = 0; k < 7; ++k) x2 += w->ex_pose[k] * w->ex_pose[k];
    if (w->n_relo_obs > 0) for (int k = 0; k < 7; ++k) x2 += w->relo_pose[k] * w->relo_pose[k];
    for (int k = 0; k < w->n_points; ++k) l2 += w->inv_depth[k] * w->inv_depth[k];
    for (int k = 0; k < 4 * w->n_lines; ++k) l2 += w->line_orth[k] * w->line_orth[k];
    L.local_x2 = l2; L.x_norm = std::sqrt(x2 + l2); L.frame_x2 = x2;
    std::memcpy(L.relo_pose_in, w->relo_pose, sizeof(L.relo_pose_in));
    L.active = true;
    hipLaunchKernelGGL(k_large_init, dim3(16), dim3(256), 0, s->stream, s->d_blobs, s->d_ws, L.d_state, L.d_ctl, L.d_rep, L.d_reduced, o.initial_trust_region_radius, L.frame_x2, L.local_x2);
    const char* lprof = std::getenv("UVS_LARGE_PROF");      // debug: per-workgroup timeline of the LAST k_large_chunks launch, written to this file
    const KOpts ko = make_kopts(o, lprof ? 7 : 0);
    const LargeCtl lc{L.d_ctl, L.rank, L.nranks};
    RcclApi& r = rccl();
    const int passes = std::max(1, o.max_num_iterations);
    const int rows = L.grid;
    const int bgrid = std::min(L.n_chunks, UVS_LARGE_OCC * s->chunk_wgs());      // k_large_backsub runs UVS_LARGE_OCC workgroups per compute unit
    HIPCHK(s, hipEventRecord(s->ev0, s->stream));
    for (int p = 0; p < passes; ++p) {
        if (s->large_chunks_nt == 512) { if (uvs_k_large_chunks512_launch(L.grid + 1, s->stream, s->d_blobs, s->d_ws, &ko, sizeof(ko), L.d_state, 0, 0, 0.0, L.d_partials, lc.ctl, lc.rank, lc.nranks, L.grid, L.d_fimg) != UVS_OK) return fused_abort(s, "k_large_chunks (512 threads): argument layout mismatch"); }
        else hipLaunchKernelGGL(k_large_chunks, dim3(L.grid + 1), dim3(NT), LDS_BYTES, s->stream, s->d_blobs, s->d_ws, ko, L.d_state, 0, 0, 0.0, L.d_partials, lc, L.grid, L.d_fimg);
        // (summing the partial rows inside k_large_solve instead of by a launch of its own was measured: one workgroup needs 15-24 us for what 314 do in 5)
        { const int n_ent = s->hdrs[0].relo2 ? LG_ROW : LG_RED; hipLaunchKernelGGL(k_large_reduce, dim3((n_ent + 15) / 16), dim3(256), 0, s->stream, L.d_partials, rows, L.d_reduced, lc, n_ent); }
        if (L.comm) { const int e = r.AllReduce(L.d_reduced, L.d_reduced, LG_XCH, kNcclDouble, kNcclSum, L.comm, s->stream); if (e != 0) return fused_abort(s, "ncclAllReduce(reduced) failed"); }
        if (s->large_solve_nt == 512) { if (uvs_k_large_solve512_launch(s->stream, s->d_blobs, s->d_ws, &ko, sizeof(ko), L.d_state, L.d_reduced, 0, 0.0, L.d_out, lc.ctl, lc.rank, lc.nranks, L.d_fimg) != UVS_OK) return fused_abort(s, "k_large_solve (512 threads): argument layout mismatch"); }
        else hipLaunchKernelGGL(k_large_solve, dim3(1), dim3(NT), LDS_BYTES, s->stream, s->d_blobs, s->d_ws, ko, L.d_state, L.d_reduced, 0, 0.0, L.d_out, lc, L.d_fimg);
        hipLaunchKernelGGL(k_large_backsub, dim3(bgrid + 1), dim3(NT), LDS_BYTES_BACKSUB, s->stream, s->d_blobs, s->d_ws, ko, L.d_state, 0, L.d_bsums, lc, bgrid, L.d_out);
        if (L.comm) {
            hipLaunchKernelGGL(k_large_sum_bsums, dim3(1), dim3(256), 0, s->stream, L.d_bsums, L.n_chunks, L.d_sc5, lc, ko.max_ticks);
            const int e = r.AllReduce(L.d_sc5, L.d_sc5, 8, kNcclDouble, kNcclSum, L.comm, s->stream); if (e != 0) return fused_abort(s, "ncclAllReduce(step scalars) failed");
            hipLaunchKernelGGL(k_large_decide, dim3(1), dim3(256), 0, s->stream, L.d_ctl, L.d_state, L.d_out, L.d_sc5, L.d_reduced, ko, L.d_rep, (const double*)nullptr, 0);
        } else hipLaunchKernelGGL(k_large_decide, dim3(1), dim3(256), 0, s->stream, L.d_ctl, L.d_state, L.d_out, L.d_sc5, L.d_reduced, ko, L.d_rep, (const double*)L.d_bsums, L.n_chunks);
    }
    HIPCHK(s, hipEventRecord(s->ev1, s->stream));
    hipLaunchKernelGGL(k_large_pack, dim3(16), dim3(256), 0, s->stream, s->d_blobs, s->d_ws, L.d_state, L.d_ctl, L.d_rep, s->d_outpack);
    HIPCHK(s, hipGetLastError());
    HIPCHK(s, hipMemcpyAsync(s->h_out, s->d_outpack, out_doubles * 8, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(s, hipStreamSynchronize(s->stream));
    if (loop_ms) HIPCHK(s, hipEventElapsedTime(loop_ms, s->ev0, s->ev1));
    if (lprof) {
        std::vector<long long> tp(1024 * 8);
        if ((s->large_chunks_nt == 512 ? uvs_k_large_chunks512_prof(tp.data(), tp.size()) == UVS_OK : hipMemcpyFromSymbol(tp.data(), HIP_SYMBOL(g_large_prof), tp.size() * 8) == hipSuccess)) { if (FILE* f = std::fopen(lprof, "wb")) { const int hdr[2] = {L.grid + 1, L.n_chunks}; std::fwrite(hdr, 4, 2, f); std::fwrite(tp.data(), 8, tp.size(), f); std::fclose(f); } }
    }
    const double* ho = (const double*)s->h_out;
    const double* ctl = ho;
    L.active = false;
    const bool unterminated = ctl[LC_DONE] == 0.0;      // cannot happen since k_large_decide tests the iteration cap on every branch; if it ever does, the caller still gets the last accepted state
    L.sel = (int)ctl[LC_SEL]; L.it = (int)ctl[LC_IT]; L.nsucc = (int)ctl[LC_NSUCC]; L.term = (int)ctl[LC_TERM]; L.status = (int)ctl[LC_STATUS]; L.cost = ctl[LC_COST]; L.done = true;
    std::memcpy(rep, ho + 64, sizeof(uvs_report));
    if (unterminated) {
        L.status = UVS_ERR_NUMERIC; L.term = UVS_TERM_NO_CONVERGENCE;
        rep->status = L.status; rep->termination = L.term; rep->num_iterations = L.it; rep->num_successful = L.nsucc; rep->final_cost = L.cost;
        s->err = "fused large-window loop did not terminate within max_num_iterations passes";
    }
    L.rep = *rep;
    const double* fr = ho + 64 + RD;
    std::memcpy(out->pose, fr, 77 * 8); std::memcpy(out->speedbias, fr + 77, 99 * 8); std::memcpy(out->ex_pose, fr + 176, 7 * 8); out->td = fr[183];
    std::memcpy(out->relo_pose, fr + 184, sizeof(out->relo_pose));      // optimized when the window carries relocalization blocks, the input value otherwise
    if (out->inv_depth && h.n_points) std::memcpy(out->inv_depth, fr + UVS_XDIM, (size_t)h.n_points * 8);
    if (out->line_orth && h.n_lines) std::memcpy(out->line_orth, fr + UVS_XDIM + h.n_points, (size_t)h.n_lines * 32);
    return L.status;
}

}  // extern "C"
