// fake_nccl.cpp -- TEST INFRASTRUCTURE, not part of the product: a stand-in for librccl.so that lets TWO PROCESSES SHARING ONE GPU run
// uvs_large_solve_fused() with nranks = 2 (tests/test_large_multirank.py).  libuvs_solver.so resolves RCCL with dlopen(UVS_RCCL_LIB), so the
// solver code under test is byte for byte what runs over xGMI: the per-rank MAX slots of the first exchange, k_large_sum_bsums and the
// second (5-scalar) all-reduce execute for real; only the transport differs.
//
// Exports the five entry points the solver binds (csrc/uvs_solver.hip: RcclApi):  ncclGetUniqueId, ncclCommInitRank, ncclAllReduce,
// ncclCommDestroy, ncclGetErrorString.  Transport: a POSIX shared-memory segment named in the unique id; an all-reduce waits for the
// stream, copies the buffer to the rank's slot, meets the other ranks at a barrier, adds the slots IN RANK ORDER (every rank gets the
// same bits, as a ring all-reduce of RCCL does) and copies the sum back.  Every wait has a deadline, so a lost peer is an error, not a hang.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {
constexpr int kMaxRanks = 8, kMaxCount = 8192;
constexpr double kDeadlineSeconds = 120.0;
struct Shared {
    std::atomic<int> arrived, generation, calls;
    double slot[kMaxRanks][kMaxCount];
};
struct Comm { Shared* sh; int nranks, rank; char name[64]; double* host; };
struct Id { char name[64]; char pad[64]; };      // ncclUniqueId is 128 opaque bytes

bool barrier(Comm* c) {
    Shared* s = c->sh;
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == c->nranks) { s->arrived.store(0); s->generation.fetch_add(1); return true; }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        sched_yield();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kDeadlineSeconds) return false;
    }
    return true;
}
}  // namespace

extern "C" {

int ncclGetUniqueId(void* out) {
    Id id; std::memset(&id, 0, sizeof(id));
    std::snprintf(id.name, sizeof(id.name), "/uvs_fake_nccl_%d_%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    const int fd = shm_open(id.name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) return 2;
    close(fd);      // a fresh segment is zero-filled: counters start at 0
    std::memcpy(out, &id, sizeof(id));
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, Id id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
    const int fd = shm_open(id.name, O_RDWR, 0600);
    if (fd < 0) return 2;
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    Comm* c = new Comm{(Shared*)p, nranks, rank, {0}, nullptr};
    std::memcpy(c->name, id.name, sizeof(c->name));
    if (hipHostMalloc((void**)&c->host, kMaxCount * sizeof(double)) != hipSuccess) { delete c; return 1; }
    if (!barrier(c)) { delete c; return 6; }      // everybody has mapped the segment
    if (rank == 0) shm_unlink(id.name);           // the mappings keep it alive
    *comm = c;
    return 0;
}

// ncclFloat64 = 8, ncclSum = 0 are the only type / operation the solver uses
int ncclAllReduce(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || datatype != 8 || op != 0 || count > (size_t)kMaxCount) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    if (hipMemcpy(c->host, send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    std::memcpy(c->sh->slot[c->rank], c->host, count * sizeof(double));
    if (!barrier(c)) return 6;
    for (size_t i = 0; i < count; ++i) { double s = c->sh->slot[0][i]; for (int r = 1; r < c->nranks; ++r) s += c->sh->slot[r][i]; c->host[i] = s; }
    if (!barrier(c)) return 6;                    // nobody overwrites a slot another rank is still reading
    if (c->rank == 0) c->sh->calls.fetch_add(1);
    if (hipMemcpy(recv, c->host, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 1;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    if (c->host) (void)hipHostFree(c->host);
    munmap(c->sh, sizeof(Shared));
    delete c;
    return 0;
}

const char* ncclGetErrorString(int e) { return e == 0 ? "success" : e == 6 ? "fake nccl: a peer did not arrive before the deadline" : "fake nccl: error"; }

// test probe: how many all-reduces the communicator has carried (rank 0 counts)
int uvs_fake_nccl_calls(void* comm) { Comm* c = (Comm*)comm; return c ? c->sh->calls.load() : -1; }

}  // extern "C"
