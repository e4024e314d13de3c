"""ctypes binding of libuvs_solver.so (the HIP C ABI of include/uvs_solver.h).

Plumbing only: every number is produced by the HIP kernels in csrc/.  Loading
fails loudly when the shared library has not been built, and `Solver(...)`
fails loudly (RuntimeError) when no GPU is present -- there is no CPU path.
"""
import ctypes as C
import os
import time

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("UVS_SOLVER_LIB", os.path.join(_HERE, "libuvs_solver.so"))      # the override is for A/B builds of the same library (tuning experiments)
_lib = None

EXPORTS = [
    "uvs_abi_version", "uvs_default_options", "uvs_create", "uvs_destroy", "uvs_last_error", "uvs_status_string",
    "uvs_solve_window", "uvs_batch_upload", "uvs_batch_solve", "uvs_batch_download", "uvs_batch_stream", "uvs_evaluate", "uvs_marginalize", "uvs_marginalize_resident", "uvs_marginalize_batch",
    "uvs_reduced_dim",
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.uvs_abi_version.restype = C.c_int
        L.uvs_default_options.argtypes = [C.POINTER(abi.Options)]
        L.uvs_create.argtypes = [C.POINTER(abi.Options), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.uvs_create.restype = C.c_int
        L.uvs_destroy.argtypes = [C.c_void_p]
        L.uvs_last_error.argtypes = [C.c_void_p]; L.uvs_last_error.restype = C.c_char_p
        L.uvs_status_string.argtypes = [C.c_int]; L.uvs_status_string.restype = C.c_char_p
        L.uvs_solve_window.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), C.POINTER(abi.StateC), C.POINTER(abi.Report)]
        L.uvs_solve_window.restype = C.c_int
        L.uvs_batch_upload.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.POINTER(abi.WindowC))]
        L.uvs_batch_upload.restype = C.c_int
        L.uvs_batch_solve.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.uvs_batch_solve.restype = C.c_int
        L.uvs_batch_download.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.StateC), C.POINTER(abi.Report)]
        L.uvs_batch_download.restype = C.c_int
        L.uvs_batch_stream.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.POINTER(abi.WindowC)), C.POINTER(abi.StateC), C.POINTER(abi.Report), C.POINTER(C.c_double)]
        L.uvs_batch_stream.restype = C.c_int
        L.uvs_evaluate.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), C.c_int, C.POINTER(abi.EvalC)]
        L.uvs_evaluate.restype = C.c_int
        L.uvs_marginalize.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), C.c_int, C.POINTER(abi.Prior)]
        L.uvs_marginalize.restype = C.c_int
        L.uvs_marginalize_resident.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), C.c_int, C.POINTER(abi.Prior)]
        L.uvs_marginalize_resident.restype = C.c_int
        L.uvs_marginalize_resident_begin.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), C.c_int]; L.uvs_marginalize_resident_begin.restype = C.c_int
        L.uvs_marginalize_wait.argtypes = [C.c_void_p, C.POINTER(abi.Prior)]; L.uvs_marginalize_wait.restype = C.c_int
        L.uvs_debug_first_iteration.argtypes = [C.c_void_p, C.POINTER(abi.WindowC)] + [abi.c_double_p] * 6
        L.uvs_debug_first_iteration.restype = C.c_int
        L.uvs_reduced_dim.argtypes = [C.POINTER(abi.Options)]; L.uvs_reduced_dim.restype = C.c_int
        L.uvs_large_begin.argtypes = [C.c_void_p, C.POINTER(abi.WindowC)]; L.uvs_large_begin.restype = C.c_int
        for name in ("uvs_large_need_linearize", "uvs_large_linearize", "uvs_large_step", "uvs_large_decide", "uvs_large_done"):
            getattr(L, name).argtypes = [C.c_void_p]; getattr(L, name).restype = C.c_int
        L.uvs_large_reduced.argtypes = [C.c_void_p, C.POINTER(C.c_int)]; L.uvs_large_reduced.restype = C.c_void_p
        L.uvs_large_scalars.argtypes = [C.c_void_p, C.POINTER(C.c_int)]; L.uvs_large_scalars.restype = C.c_void_p
        L.uvs_large_exchange_host.argtypes = [C.c_void_p, C.c_int, abi.c_double_p, C.c_int]; L.uvs_large_exchange_host.restype = C.c_int
        L.uvs_large_local_x2.argtypes = [C.c_void_p]; L.uvs_large_local_x2.restype = C.c_double
        L.uvs_large_set_landmark_x2.argtypes = [C.c_void_p, C.c_double]
        L.uvs_large_finish.argtypes = [C.c_void_p, C.POINTER(abi.StateC), C.POINTER(abi.Report)]; L.uvs_large_finish.restype = C.c_int
        L.uvs_large_solve.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), C.POINTER(abi.StateC), C.POINTER(abi.Report)]; L.uvs_large_solve.restype = C.c_int
        L.uvs_large_comm_unique_id.argtypes = [C.c_char_p]; L.uvs_large_comm_unique_id.restype = C.c_int
        L.uvs_large_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]; L.uvs_large_comm_init.restype = C.c_int
        L.uvs_large_comm_destroy.argtypes = [C.c_void_p]; L.uvs_large_comm_destroy.restype = None
        L.uvs_large_solve_fused.argtypes = [C.c_void_p, C.POINTER(abi.WindowC), C.POINTER(abi.StateC), C.POINTER(abi.Report), C.POINTER(C.c_float)]; L.uvs_large_solve_fused.restype = C.c_int
        _lib = L
    return _lib


class _DevPtr:
    """Exposes a raw device pointer through __cuda_array_interface__ so that torch can wrap it without a copy."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None}


def _device_tensor(ptr, n, device=None):
    import torch
    return torch.as_tensor(_DevPtr(ptr, n), device=device or "cuda")


class Solver:
    """Owns one `uvs_solver` handle (device buffers + stream) on one GPU."""

    def __init__(self, opts=None, device=0, max_batch=1024, max_points=1000, max_point_obs=16000, max_lines=1000, max_line_obs=16000):
        self.opts = opts or abi.default_options()
        self._h = C.c_void_p()
        rc = lib().uvs_create(C.byref(self.opts), device, max_batch, max_points, max_point_obs, max_lines, max_line_obs, C.byref(self._h))
        if rc != abi.UVS_OK:
            raise RuntimeError(f"uvs_create failed: {lib().uvs_status_string(rc).decode()} (rc={rc}); the HIP path is the only path")
        self._keep = None
        self._windows = None

    def close(self):
        if self._h:
            lib().uvs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=(abi.UVS_OK,)):
        if rc not in allow:
            raise RuntimeError(f"uvs error {rc}: {lib().uvs_status_string(rc).decode()} / {lib().uvs_last_error(self._h).decode()}")
        return rc

    # ---- single window (host buffers in / out, PCIe inclusive) -------------
    def solve(self, w: abi.Window):
        wc, keep = w.to_c()
        st = abi.State(len(w.inv_depth), len(w.line_orth)); sc = st.alloc_c()
        rep = abi.Report()
        t0 = time.perf_counter()
        rc = lib().uvs_solve_window(self._h, C.byref(wc), C.byref(sc), C.byref(rep))
        self.last_solve_ms = (time.perf_counter() - t0) * 1e3      # the C-ABI call alone: pack + H2D + kernel + D2H (what a C++ caller pays)
        self._check(rc, allow=(abi.UVS_OK, abi.UVS_ERR_NUMERIC))
        return st.from_c(sc), rep

    # ---- batch of independent windows, device resident --------------------
    def upload(self, windows):
        cs = [w.to_c() for w in windows]
        arr = (C.POINTER(abi.WindowC) * len(cs))(*[C.pointer(c[0]) for c in cs])
        t0 = time.perf_counter()
        self._check(lib().uvs_batch_upload(self._h, len(cs), arr))
        self.last_upload_ms = (time.perf_counter() - t0) * 1e3      # the C-ABI call alone: host packing + H2D (the ctypes conversion above is python overhead)
        self._windows = list(windows)

    def solve_resident(self):
        """Runs the solve kernel on the uploaded batch; returns the HIP-event time of the launch in ms."""
        ms = C.c_float(0.0)
        self._check(lib().uvs_batch_solve(self._h, C.byref(ms)))
        return float(ms.value)

    def download(self, n=None):
        ws = self._windows
        n = n or len(ws)
        states = [abi.State(len(w.inv_depth), len(w.line_orth)) for w in ws[:n]]
        sarr = (abi.StateC * n)()
        for i, st in enumerate(states):
            sarr[i].inv_depth = abi._dp(st.inv_depth); sarr[i].line_orth = abi._dp(st.line_orth)
        reps = (abi.Report * n)()
        t0 = time.perf_counter()
        rc = lib().uvs_batch_download(self._h, n, sarr, reps)
        self.last_download_ms = (time.perf_counter() - t0) * 1e3      # the C-ABI call alone
        self._check(rc, allow=(abi.UVS_OK, abi.UVS_ERR_NUMERIC))
        for i, st in enumerate(states):
            st.from_c(sarr[i])
        return states, list(reps)

    def stream(self, windows, per_batch, want_states=True):
        """len(windows) / per_batch batches end to end (uvs_batch_stream): host packing, upload, solve and download of consecutive batches
        overlap.  Returns (states, reports, wall_ms of the C-ABI call)."""
        n = len(windows)
        assert n % per_batch == 0
        cs = [w.to_c() for w in windows]
        arr = (C.POINTER(abi.WindowC) * n)(*[C.pointer(c[0]) for c in cs])
        states = [abi.State(len(w.inv_depth), len(w.line_orth)) for w in windows] if want_states else []
        sarr = (abi.StateC * n)() if want_states else None
        for i, st in enumerate(states):
            sarr[i].inv_depth = abi._dp(st.inv_depth); sarr[i].line_orth = abi._dp(st.line_orth)
        reps = (abi.Report * n)()
        ms = C.c_double(0.0)
        self._check(lib().uvs_batch_stream(self._h, n // per_batch, per_batch, arr, sarr, reps, C.byref(ms)), allow=(abi.UVS_OK, abi.UVS_ERR_NUMERIC))
        for i, st in enumerate(states):
            st.from_c(sarr[i])
        return states, list(reps), float(ms.value)

    # ---- one large window over the whole GPU / several GPUs (BASELINE configs[3]) ----
    def large_solve(self, w: abi.Window, dist=None, device=None):
        """Landmark-sharded solve of ONE window.  With `dist` (an initialised torch.distributed module) `w` must hold only this
        rank's landmarks (synth.shard_landmarks); the pose-block partials and 5 scalars are all-reduced over RCCL in place."""
        wc, keep = w.to_c()
        st = abi.State(len(w.inv_depth), len(w.line_orth)); sc = st.alloc_c()
        rep = abi.Report()
        L = lib()
        if dist is None:
            self._check(L.uvs_large_solve(self._h, C.byref(wc), C.byref(sc), C.byref(rep)), allow=(abi.UVS_OK, abi.UVS_ERR_NUMERIC))
            return st.from_c(sc), rep
        import torch
        self._check(L.uvs_large_begin(self._h, C.byref(wc)))
        gpu_aware = dist.get_backend() == "nccl"          # RCCL reduces the solver's device buffers in place; otherwise stage through the host
        x2 = torch.tensor([L.uvs_large_local_x2(self._h)], dtype=torch.float64, device=device if gpu_aware else "cpu")
        dist.all_reduce(x2)
        L.uvs_large_set_landmark_x2(self._h, float(x2.item()))
        n = C.c_int(0)
        p_red = L.uvs_large_reduced(self._h, C.byref(n)); n_red = n.value
        p_sc = L.uvs_large_scalars(self._h, C.byref(n)); n_sc = n.value
        if gpu_aware:
            red = _device_tensor(p_red, n_red, device); scal = _device_tensor(p_sc, n_sc, device)

        def exchange(which):
            """SUM all-reduce of vector `which`; entry LG_ACC + 1 (= n - 7) of the reduced vector is a MAX."""
            if gpu_aware:
                t = red if which == 0 else scal
                mx = t[-7:-6].clone() if which == 0 else None
                dist.all_reduce(t)
                if which == 0:
                    dist.all_reduce(mx, op=dist.ReduceOp.MAX); t[-7:-6] = mx
                torch.cuda.synchronize()
            else:
                buf = np.zeros(n_red if which == 0 else n_sc)
                self._check(L.uvs_large_exchange_host(self._h, which, abi._dp(buf), 0))
                t = torch.from_numpy(buf)
                mx = t[-7:-6].clone() if which == 0 else None
                dist.all_reduce(t)
                if which == 0:
                    dist.all_reduce(mx, op=dist.ReduceOp.MAX); t[-7:-6] = mx
                self._check(L.uvs_large_exchange_host(self._h, which, abi._dp(buf), 1))

        while not L.uvs_large_done(self._h):
            if L.uvs_large_need_linearize(self._h):
                self._check(L.uvs_large_linearize(self._h))
                exchange(0)                                # the pose-block partials (46.7 KB)
            self._check(L.uvs_large_step(self._h))
            exchange(1)
            self._check(L.uvs_large_decide(self._h))
        self._check(L.uvs_large_finish(self._h, C.byref(sc), C.byref(rep)), allow=(abi.UVS_OK, abi.UVS_ERR_NUMERIC))
        return st.from_c(sc), rep

    def large_comm_init(self, dist=None):
        """The handle's own RCCL communicator for large_solve_fused(): rank 0 draws the id, torch.distributed (any backend) carries its
        128 bytes to the other ranks, every rank joins.  Without `dist` (or with one rank) nothing is exchanged and no RCCL is needed."""
        L = lib()
        if dist == "self":          # one-rank communicator through RCCL (tests: the dlopen'ed API, in-place all-reduce on the handle's stream)
            buf = C.create_string_buffer(128)
            self._check(L.uvs_large_comm_unique_id(buf)); self._check(L.uvs_large_comm_init(self._h, 1, 0, buf.raw)); return
        if dist is None or dist.get_world_size() == 1:
            self._check(L.uvs_large_comm_init(self._h, 1, 0, None)); return
        rank, world = dist.get_rank(), dist.get_world_size()
        buf = C.create_string_buffer(128)
        if rank == 0:
            self._check(L.uvs_large_comm_unique_id(buf))
        box = [buf.raw]
        dist.broadcast_object_list(box, src=0)
        self._check(L.uvs_large_comm_init(self._h, world, rank, box[0]))

    def large_solve_fused(self, w: abi.Window):
        """One landmark shard of ONE large window through the fused loop (include/uvs_solver.h: uvs_large_solve_fused): the whole LM loop
        is enqueued on the handle's stream, the pose-block partials and the step scalars are all-reduced in place by the handle's RCCL
        communicator (large_comm_init), accept / reject runs on the device.  Returns (state, report, loop_ms)."""
        wc, keep = w.to_c()
        st = abi.State(len(w.inv_depth), len(w.line_orth)); sc = st.alloc_c()
        rep = abi.Report(); ms = C.c_float(0.0)
        t0 = time.perf_counter()
        rc = lib().uvs_large_solve_fused(self._h, C.byref(wc), C.byref(sc), C.byref(rep), C.byref(ms))
        self.last_solve_ms = (time.perf_counter() - t0) * 1e3      # the C-ABI call alone: pack + H2D + LM loop + D2H
        self._check(rc, allow=(abi.UVS_OK, abi.UVS_ERR_NUMERIC))
        return st.from_c(sc), rep, float(ms.value)

    # ---- diagnostics -------------------------------------------------------
    def evaluate(self, w: abi.Window, robust=True):
        wc, keep = w.to_c()
        ev = abi.Eval(w); ec = ev.alloc_c()
        self._check(lib().uvs_evaluate(self._h, C.byref(wc), int(robust), C.byref(ec)))
        ev.cost = ec.cost
        return ev

    def marginalize(self, w: abi.Window, flag=0, resident=False):
        """resident=True: the factors of `w` are those of the last single-window upload / solve of this handle; only its state is sent."""
        wc, keep = w.to_c()
        p = abi.Prior()
        fn = lib().uvs_marginalize_resident if resident else lib().uvs_marginalize
        self._check(fn(self._h, C.byref(wc), flag, C.byref(p)))
        return p

    def marginalize_batch(self, windows, flags, check=True):
        """uvs_marginalize_batch: the marginalization of independent windows with the cubic work of all of them in two launches; returns (priors, per-window status codes).
        check=False: a failing window does not raise -- its status code says so, the other windows' priors are valid."""
        n = len(windows)
        keeps = [w.to_c() for w in windows]
        arr = (C.POINTER(abi.WindowC) * n)(*[C.pointer(k[0]) for k in keeps])
        fl = (C.c_int * n)(*[int(f) for f in flags])
        pri = (abi.Prior * n)()
        st = (C.c_int * n)()
        L = lib()
        L.uvs_marginalize_batch.restype = C.c_int
        rc = L.uvs_marginalize_batch(self._h, n, arr, fl, pri, st)
        if check: self._check(rc)
        return [pri[i] for i in range(n)], list(st)

    def marginalize_begin(self, w: abi.Window, flag=0):
        """uvs_marginalize_resident_begin: the marginalization of the resident window on a worker thread of the handle; marginalize_wait() delivers the prior.
        The C structures of `w` are kept alive on this object until then (the ABI's contract for the caller's arrays)."""
        wc, keep = w.to_c()
        self._marg_keep = (wc, keep, w)
        self._check(lib().uvs_marginalize_resident_begin(self._h, C.byref(wc), flag))

    def marginalize_wait(self):
        p = abi.Prior()
        rc = lib().uvs_marginalize_wait(self._h, C.byref(p))
        self._marg_keep = None
        self._check(rc)
        return p

    def debug_first_iteration(self, w: abi.Window):
        wc, keep = w.to_c()
        S = np.zeros((176, 176)); g = np.zeros(176); hd = np.zeros(176); dd = np.zeros(176); step = np.zeros(176); scal = np.zeros(40)
        self._check(lib().uvs_debug_first_iteration(self._h, C.byref(wc), *[abi._dp(a) for a in (S, g, hd, dd, step, scal)]))
        return dict(S=S, g=g, hd=hd, dd=dd, step=step, cost=scal[0], gmax=scal[1], chol_ok=scal[2], mcc=scal[3], step2=scal[4],
                    cycles=dict(zip(['setup', 'obs', 'lmprep', 'gather', 'assemble', 'chol', 'trsv', 'backsub', 'cost', 'misc', 'chol_diag', 'chol_panel', 'chol_trail', 'asm_imu', 'asm_zero', 'asm_add'], scal[8:24])), sub_timers=dict(cost_phase=dict(zip(['stage_dx', 'prior_residual', 'observations', 'imu'], scal[24:28])), chol_busy_per_wave=scal[28:32].copy()))
