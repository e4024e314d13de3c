"""ctypes mirror of include/uvs_solver.h (the C ABI of the sliding-window back-end).

Pure data-layout definitions plus helpers that turn a python-side `Window`
(numpy arrays, see synth.py) into a `uvs_window` and back.  Nothing here
computes: the arithmetic lives in csrc/ (HIP) and is reached through
libuvs_solver.so.  The same structures are used by the tests to call the CPU
oracle (oracle/liboracle.so), which shares the header.
"""
import ctypes as C
import numpy as np

WINDOW_SIZE = 10
NUM_FRAMES = WINDOW_SIZE + 1
MAX_ITER = 64
MAX_PRIOR_BLOCKS = 16
MAX_PRIOR_DIM = 96

UVS_OK, UVS_ERR_INVALID_ARG, UVS_ERR_UNSUPPORTED, UVS_ERR_NO_DEVICE, UVS_ERR_HIP, UVS_ERR_CAPACITY, UVS_ERR_NUMERIC = range(7)
UVS_BLOCK_POSE, UVS_BLOCK_SPEEDBIAS, UVS_BLOCK_EX_POSE, UVS_BLOCK_TD = range(4)      # uvs_prior.block_kind
TERM_NAMES = ["NO_CONVERGENCE", "GRADIENT_TOL", "PARAMETER_TOL", "FUNCTION_TOL", "MIN_RADIUS", "INVALID_STEPS", "NUMERIC_FAILURE"]
BLOCK_POSE, BLOCK_SPEEDBIAS, BLOCK_EX_POSE, BLOCK_TD = range(4)

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int32)


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32), ("estimate_extrinsic", C.c_int32), ("estimate_td", C.c_int32),
        ("function_tol_keeps_candidate", C.c_int32),
        ("focal_length", C.c_double), ("point_sqrt_info", C.c_double), ("line_factor", C.c_double), ("vp_factor", C.c_double),
        ("loss_point", C.c_double), ("loss_line", C.c_double), ("loss_vp", C.c_double), ("gravity", C.c_double * 3),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double), ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("max_consecutive_invalid_steps", C.c_int32), ("jacobi_scaling", C.c_int32),
        ("max_solver_time_in_seconds", C.c_double),
    ]


def default_options():
    """EuRoC values (config/euroc/euroc_config.yaml) + Ceres defaults; mirrors uvs_default_options()."""
    o = Options()
    o.max_num_iterations = 10
    o.estimate_extrinsic = 0
    o.estimate_td = 0
    o.function_tol_keeps_candidate = 0
    o.focal_length = 461.6
    o.point_sqrt_info = 461.6 / 1.6
    o.line_factor = 300.0
    o.vp_factor = 10.0
    o.loss_point, o.loss_line, o.loss_vp = 1.0, 0.1, 1.0
    o.gravity[0], o.gravity[1], o.gravity[2] = 0.0, 0.0, 9.81007
    o.initial_trust_region_radius = 1e4
    o.max_trust_region_radius = 1e16
    o.min_trust_region_radius = 1e-32
    o.min_relative_decrease = 1e-3
    o.min_lm_diagonal = 1e-6
    o.max_lm_diagonal = 1e32
    o.function_tolerance = 1e-6
    o.gradient_tolerance = 1e-10
    o.parameter_tolerance = 1e-8
    o.max_consecutive_invalid_steps = 5
    o.jacobi_scaling = 1
    return o


class ImuBlock(C.Structure):
    _fields_ = [
        ("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
        ("linearized_ba", C.c_double * 3), ("linearized_bg", C.c_double * 3),
        ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225),
        ("frame_i", C.c_int32), ("skip", C.c_int32),
    ]


class Prior(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("n_blocks", C.c_int32),
        ("block_kind", C.c_int32 * MAX_PRIOR_BLOCKS), ("block_frame", C.c_int32 * MAX_PRIOR_BLOCKS),
        ("block_size", C.c_int32 * MAX_PRIOR_BLOCKS), ("block_idx", C.c_int32 * MAX_PRIOR_BLOCKS),
        ("x0_off", C.c_int32 * MAX_PRIOR_BLOCKS),
        ("x0", C.c_double * (MAX_PRIOR_BLOCKS * 9)),
        ("linearized_residuals", C.c_double * MAX_PRIOR_DIM),
        ("linearized_jacobians", C.c_double * (MAX_PRIOR_DIM * MAX_PRIOR_DIM)),
    ]

    def J0(self):
        n = self.n
        return np.ctypeslib.as_array(self.linearized_jacobians)[: n * n].reshape(n, n).copy()

    def r0(self):
        return np.ctypeslib.as_array(self.linearized_residuals)[: self.n].copy()

    def copy(self):
        p = Prior()
        C.memmove(C.byref(p), C.byref(self), C.sizeof(Prior))
        return p


class WindowC(C.Structure):
    _fields_ = [
        ("pose", (C.c_double * 7) * NUM_FRAMES), ("speedbias", (C.c_double * 9) * NUM_FRAMES), ("ex_pose", C.c_double * 7),
        ("td", C.c_double),
        ("n_points", C.c_int32), ("n_point_obs", C.c_int32),
        ("inv_depth", c_double_p), ("pt_lm", c_int_p), ("pt_fi", c_int_p), ("pt_fj", c_int_p), ("pt_pi", c_double_p), ("pt_pj", c_double_p),
        ("n_lines", C.c_int32), ("n_line_obs", C.c_int32),
        ("line_orth", c_double_p), ("ln_lm", c_int_p), ("ln_fj", c_int_p), ("ln_sp", c_double_p), ("ln_ep", c_double_p),
        ("ln_has_vp", c_int_p), ("ln_vp", c_double_p),
        ("n_imu", C.c_int32), ("imu", C.POINTER(ImuBlock)),
        ("prior", C.POINTER(Prior)),
        ("pt_vel_i", c_double_p), ("pt_vel_j", c_double_p), ("pt_td_i", c_double_p), ("pt_td_j", c_double_p),
        ("n_relo_obs", C.c_int32), ("relo_pose", C.c_double * 7), ("relo_lm", c_int_p), ("relo_pi", c_double_p), ("relo_pj", c_double_p),
    ]


class StateC(C.Structure):
    _fields_ = [
        ("pose", (C.c_double * 7) * NUM_FRAMES), ("speedbias", (C.c_double * 9) * NUM_FRAMES), ("ex_pose", C.c_double * 7),
        ("td", C.c_double), ("inv_depth", c_double_p), ("line_orth", c_double_p), ("relo_pose", C.c_double * 7),
    ]


class Report(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("termination", C.c_int32), ("num_iterations", C.c_int32), ("num_successful", C.c_int32),
        ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("cost", C.c_double * (MAX_ITER + 1)), ("candidate_cost", C.c_double * (MAX_ITER + 1)),
        ("model_cost_change", C.c_double * (MAX_ITER + 1)), ("relative_decrease", C.c_double * (MAX_ITER + 1)),
        ("radius", C.c_double * (MAX_ITER + 1)), ("step_norm", C.c_double * (MAX_ITER + 1)),
        ("gradient_max_norm", C.c_double * (MAX_ITER + 1)), ("accepted", C.c_int32 * (MAX_ITER + 1)),
    ]

    def trace(self):
        k = self.num_iterations + 1
        f = lambda a: np.array(a[:k])
        return dict(cost=f(self.cost), candidate_cost=f(self.candidate_cost), model_cost_change=f(self.model_cost_change),
                    relative_decrease=f(self.relative_decrease), radius=f(self.radius), step_norm=f(self.step_norm),
                    gradient_max_norm=f(self.gradient_max_norm), accepted=np.array(self.accepted[:k]))


class EvalC(C.Structure):
    _fields_ = [
        ("pt_r", c_double_p), ("pt_J", c_double_p), ("ln_r", c_double_p), ("ln_J", c_double_p), ("vp_r", c_double_p), ("vp_J", c_double_p),
        ("imu_r", c_double_p), ("imu_J", c_double_p), ("prior_r", c_double_p), ("cost", C.c_double), ("pt_Jtd", c_double_p),
    ]


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


class Window:
    """Python-side sliding window: numpy mirrors of the para_* arrays and the factor lists.

    Field meanings and reference citations are those of `uvs_window` in include/uvs_solver.h.
    """

    def __init__(self):
        self.pose = np.zeros((NUM_FRAMES, 7)); self.pose[:, 6] = 1.0
        self.speedbias = np.zeros((NUM_FRAMES, 9))
        self.ex_pose = np.zeros(7); self.ex_pose[6] = 1.0
        self.td = 0.0
        self.inv_depth = np.zeros(0)
        self.pt_lm = np.zeros(0, np.int32); self.pt_fi = np.zeros(0, np.int32); self.pt_fj = np.zeros(0, np.int32)
        self.pt_pi = np.zeros((0, 3)); self.pt_pj = np.zeros((0, 3))
        self.pt_vel_i = None; self.pt_vel_j = None; self.pt_td_i = None; self.pt_td_j = None      # ProjectionTdFactor inputs (estimate_td), [n_obs,2] / [n_obs]
        # relocalization blocks (estimator.cpp:944-978): landmark index, pts_i (first observation), pts_j (match point); relo_Pose
        self.relo_pose = np.zeros(7); self.relo_pose[6] = 1.0
        self.relo_frame = 0    # relo_frame_local_index: not an input of the solve (Estimator::double2vector reads it), carried by the window file
        self.relo_lm = np.zeros(0, np.int32); self.relo_pi = np.zeros((0, 3)); self.relo_pj = np.zeros((0, 3))
        self.line_orth = np.zeros((0, 4))
        self.ln_lm = np.zeros(0, np.int32); self.ln_fj = np.zeros(0, np.int32)
        self.ln_sp = np.zeros((0, 3)); self.ln_ep = np.zeros((0, 3))
        self.ln_has_vp = np.zeros(0, np.int32); self.ln_vp = np.zeros((0, 3))
        self.imu = []          # list of dict(sum_dt, delta_p, delta_q, delta_v, linearized_ba, linearized_bg, jacobian(15,15), covariance(15,15), frame_i, skip)
        self.prior = None      # abi.Prior or None
        self.truth = None      # optional dict with ground-truth state (synthetic windows)

    # -- conversion ---------------------------------------------------------
    def to_c(self):
        """Returns (WindowC, keepalive) -- keepalive must outlive every use of the struct."""
        k = {}
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        w = WindowC()
        pose, sb, ex = f64(self.pose), f64(self.speedbias), f64(self.ex_pose)
        C.memmove(w.pose, pose.ctypes.data, pose.nbytes)
        C.memmove(w.speedbias, sb.ctypes.data, sb.nbytes)
        C.memmove(w.ex_pose, ex.ctypes.data, ex.nbytes)
        w.td = float(self.td)
        for name in ("inv_depth", "pt_pi", "pt_pj", "line_orth", "ln_sp", "ln_ep", "ln_vp"):
            k[name] = f64(getattr(self, name)); setattr(w, name, _dp(k[name]))
        for name in ("pt_lm", "pt_fi", "pt_fj", "ln_lm", "ln_fj", "ln_has_vp"):
            k[name] = i32(getattr(self, name)); setattr(w, name, _ip(k[name]))
        for name in ("pt_vel_i", "pt_vel_j", "pt_td_i", "pt_td_j"):
            if getattr(self, name) is not None:
                k[name] = f64(getattr(self, name)); setattr(w, name, _dp(k[name]))
        k["relo_lm"] = i32(self.relo_lm); k["relo_pi"] = f64(self.relo_pi); k["relo_pj"] = f64(self.relo_pj); relo = f64(self.relo_pose)
        w.n_relo_obs = len(k["relo_lm"]); w.relo_lm = _ip(k["relo_lm"]); w.relo_pi = _dp(k["relo_pi"]); w.relo_pj = _dp(k["relo_pj"])
        C.memmove(w.relo_pose, relo.ctypes.data, relo.nbytes)
        w.n_points = len(k["inv_depth"]); w.n_point_obs = len(k["pt_lm"])
        w.n_lines = len(k["line_orth"]); w.n_line_obs = len(k["ln_lm"])
        n_imu = len(self.imu)
        arr = (ImuBlock * max(n_imu, 1))()
        for b, d in enumerate(self.imu):
            ib = arr[b]
            ib.sum_dt = float(d["sum_dt"])
            for name, n in (("delta_p", 3), ("delta_q", 4), ("delta_v", 3), ("linearized_ba", 3), ("linearized_bg", 3)):
                v = f64(d[name]).ravel(); assert v.size == n
                C.memmove(getattr(ib, name), v.ctypes.data, v.nbytes)
            for name in ("jacobian", "covariance"):
                v = f64(d[name]).reshape(225)
                C.memmove(getattr(ib, name), v.ctypes.data, v.nbytes)
            ib.frame_i = int(d["frame_i"]); ib.skip = int(d.get("skip", 0))
        k["imu"] = arr
        w.n_imu = n_imu
        w.imu = C.cast(arr, C.POINTER(ImuBlock))
        if self.prior is not None and self.prior.n > 0:
            k["prior"] = self.prior
            w.prior = C.pointer(self.prior)
        else:
            w.prior = None
        return w, k

    # -- record / replay file (layout documented in host/window_io.h) ------------
    def save(self, path):
        f64 = lambda a: np.ascontiguousarray(a, dtype="<f8").tobytes()
        i32 = lambda a: np.ascontiguousarray(a, dtype="<i4").tobytes()
        npo, nlo = len(self.pt_lm), len(self.ln_lm)
        pn = self.prior.n if self.prior is not None else 0
        pnb = self.prior.n_blocks if pn else 0
        with open(path, "wb") as f:
            f.write(b"UVSWIN01")
            has_td = self.pt_vel_i is not None
            nrl = len(self.relo_lm)
            f.write(i32([len(self.inv_depth), npo, len(self.line_orth), nlo, len(self.imu), pn, pnb, (1 if has_td else 0) | (2 if nrl else 0)]))
            f.write(f64(self.pose)); f.write(f64(self.speedbias)); f.write(f64(self.ex_pose)); f.write(f64([self.td]))
            f.write(f64(self.inv_depth))
            f.write(i32(self.pt_lm)); f.write(i32(self.pt_fi)); f.write(i32(self.pt_fj))
            if npo % 2: f.write(i32([0]))
            f.write(f64(self.pt_pi)); f.write(f64(self.pt_pj))
            if has_td:      # ProjectionTdFactor inputs (header word 8 = 1)
                f.write(f64(self.pt_vel_i)); f.write(f64(self.pt_vel_j)); f.write(f64(self.pt_td_i)); f.write(f64(self.pt_td_j))
            f.write(f64(self.line_orth))
            f.write(i32(self.ln_lm)); f.write(i32(self.ln_fj)); f.write(i32(self.ln_has_vp))
            if nlo % 2: f.write(i32([0]))
            f.write(f64(self.ln_sp)); f.write(f64(self.ln_ep)); f.write(f64(self.ln_vp))
            for b in self.imu:
                f.write(f64([b["sum_dt"]])); f.write(f64(b["delta_p"])); f.write(f64(b["delta_q"])); f.write(f64(b["delta_v"]))
                f.write(f64(b["linearized_ba"])); f.write(f64(b["linearized_bg"])); f.write(f64(np.asarray(b["jacobian"]).reshape(225)))
                f.write(f64(np.asarray(b["covariance"]).reshape(225))); f.write(i32([b["frame_i"], b.get("skip", 0)]))
            if pn:
                p = self.prior
                for name in ("block_kind", "block_frame", "block_size", "block_idx", "x0_off"):
                    f.write(i32(list(getattr(p, name))))
                f.write(f64(list(p.x0))); f.write(f64(p.r0())); f.write(f64(p.J0()))
            if nrl:         # relocalization section (header word 8, bit 1): count, relo_frame_local_index, relo_Pose, landmark indices (+pad), pts_i, pts_j
                f.write(i32([nrl, int(self.relo_frame)])); f.write(f64(self.relo_pose)); f.write(i32(self.relo_lm))
                if nrl % 2: f.write(i32([0]))
                f.write(f64(self.relo_pi)); f.write(f64(self.relo_pj))

    @staticmethod
    def load(path):
        """Reads a window file (the layout of save() / host/window_io.h), e.g. one written by the UVS_DUMP_WINDOWS record hook."""
        buf = open(path, "rb").read()
        if buf[:8] != b"UVSWIN01": raise ValueError("not a window file: %s" % path)
        pos = [8]

        def take(dtype, n):
            a = np.frombuffer(buf, dtype=dtype, count=n, offset=pos[0]).copy(); pos[0] += a.nbytes; return a

        np_, npo, nl, nlo, ni, pn, pnb, flags = (int(v) for v in take("<i4", 8))
        has_td, has_relo = flags & 1, flags & 2
        # header counts are untrusted: the prior lands in fixed-size ctypes arrays
        if min(np_, npo, nl, nlo, ni, pn, pnb) < 0 or ni > NUM_FRAMES - 1 or pn > MAX_PRIOR_DIM or pnb > MAX_PRIOR_BLOCKS:
            raise ValueError("corrupt window file header: %s" % path)
        w = Window()
        w.pose = take("<f8", 77).reshape(NUM_FRAMES, 7); w.speedbias = take("<f8", 99).reshape(NUM_FRAMES, 9); w.ex_pose = take("<f8", 7); w.td = float(take("<f8", 1)[0])
        w.inv_depth = take("<f8", np_)
        w.pt_lm, w.pt_fi, w.pt_fj = take("<i4", npo), take("<i4", npo), take("<i4", npo)
        if npo % 2: take("<i4", 1)
        w.pt_pi = take("<f8", 3 * npo).reshape(-1, 3); w.pt_pj = take("<f8", 3 * npo).reshape(-1, 3)
        if has_td:
            w.pt_vel_i = take("<f8", 2 * npo).reshape(-1, 2); w.pt_vel_j = take("<f8", 2 * npo).reshape(-1, 2); w.pt_td_i = take("<f8", npo); w.pt_td_j = take("<f8", npo)
        w.line_orth = take("<f8", 4 * nl).reshape(-1, 4)
        w.ln_lm, w.ln_fj, w.ln_has_vp = take("<i4", nlo), take("<i4", nlo), take("<i4", nlo)
        if nlo % 2: take("<i4", 1)
        w.ln_sp = take("<f8", 3 * nlo).reshape(-1, 3); w.ln_ep = take("<f8", 3 * nlo).reshape(-1, 3); w.ln_vp = take("<f8", 3 * nlo).reshape(-1, 3)
        for _ in range(ni):
            h = take("<f8", 17); jac = take("<f8", 225).reshape(15, 15); cov = take("<f8", 225).reshape(15, 15); fs = take("<i4", 2)
            w.imu.append(dict(sum_dt=float(h[0]), delta_p=h[1:4], delta_q=h[4:8], delta_v=h[8:11], linearized_ba=h[11:14], linearized_bg=h[14:17],
                              jacobian=jac, covariance=cov, frame_i=int(fs[0]), skip=int(fs[1])))
        if pn > 0:
            p = Prior(); p.n = pn; p.n_blocks = pnb
            for name in ("block_kind", "block_frame", "block_size", "block_idx", "x0_off"):
                v = take("<i4", 16)
                for k in range(16): getattr(p, name)[k] = int(v[k])
            x0 = take("<f8", 144); r0 = take("<f8", pn); J0 = take("<f8", pn * pn)
            for k in range(144): p.x0[k] = x0[k]
            C.memmove(p.linearized_residuals, r0.ctypes.data, r0.nbytes); C.memmove(p.linearized_jacobians, J0.ctypes.data, J0.nbytes)
            w.prior = p
        if has_relo:
            nrl, w.relo_frame = (int(v) for v in take("<i4", 2)); w.relo_pose = take("<f8", 7); w.relo_lm = take("<i4", nrl)
            if nrl % 2: take("<i4", 1)
            w.relo_pi = take("<f8", 3 * nrl).reshape(-1, 3); w.relo_pj = take("<f8", 3 * nrl).reshape(-1, 3)
        return w

    def copy(self):
        import copy
        o = Window()
        for name, v in self.__dict__.items():
            if isinstance(v, np.ndarray):
                setattr(o, name, v.copy())
            elif name == "prior":
                o.prior = v.copy() if v is not None else None
            else:
                setattr(o, name, copy.deepcopy(v))
        return o

    def with_state(self, st):
        """New window whose state is `st` (a State), factors unchanged."""
        o = self.copy()
        o.pose = st.pose.copy(); o.speedbias = st.speedbias.copy(); o.ex_pose = st.ex_pose.copy()
        o.inv_depth = st.inv_depth.copy(); o.line_orth = st.line_orth.copy()
        o.td = float(getattr(st, "td", o.td))
        return o


class State:
    """Solver output (para_* arrays after the solve, before double2vector)."""

    def __init__(self, n_points, n_lines):
        self.pose = np.zeros((NUM_FRAMES, 7)); self.speedbias = np.zeros((NUM_FRAMES, 9)); self.ex_pose = np.zeros(7)
        self.td = 0.0
        self.relo_pose = np.zeros(7)
        self.inv_depth = np.zeros(n_points); self.line_orth = np.zeros((n_lines, 4))

    def alloc_c(self):
        s = StateC()
        s.inv_depth = _dp(self.inv_depth)
        s.line_orth = _dp(self.line_orth)
        return s

    def from_c(self, s):
        self.pose = np.array(s.pose).reshape(NUM_FRAMES, 7)
        self.speedbias = np.array(s.speedbias).reshape(NUM_FRAMES, 9)
        self.ex_pose = np.array(s.ex_pose)
        self.td = s.td
        self.relo_pose = np.array(s.relo_pose)
        return self


class Eval:
    def __init__(self, w: Window):
        npo, nlo, ni = len(w.pt_lm), len(w.ln_lm), len(w.imu)
        n = w.prior.n if w.prior is not None else 0
        self.pt_r = np.zeros((npo, 2)); self.pt_J = np.zeros((npo, 2, 19)); self.pt_Jtd = np.zeros((npo, 2))
        self.ln_r = np.zeros((nlo, 2)); self.ln_J = np.zeros((nlo, 2, 10))
        self.vp_r = np.zeros((nlo, 1)); self.vp_J = np.zeros((nlo, 1, 10))
        self.imu_r = np.zeros((ni, 15)); self.imu_J = np.zeros((ni, 15, 30))
        self.prior_r = np.zeros(max(n, 1))
        self.cost = 0.0

    def alloc_c(self):
        e = EvalC()
        for name in ("pt_r", "pt_J", "ln_r", "ln_J", "vp_r", "vp_J", "imu_r", "imu_J", "prior_r", "pt_Jtd"):
            setattr(e, name, _dp(getattr(self, name)))
        return e
