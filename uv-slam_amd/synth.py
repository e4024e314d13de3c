"""Synthetic sliding windows "W10-P150-L40-V3" (SURVEY.md Appendix C / section 8d).

EuRoC-calibrated 11-frame windows with IMU pre-integration blocks, point tracks,
Pluecker line tracks and vanishing-point tags, generated with numpy PCG64
(`seed = 1000 + window_index`).  Ground truth is constructed so that every residual
family is exactly zero at truth when noise is off:
  * frame states are DEFINED by composing the pre-integrated deltas (Appendix C (i));
  * point / line / VP measurements are exact projections of the true landmarks.

The IMU recursion below is the midpoint rule of the reference's IntegrationBase
(vins_estimator/src/factor/integration_base.h:54-158); it is host-side input
preparation, not part of the timed solve.
"""
import numpy as np
from . import abi

# ---- EuRoC constants (config/euroc/euroc_config.yaml:20,31-43,60-64,85-87)
FOCAL_LENGTH = 461.6
RIC_YAML = np.array([[0.0148655429818, -0.999880929698, 0.00414029679422],
                     [0.999557249008, 0.0149672133247, 0.025715529948],
                     [-0.0257744366974, 0.00375618835797, 0.999660727178]])
TIC_YAML = np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])
ACC_N, GYR_N, ACC_W, GYR_W = 0.08, 0.004, 0.00004, 2.0e-6
G_NORM = 9.81007
G = np.array([0.0, 0.0, G_NORM])
IMU_DT = 0.005  # 200 Hz


# ---------------------------------------------------------------- rotation helpers (quaternions stored x,y,z,w)
def quat_mul(a, b):
    ax, ay, az, aw = a; bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def R_to_quat(R):
    """Shepperd's method; returns (x,y,z,w) with w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    q = q / np.linalg.norm(q)
    return q if q[3] >= 0 else -q


def exp_quat(theta):
    a = np.linalg.norm(theta)
    if a < 1e-12:
        q = np.array([theta[0] / 2, theta[1] / 2, theta[2] / 2, 1.0])
        return q / np.linalg.norm(q)
    ax = theta / a
    return np.array([*(np.sin(a / 2) * ax), np.cos(a / 2)])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def ex_pose_euroc():
    """RIC is re-orthonormalised through a quaternion exactly as readParameters does (parameters.cpp:113-115)."""
    q = R_to_quat(RIC_YAML)
    return np.array([*TIC_YAML, *q])


# ---------------------------------------------------------------- IMU pre-integration (integration_base.h)
class PreIntegration:
    """numpy restatement of IntegrationBase (fields integration_base.h:188-203)."""

    def __init__(self, acc_0, gyr_0, ba, bg):
        self.acc_0, self.gyr_0 = np.array(acc_0, float), np.array(gyr_0, float)
        self.linearized_ba, self.linearized_bg = np.array(ba, float), np.array(bg, float)
        self.jacobian = np.eye(15)
        self.covariance = np.zeros((15, 15))
        self.sum_dt = 0.0
        self.delta_p = np.zeros(3)
        self.delta_q = np.array([0.0, 0.0, 0.0, 1.0])
        self.delta_v = np.zeros(3)
        n = np.zeros((18, 18))                                   # :21-27
        n[0:3, 0:3] = ACC_N ** 2 * np.eye(3); n[3:6, 3:6] = GYR_N ** 2 * np.eye(3)
        n[6:9, 6:9] = ACC_N ** 2 * np.eye(3); n[9:12, 9:12] = GYR_N ** 2 * np.eye(3)
        n[12:15, 12:15] = ACC_W ** 2 * np.eye(3); n[15:18, 15:18] = GYR_W ** 2 * np.eye(3)
        self.noise = n

    def push_back(self, dt, acc_1, gyr_1):                        # propagate :130-158 + midPointIntegration :54-128
        acc_1, gyr_1 = np.array(acc_1, float), np.array(gyr_1, float)
        ba, bg = self.linearized_ba, self.linearized_bg
        dq, dp, dv = self.delta_q, self.delta_p, self.delta_v
        Rq = quat_to_R(dq)
        un_acc_0 = Rq @ (self.acc_0 - ba)
        un_gyr = 0.5 * (self.gyr_0 + gyr_1) - bg
        rq = quat_mul(dq, np.array([un_gyr[0] * dt / 2, un_gyr[1] * dt / 2, un_gyr[2] * dt / 2, 1.0]))
        Rr = quat_to_R_raw(rq)
        un_acc_1 = Rr @ (acc_1 - ba)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        w_x = un_gyr
        a0x, a1x = self.acc_0 - ba, acc_1 - ba
        Rw, Ra0, Ra1 = skew(w_x), skew(a0x), skew(a1x)
        I3 = np.eye(3)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I3
        F[0:3, 3:6] = -0.25 * Rq @ Ra0 * dt * dt + -0.25 * Rr @ Ra1 @ (I3 - Rw * dt) * dt * dt
        F[0:3, 6:9] = I3 * dt
        F[0:3, 9:12] = -0.25 * (Rq + Rr) * dt * dt
        F[0:3, 12:15] = -0.25 * Rr @ Ra1 * dt * dt * -dt
        F[3:6, 3:6] = I3 - Rw * dt
        F[3:6, 12:15] = -1.0 * I3 * dt
        F[6:9, 3:6] = -0.5 * Rq @ Ra0 * dt + -0.5 * Rr @ Ra1 @ (I3 - Rw * dt) * dt
        F[6:9, 6:9] = I3
        F[6:9, 9:12] = -0.5 * (Rq + Rr) * dt
        F[6:9, 12:15] = -0.5 * Rr @ Ra1 * dt * -dt
        F[9:12, 9:12] = I3
        F[12:15, 12:15] = I3
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * Rq * dt * dt
        V[0:3, 3:6] = 0.25 * -Rr @ Ra1 * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * Rr * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I3 * dt
        V[3:6, 9:12] = 0.5 * I3 * dt
        V[6:9, 0:3] = 0.5 * Rq * dt
        V[6:9, 3:6] = 0.5 * -Rr @ Ra1 * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * Rr * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I3 * dt
        V[12:15, 15:18] = I3 * dt
        self.jacobian = F @ self.jacobian
        self.covariance = F @ self.covariance @ F.T + V @ self.noise @ V.T
        self.delta_p, self.delta_v = rp, rv
        self.delta_q = rq / np.linalg.norm(rq)                    # :153
        self.sum_dt += dt
        self.acc_0, self.gyr_0 = acc_1, gyr_1

    def as_block(self, frame_i):
        return dict(sum_dt=self.sum_dt, delta_p=self.delta_p.copy(), delta_q=self.delta_q.copy(), delta_v=self.delta_v.copy(),
                    linearized_ba=self.linearized_ba.copy(), linearized_bg=self.linearized_bg.copy(),
                    jacobian=self.jacobian.copy(), covariance=self.covariance.copy(), frame_i=frame_i,
                    skip=int(self.sum_dt > 10.0))


def quat_to_R_raw(q):
    """Eigen toRotationMatrix() polynomial on a possibly non-unit quaternion (result_delta_q before normalisation)."""
    return quat_to_R(q)


# ---------------------------------------------------------------- line helpers (Pluecker convention)
def line_to_orth(A, d):
    """World line through A with unit direction d -> (psi_x, psi_y, psi_z, phi)."""
    d = d / np.linalg.norm(d)
    n = np.cross(A, d)
    nn = np.linalg.norm(n)
    U = np.stack([n / nn, d, np.cross(n / nn, d)], axis=1)
    b = np.arcsin(np.clip(U[0, 2], -1, 1))
    a = np.arctan2(-U[1, 2], U[2, 2])
    c = np.arctan2(-U[0, 1], U[0, 0])
    phi = np.arctan2(1.0, nn)
    return np.array([a, b, c, phi])


# ---------------------------------------------------------------- generator
def make_window(index=0, n_points=150, n_lines=40, n_tagged=30, pt_track=6, ln_track=7, noise=True, perturb=True,
                seed_base=1000, with_prior=False, marginalize_fn=None, n_frames_before=1, pixel_sigma=0.5,
                pt_start_mod=None, ln_start_mod=None):
    """Canonical window for `index` (seed = seed_base + index).

    with_prior=True needs `marginalize_fn(window, flag) -> abi.Prior`; the prior is
    then the result of marginalizing the oldest frame of the PREVIOUS window
    (same trajectory extended one frame into the past), Appendix C.
    """
    for attempt in range(50):
        rng = np.random.default_rng([seed_base + index, attempt])
        try:
            return _make(rng, n_points, n_lines, n_tagged, pt_track, ln_track, noise, perturb, with_prior, marginalize_fn,
                         n_frames_before, pixel_sigma, pt_start_mod, ln_start_mod)
        except _Regenerate:
            continue
    raise RuntimeError("could not generate a valid window")


class _Regenerate(Exception):
    pass


def _simulate_frames(rng, n_total, samples=None):
    """Returns per-frame truth (P,Q,V), true biases, and the IMU blocks linking consecutive frames.
    `samples` (a list) receives, per frame, the raw (dt, acc, gyr) messages a front-end would feed Estimator::processIMU."""
    ba = rng.normal(0, 0.02, 3)
    bg = rng.normal(0, 0.002, 3)
    # smooth excitation
    gf = rng.uniform(0.2, 0.6, 3); gph = rng.uniform(0, 2 * np.pi, 3); gam = rng.uniform(0.1, 0.3, 3)
    af = rng.uniform(0.2, 0.6, 3); aph = rng.uniform(0, 2 * np.pi, 3); aam = rng.uniform(0.2, 0.6, 3)
    gyr_true = lambda t: gam * np.sin(2 * np.pi * gf * t + gph)
    acc_world = lambda t: aam * np.sin(2 * np.pi * af * t + aph)
    yaw = rng.uniform(-np.pi, np.pi)
    q = quat_mul(exp_quat(np.array([0, 0, yaw])), exp_quat(rng.normal(0, 0.05, 3)))
    heading = rng.uniform(-np.pi, np.pi)
    P = np.zeros(3); V = np.array([np.cos(heading), np.sin(heading), rng.normal(0, 0.1)])
    Ps, Qs, Vs, blocks = [P.copy()], [q.copy()], [V.copy()], []
    t = 0.0
    q_sim = q.copy()     # orientation used only to synthesise accelerometer samples
    for f in range(n_total - 1):
        steps = int(rng.integers(20, 61))     # 0.10 .. 0.30 s at 200 Hz
        meas = lambda tt, qq: (quat_to_R(qq).T @ (acc_world(tt) + G) + ba, gyr_true(tt) + bg)
        a0, g0 = meas(t, q_sim)
        pre = PreIntegration(a0, g0, ba, bg)
        if samples is not None:
            if f == 0: samples.append([(0.0, a0, g0)])
            samples.append([])
        for _ in range(steps):
            w_mid = gyr_true(t + 0.5 * IMU_DT)
            q_sim = quat_mul(q_sim, exp_quat(w_mid * IMU_DT)); q_sim /= np.linalg.norm(q_sim)
            t += IMU_DT
            a1, g1 = meas(t, q_sim)
            pre.push_back(IMU_DT, a1, g1)
            if samples is not None: samples[-1].append((IMU_DT, a1, g1))
        Ri = quat_to_R(Qs[-1]); dt = pre.sum_dt
        Pn = Ps[-1] + Vs[-1] * dt - 0.5 * G * dt * dt + Ri @ pre.delta_p
        Vn = Vs[-1] - G * dt + Ri @ pre.delta_v
        Qn = quat_mul(Qs[-1], pre.delta_q); Qn /= np.linalg.norm(Qn)
        q_sim = Qn.copy()
        Ps.append(Pn); Vs.append(Vn); Qs.append(Qn)
        blocks.append(pre)
    return np.array(Ps), np.array(Qs), np.array(Vs), ba, bg, blocks


def _cam(P, Q, ex):
    R_wb = quat_to_R(Q); ric = quat_to_R(ex[3:]); tic = ex[:3]
    return R_wb @ ric, R_wb @ tic + P


def _build(rng, Ps, Qs, Vs, ba, bg, blocks, ex, first, n_points, n_lines, n_tagged, pt_track, ln_track, noise, perturb,
           pixel_sigma, pt_start_mod, ln_start_mod, manhattan, long_tracks=0):
    """Window over frames first..first+10 of the simulated trajectory."""
    NF = abi.NUM_FRAMES
    sig = pixel_sigma / FOCAL_LENGTH if noise else 0.0
    Pw, Qw, Vw = Ps[first:first + NF], Qs[first:first + NF], Vs[first:first + NF]
    cams = [_cam(Pw[f], Qw[f], ex) for f in range(NF)]
    w = abi.Window()
    w.ex_pose = ex.copy()
    # ---- truth state
    truth = dict(pose=np.hstack([Pw, Qw]), speedbias=np.hstack([Vw, np.tile(ba, (NF, 1)), np.tile(bg, (NF, 1))]))
    # ---- points
    pt_start_mod = pt_start_mod or (NF - pt_track + 1)
    inv_depth, lm, fi, fj, pi, pj = [], [], [], [], [], []
    for k in range(n_points + long_tracks):
        # the last `long_tracks` landmarks are anchored at frame 0 and tracked through the whole window: they are what
        # makes the marginalization prior of the NEXT window span Pose[0..9] (n = 75), as in a real VINS window
        trk = pt_track if k < n_points else NF
        s = k % pt_start_mod if k < n_points else 0
        for _ in range(100):
            xy = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.4, 0.4)]); depth = rng.uniform(2.0, 10.0)
            Rc, tc = cams[s]
            X = Rc @ (depth * np.array([xy[0], xy[1], 1.0])) + tc
            obs, ok = [], True
            for f in range(s, s + trk):
                Rf, tf = cams[f]
                pc = Rf.T @ (X - tf)
                if pc[2] < 0.2: ok = False; break
                obs.append(np.array([pc[0] / pc[2] + rng.normal(0, 1) * sig, pc[1] / pc[2] + rng.normal(0, 1) * sig, 1.0]))
            if ok: break
        else:
            raise _Regenerate()
        inv_depth.append(1.0 / depth)
        for o in range(1, trk):
            lm.append(k); fi.append(s); fj.append(s + o); pi.append(obs[0]); pj.append(obs[o])
    # ---- lines
    ln_start_mod = ln_start_mod or (NF - ln_track + 1)
    orth, llm, lfj, lsp, lep, lhas, lvp = [], [], [], [], [], [], []
    for l in range(n_lines):
        s = l % ln_start_mod
        tagged = l < n_tagged
        for _ in range(200):
            c = s + ln_track // 2
            Rc, tc = cams[c]
            depth = rng.uniform(3.0, 8.0)
            Xm = Rc @ (depth * np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), 1.0])) + tc
            if tagged:
                d = manhattan[:, l % 3].copy()
            else:
                d = rng.normal(0, 1, 3); d /= np.linalg.norm(d)
            h = rng.uniform(0.5, 1.5)
            ok, rows = True, []
            for f in range(s, s + ln_track):
                Rf, tf = cams[f]
                t1, t2 = rng.uniform(-h, -0.3 * h), rng.uniform(0.3 * h, h)
                a = Rf.T @ (Xm + t1 * d - tf); b = Rf.T @ (Xm + t2 * d - tf)
                if a[2] < 0.2 or b[2] < 0.2: ok = False; break
                sp = np.array([a[0] / a[2] + rng.normal(0, 1) * sig, a[1] / a[2] + rng.normal(0, 1) * sig, 1.0])
                ep = np.array([b[0] / b[2] + rng.normal(0, 1) * sig, b[1] / b[2] + rng.normal(0, 1) * sig, 1.0])
                vp = np.zeros(3)
                if tagged:
                    v = Rf.T @ d
                    if abs(v[2]) < 0.05: raise _Regenerate()
                    vp = v / v[2]
                rows.append((f, sp, ep, vp))
            if ok: break
        else:
            raise _Regenerate()
        orth.append(line_to_orth(Xm, d))
        for f, sp, ep, vp in rows:
            llm.append(l); lfj.append(f); lsp.append(sp); lep.append(ep); lhas.append(int(tagged)); lvp.append(vp)
    truth["inv_depth"] = np.array(inv_depth); truth["line_orth"] = np.array(orth).reshape(-1, 4)
    w.truth = truth
    w.pt_lm, w.pt_fi, w.pt_fj = np.array(lm, np.int32), np.array(fi, np.int32), np.array(fj, np.int32)
    w.pt_pi, w.pt_pj = np.array(pi).reshape(-1, 3), np.array(pj).reshape(-1, 3)
    w.ln_lm, w.ln_fj = np.array(llm, np.int32), np.array(lfj, np.int32)
    w.ln_sp, w.ln_ep = np.array(lsp).reshape(-1, 3), np.array(lep).reshape(-1, 3)
    w.ln_has_vp, w.ln_vp = np.array(lhas, np.int32), np.array(lvp).reshape(-1, 3)
    w.imu = [blocks[first + f].as_block(f) for f in range(NF - 1)]
    # ---- initial state = truth (+) perturbation
    w.pose = truth["pose"].copy(); w.speedbias = truth["speedbias"].copy()
    w.inv_depth = truth["inv_depth"].copy(); w.line_orth = truth["line_orth"].copy()
    if perturb:
        for f in range(NF):
            w.pose[f, :3] += rng.normal(0, 0.02, 3)
            q = quat_mul(w.pose[f, 3:], exp_quat(rng.normal(0, np.deg2rad(0.5), 3)))
            w.pose[f, 3:] = q / np.linalg.norm(q)
            w.speedbias[f, 0:3] += rng.normal(0, 0.05, 3)
            w.speedbias[f, 3:6] += rng.normal(0, 0.01, 3)
            w.speedbias[f, 6:9] += rng.normal(0, 0.001, 3)
        w.inv_depth = w.inv_depth * (1.0 + rng.normal(0, 0.1, len(w.inv_depth)))
        w.line_orth = w.line_orth + rng.normal(0, 0.02, w.line_orth.shape)
    return w


def _make(rng, n_points, n_lines, n_tagged, pt_track, ln_track, noise, perturb, with_prior, marginalize_fn, n_before,
          pixel_sigma, pt_start_mod, ln_start_mod):
    NF = abi.NUM_FRAMES
    ex = ex_pose_euroc()
    n_before = n_before if with_prior else 0
    Ps, Qs, Vs, ba, bg, blocks = _simulate_frames(rng, NF + n_before)
    # three orthogonal world VP directions, oriented so that none is (near) perpendicular to the optical axis of
    # the window's middle camera: each axis starts 54.7 deg off the axis (v_z = 1/sqrt(3)), random spin about it.
    B = np.array([[np.sqrt(2.0 / 3.0), -1.0 / np.sqrt(6.0), -1.0 / np.sqrt(6.0)],
                  [0.0, 1.0 / np.sqrt(2.0), -1.0 / np.sqrt(2.0)],
                  [1.0 / np.sqrt(3.0), 1.0 / np.sqrt(3.0), 1.0 / np.sqrt(3.0)]])
    spin = quat_to_R(exp_quat(np.array([0, 0, rng.uniform(-np.pi, np.pi)])))
    Rc_mid, _ = _cam(Ps[n_before + NF // 2], Qs[n_before + NF // 2], ex)
    manhattan = Rc_mid @ spin @ B
    args = (n_points, n_lines, n_tagged, pt_track, ln_track, noise)
    prior = None
    if with_prior:
        if marginalize_fn is None:
            raise ValueError("with_prior=True needs marginalize_fn")
        for first in range(n_before):
            prev = _build(rng, Ps, Qs, Vs, ba, bg, blocks, ex, first, *args, perturb, pixel_sigma, pt_start_mod, ln_start_mod, manhattan,
                          long_tracks=12)
            if perturb:   # the previous estimate is closer to truth than a fresh initial guess
                t = prev.truth
                prev.pose[:, :3] = t["pose"][:, :3] + 0.3 * (prev.pose[:, :3] - t["pose"][:, :3])
                prev.speedbias = t["speedbias"] + 0.3 * (prev.speedbias - t["speedbias"])
                prev.inv_depth = t["inv_depth"] + 0.3 * (prev.inv_depth - t["inv_depth"])
                prev.line_orth = t["line_orth"] + 0.3 * (prev.line_orth - t["line_orth"])
            prev.prior = prior
            prior = marginalize_fn(prev, 0)
    w = _build(rng, Ps, Qs, Vs, ba, bg, blocks, ex, n_before, *args, perturb, pixel_sigma, pt_start_mod, ln_start_mod, manhattan)
    w.prior = prior
    return w


def algorithmic_bytes(w: abi.Window):
    """B_in + B_out of SURVEY.md section 8d for window w (FP64, inputs read once, state written once)."""
    n_po, n_lo = len(w.pt_lm), len(w.ln_lm)
    n_vp = int(np.sum(w.ln_has_vp))
    F, n_p, n_l, n_imu = abi.NUM_FRAMES, len(w.inv_depth), len(w.line_orth), len(w.imu)
    n = w.prior.n if w.prior is not None else 0
    b_in = 8 * (6 * n_po + 6 * n_lo + 3 * n_vp) + 4 * (3 * n_po + 2 * n_lo + 2 * n_vp) + 8 * (16 * F + 7 + n_p + 4 * n_l) + 8 * 467 * n_imu
    if n:
        b_in += 8 * (n * n + n + 86)
    b_out = 8 * (16 * F + n_p + 4 * n_l)
    return b_in + b_out


def shard_landmarks(w: abi.Window, rank: int, world: int):
    """Sub-window holding the landmarks k with k % world == rank (frames / IMU / prior replicated) -- SURVEY.md section 8e."""
    o = w.copy()
    pk = np.arange(len(w.inv_depth)) % world == rank
    lk = np.arange(len(w.line_orth)) % world == rank
    pmap = np.cumsum(pk) - 1; lmap = np.cumsum(lk) - 1
    po = pk[w.pt_lm] if len(w.pt_lm) else np.zeros(0, bool)
    lo = lk[w.ln_lm] if len(w.ln_lm) else np.zeros(0, bool)
    o.inv_depth = w.inv_depth[pk]; o.line_orth = w.line_orth[lk]
    o.pt_lm = pmap[w.pt_lm[po]].astype(np.int32); o.pt_fi = w.pt_fi[po]; o.pt_fj = w.pt_fj[po]; o.pt_pi = w.pt_pi[po]; o.pt_pj = w.pt_pj[po]
    o.ln_lm = lmap[w.ln_lm[lo]].astype(np.int32); o.ln_fj = w.ln_fj[lo]; o.ln_sp = w.ln_sp[lo]; o.ln_ep = w.ln_ep[lo]
    o.ln_has_vp = w.ln_has_vp[lo]; o.ln_vp = w.ln_vp[lo]
    return o, np.nonzero(pk)[0], np.nonzero(lk)[0]


def add_time_offset(w, td_true=0.004, vel_sigma=0.6, seed=0):
    """Turns `w` into an ESTIMATE_TD window (ProjectionTdFactor, projection_td_factor.cpp:34-145): every point observation gets an
    image-plane velocity (normalised units / s; the anchor observation's is shared by the landmark, it is feature_per_frame[0].velocity),
    cur_td = 0, and the stored observations are displaced by td_true * velocity, so that the factor's time shift
    pts - (td - cur_td) * velocity recovers the original projection exactly at td = td_true.  The state starts at td = 0."""
    rng = np.random.default_rng([seed, 77])
    o = w.copy()
    n = len(o.pt_lm)
    v_lm = vel_sigma * rng.standard_normal((len(o.inv_depth), 2))
    o.pt_vel_i = v_lm[o.pt_lm].copy()
    o.pt_vel_j = vel_sigma * rng.standard_normal((n, 2))
    o.pt_td_i = np.zeros(n); o.pt_td_j = np.zeros(n)
    o.pt_pi = o.pt_pi.copy(); o.pt_pj = o.pt_pj.copy()
    o.pt_pi[:, :2] += td_true * o.pt_vel_i
    o.pt_pj[:, :2] += td_true * o.pt_vel_j
    o.td = 0.0
    if o.truth is not None:
        o.truth = dict(o.truth); o.truth["td"] = td_true
    return o


def add_relocalization(w, relo_frame=4, fraction=0.6, offset=(0.25, 4.0), pixel_sigma=0.0, seed=0):
    """Adds the relocalization blocks of estimator.cpp:944-978 to `w`: an old keyframe whose true pose is the truth of frame `relo_frame`
    moved by `offset` = (metres, degrees) in a random direction re-observes `fraction` of the landmarks that start at or before
    `relo_frame` (the reference's `start <= relo_frame_local_index`); pts_j = its normalised observation (+ noise), pts_i = the
    landmark's first observation.  relo_Pose starts at the window's current Pose[relo_frame] (Estimator::setReloFrame,
    estimator.cpp:1361-1379); truth["relo_pose"] holds the pose that zeroes the blocks of a noise-free window."""
    rng = np.random.default_rng([seed, 91])
    o = w.copy()
    t = o.truth
    focal = 461.6
    d = rng.standard_normal(3); d /= np.linalg.norm(d)
    a = rng.standard_normal(3); a /= np.linalg.norm(a)
    P = t["pose"][relo_frame, :3] + offset[0] * d
    Q = quat_mul(t["pose"][relo_frame, 3:], exp_quat(np.deg2rad(offset[1]) * a)); Q = Q / np.linalg.norm(Q)
    Rr, tr = _cam(P, Q, o.ex_pose)
    lms, pis, pjs = [], [], []
    first = np.r_[True, o.pt_lm[1:] != o.pt_lm[:-1]] if len(o.pt_lm) else np.zeros(0, bool)
    for k in np.nonzero(first)[0]:
        lm, fi = int(o.pt_lm[k]), int(o.pt_fi[k])
        if fi > relo_frame or rng.uniform() > fraction: continue
        Ri, ti = _cam(t["pose"][fi, :3], t["pose"][fi, 3:], o.ex_pose)
        X = Ri @ (o.pt_pi[k] / t["inv_depth"][lm]) + ti
        pc = Rr.T @ (X - tr)
        if pc[2] < 0.2: continue
        lms.append(lm); pis.append(o.pt_pi[k].copy())
        pjs.append(np.array([pc[0] / pc[2] + rng.normal(0, 1) * pixel_sigma / focal, pc[1] / pc[2] + rng.normal(0, 1) * pixel_sigma / focal, 1.0]))
    o.relo_lm = np.array(lms, np.int32); o.relo_pi = np.array(pis).reshape(-1, 3); o.relo_pj = np.array(pjs).reshape(-1, 3)
    o.relo_pose = o.pose[relo_frame].copy(); o.relo_frame = int(relo_frame)
    o.truth = dict(t); o.truth["relo_pose"] = np.r_[P, Q]
    return o
