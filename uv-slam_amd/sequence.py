"""Synthetic FRAME SEQUENCE for closed-loop replay (the stand-in for BASELINE configs[4], a rosbag through the unchanged front-end).

A smooth trajectory of `n_frames` keyframe candidates with raw 200 Hz IMU messages, world points / lines / Manhattan vanishing
points that enter and leave the field of view, and per-frame feature messages in the layout the reference's front-end publishes
(points: id -> (x, y, z, u, v, vx, vy), estimator_node.cpp:424-451; lines: id -> the 15-vector of feature_manager.h:32-53).
`save()` writes the file `uvs_host_replay_sequence()` (uv-slam_amd/host/host_capi.cpp) reads.
"""
import numpy as np

from . import abi
from .synth import (FOCAL_LENGTH, G, _cam, _simulate_frames, ex_pose_euroc, exp_quat, quat_mul, quat_to_R)

MAGIC = float(0x55565351)


class Sequence:
    pass


def make_sequence(seed=0, n_frames=40, pts_per_frame=11, lines_per_frame=3, pixel_sigma=0.5, perturb=True):
    rng = np.random.default_rng([4242, seed])
    NF = abi.NUM_FRAMES
    ex = ex_pose_euroc()
    samples = []
    Ps, Qs, Vs, ba, bg, blocks = _simulate_frames(rng, n_frames, samples)
    cams = [_cam(Ps[f], Qs[f], ex) for f in range(n_frames)]
    sig = pixel_sigma / FOCAL_LENGTH
    B = np.array([[np.sqrt(2.0 / 3.0), -1.0 / np.sqrt(6.0), -1.0 / np.sqrt(6.0)], [0.0, 1.0 / np.sqrt(2.0), -1.0 / np.sqrt(2.0)],
                  [1.0 / np.sqrt(3.0), 1.0 / np.sqrt(3.0), 1.0 / np.sqrt(3.0)]])
    manhattan = cams[n_frames // 2][0] @ B      # three orthogonal world directions, none near the image plane of the middle camera
    pts = [dict() for _ in range(n_frames)]     # frame -> {id: 7-vector}
    lns = [dict() for _ in range(n_frames)]     # frame -> {id: 15-vector}

    def visible(pc):
        return pc[2] > 0.3 and abs(pc[0] / pc[2]) < 0.75 and abs(pc[1] / pc[2]) < 0.5

    pid = 0
    for s in range(n_frames):
        for _ in range(pts_per_frame):
            Rc, tc = cams[s]
            depth = rng.uniform(2.0, 10.0)
            X = Rc @ (depth * np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.4, 0.4), 1.0])) + tc
            life = int(rng.integers(3, 12))
            for f in range(s, min(n_frames, s + life)):
                Rf, tf = cams[f]
                pc = Rf.T @ (X - tf)
                if not visible(pc): break
                x, y = pc[0] / pc[2] + rng.normal(0, 1) * sig, pc[1] / pc[2] + rng.normal(0, 1) * sig
                pts[f][pid] = np.array([x, y, 1.0, 0.0, 240.0, 0.0, 0.0])
            pid += 1
    lid = 0
    for s in range(n_frames):
        for k in range(lines_per_frame):
            Rc, tc = cams[s]
            depth = rng.uniform(3.0, 8.0)
            Xm = Rc @ (depth * np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), 1.0])) + tc
            tagged = (lid % 4) != 3
            d = manhattan[:, lid % 3].copy() if tagged else rng.normal(0, 1, 3)
            d /= np.linalg.norm(d)
            h = rng.uniform(0.5, 1.5)
            life = int(rng.integers(4, 14))
            for f in range(s, min(n_frames, s + life)):
                Rf, tf = cams[f]
                t1, t2 = rng.uniform(-h, -0.3 * h), rng.uniform(0.3 * h, h)
                a = Rf.T @ (Xm + t1 * d - tf); b = Rf.T @ (Xm + t2 * d - tf)
                if not (visible(a) and visible(b)): break
                m = np.zeros(15)
                m[0:2] = a[:2] / a[2] + rng.normal(0, 1, 2) * sig
                m[2:4] = b[:2] / b[2] + rng.normal(0, 1, 2) * sig
                if tagged:
                    v = Rf.T @ d
                    if abs(v[2]) >= 0.05: m[12:15] = v / v[2]      # vp(2) == 1 <=> "has a vanishing point" (estimator.cpp:920)
                lns[f][lid] = m
            lid += 1
    seq = Sequence()
    seq.n_frames = n_frames
    seq.truth_pose = np.hstack([Ps, Qs]); seq.truth_vel = Vs.copy(); seq.ba, seq.bg = ba, bg
    seq.stamps = np.concatenate([[0.0], np.cumsum([b.sum_dt for b in blocks])])
    seq.samples, seq.points, seq.lines = samples, pts, lns
    pose0 = seq.truth_pose[:NF].copy()
    sb0 = np.hstack([Vs[:NF], np.tile(ba, (NF, 1)), np.tile(bg, (NF, 1))])
    if perturb:       # an imperfect visual-inertial alignment
        for f in range(NF):
            pose0[f, :3] += rng.normal(0, 0.02, 3)
            q = quat_mul(pose0[f, 3:], exp_quat(rng.normal(0, np.deg2rad(0.5), 3))); pose0[f, 3:] = q / np.linalg.norm(q)
            sb0[f, 0:3] += rng.normal(0, 0.05, 3); sb0[f, 3:6] += rng.normal(0, 0.01, 3); sb0[f, 6:9] += rng.normal(0, 0.001, 3)
    seq.pose0, seq.sb0 = pose0, sb0
    return seq


def make_groundtruth_sequence(gt, seed=0, t_start=3.0, t_end=None, frame_stride=20, max_points=150, max_lines=40, pixel_sigma=0.5,
                              acc_sigma=0.02, gyr_sigma=0.002, p_drop=0.01, perturb=True):
    """Frame sequence ALONG A RECORDED GROUND-TRUTH TRAJECTORY (the stand-in for BASELINE configs[4], "EuRoC MH_05_difficult replay"):
    `gt` is the dict `trajectory.read_euroc_groundtruth` / `load_groundtruth_fixture` returns (200 Hz rows: stamp, position, attitude,
    velocity, gyroscope and accelerometer bias -- the layout benchmark_publisher_node.cpp:32-54 parses).

    * keyframe candidates: every `frame_stride`-th ground-truth row from `t_start` seconds after the first row (20 rows = the 10 Hz the
      reference's front-end publishes at, `freq: 10` in euroc_config.yaml); the body trajectory is the C2 cubic spline (positions) /
      rotation spline (attitudes) THROUGH THOSE ROWS, so every frame's true pose is the ground-truth row itself and the ATE of the
      replay can be scored against the recorded file with the reference's own association (trajectory.associate).
    * IMU: one sample per ground-truth row (200 Hz), a = R^T (p'' + g) + b_a + noise, w = w_body + b_g + noise with the RECORDED,
      slowly drifting biases of the rows.
    * camera: a tracker-like front-end -- up to `max_points` points and `max_lines` lines are followed from frame to frame until they
      leave the field of view (or are lost with probability `p_drop` per frame) and replaced by fresh ones in the current view; three
      of four lines are parallel to a world axis of the hall and carry the vanishing point of that axis while it is off the image plane's
      horizon (vp(2) == 1 <=> "has a vanishing point", estimator.cpp:920).
    Messages and file layout as `make_sequence`."""
    from scipy.interpolate import CubicSpline
    from scipy.spatial.transform import Rotation, RotationSpline
    # independent streams, each consumed in time order: a sequence cut short with `t_end` is a PREFIX of the longer one, message for message
    rng, rng_acc, rng_gyr, rng_init = (np.random.default_rng([52525, seed, k]) for k in range(4))
    NF = abi.NUM_FRAMES
    ex = ex_pose_euroc()
    t_abs = np.asarray(gt["t"], dtype=np.float64)
    i0 = int(np.searchsorted(t_abs, t_abs[0] + t_start))
    i1 = len(t_abs) if t_end is None else int(np.searchsorted(t_abs, t_abs[0] + t_end))
    knots = np.arange(i0, len(t_abs), frame_stride)            # the splines always run through the WHOLE recording (a spline is global: cut short, it would differ)
    rows = knots[knots < i1]                                   # ground-truth row of every frame
    n_frames = len(rows)
    if n_frames < NF + 1:
        raise ValueError("ground truth too short: %d frames" % n_frames)
    tk = t_abs[rows] - t_abs[i0]
    q_knots = gt["q_wxyz"][knots][:, [1, 2, 3, 0]]
    q_knots = q_knots / np.linalg.norm(q_knots, axis=1, keepdims=True)      # (the file prints six decimals)
    q_xyzw = q_knots[:n_frames]
    pos = CubicSpline(t_abs[knots] - t_abs[i0], gt["p"][knots])
    rot = RotationSpline(t_abs[knots] - t_abs[i0], Rotation.from_quat(q_knots))
    # ---- 200 Hz IMU on the rows' stamps
    srows = np.arange(rows[0], rows[-1] + 1)
    ts = t_abs[srows] - t_abs[i0]
    Rw = rot(ts).as_matrix()
    acc = np.einsum("nji,nj->ni", Rw, pos(ts, 2) + G) + gt["ba"][srows] + acc_sigma * rng_acc.standard_normal((len(ts), 3))
    gyr = rot(ts, 1) + gt["bw"][srows] + gyr_sigma * rng_gyr.standard_normal((len(ts), 3))      # RotationSpline's rate is the body rate
    samples = [[(0.0, acc[0], gyr[0])]]
    for f in range(1, n_frames):
        a, b = rows[f - 1] - rows[0], rows[f] - rows[0]
        samples.append([(float(ts[k] - ts[k - 1]), acc[k], gyr[k]) for k in range(a + 1, b + 1)])
    # ---- truth at the frames
    Ps = gt["p"][rows].astype(np.float64)
    Qs = q_xyzw.copy()
    Vs = pos(tk, 1)
    cams = [_cam(Ps[f], Qs[f], ex) for f in range(n_frames)]
    sig = pixel_sigma / FOCAL_LENGTH
    half_w, half_h = 0.75, 0.5                                  # normalised-plane half extents of the 752 x 480 image at f = 461.6

    def project(X, f):                                          # world points [n, 3] -> camera frame
        Rc, tc = cams[f]
        return (X - tc) @ Rc

    def in_view(pc):
        z = np.where(pc[:, 2] > 0.3, pc[:, 2], 1.0)
        return (pc[:, 2] > 0.3) & (np.abs(pc[:, 0] / z) < half_w) & (np.abs(pc[:, 1] / z) < half_h)

    pts = [dict() for _ in range(n_frames)]
    lns = [dict() for _ in range(n_frames)]
    p_id = np.zeros(0, np.int64); p_X = np.zeros((0, 3))
    l_id = np.zeros(0, np.int64); l_X = np.zeros((0, 3)); l_d = np.zeros((0, 3)); l_h = np.zeros(0); l_axis = np.zeros(0, np.int64)
    next_pid = next_lid = 0
    for f in range(n_frames):
        Rc, tc = cams[f]
        # points still followed
        if len(p_id):
            keep = in_view(project(p_X, f)) & (rng.random(len(p_id)) >= p_drop)
            p_id, p_X = p_id[keep], p_X[keep]
        n_new = max_points - len(p_id)
        if n_new > 0:
            depth = rng.uniform(2.0, 10.0, n_new)
            ray = np.stack([rng.uniform(-0.9 * half_w, 0.9 * half_w, n_new), rng.uniform(-0.9 * half_h, 0.9 * half_h, n_new), np.ones(n_new)], axis=1)
            p_X = np.vstack([p_X, (ray * depth[:, None]) @ Rc.T + tc])
            p_id = np.concatenate([p_id, next_pid + np.arange(n_new)]); next_pid += n_new
        pc = project(p_X, f)
        xy = pc[:, :2] / pc[:, 2:3] + rng.normal(0, 1, (len(p_id), 2)) * sig
        for k in range(len(p_id)):
            pts[f][int(p_id[k])] = np.array([xy[k, 0], xy[k, 1], 1.0, 0.0, 240.0, 0.0, 0.0])
        # lines: the segment seen this frame is a random piece of the support [-h, h] around the midpoint (detected endpoints slide)
        def segment(n):
            return rng.uniform(-1.0, -0.3, n), rng.uniform(0.3, 1.0, n)
        if len(l_id):
            u1, u2 = segment(len(l_id))
            a = project(l_X + (u1 * l_h)[:, None] * l_d, f); b = project(l_X + (u2 * l_h)[:, None] * l_d, f)
            keep = in_view(a) & in_view(b) & (rng.random(len(l_id)) >= p_drop)
            l_id, l_X, l_d, l_h, l_axis, a, b = l_id[keep], l_X[keep], l_d[keep], l_h[keep], l_axis[keep], a[keep], b[keep]
        else:
            a = b = np.zeros((0, 3))
        tries = 0
        while len(l_id) < max_lines and tries < 8:
            tries += 1
            n_new = max_lines - len(l_id)
            depth = rng.uniform(3.0, 8.0, n_new)
            ray = np.stack([rng.uniform(-0.4, 0.4, n_new), rng.uniform(-0.3, 0.3, n_new), np.ones(n_new)], axis=1)
            Xm = (ray * depth[:, None]) @ Rc.T + tc
            ids = next_lid + np.arange(n_new)
            axis = np.where(ids % 4 != 3, ids % 3, -1)                       # three of four lines follow a world axis of the hall
            d = rng.normal(0, 1, (n_new, 3))
            for k in range(n_new):
                if axis[k] >= 0: d[k] = np.eye(3)[axis[k]]
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            h = rng.uniform(0.5, 1.5, n_new)
            u1, u2 = segment(n_new)
            na = project(Xm + (u1 * h)[:, None] * d, f); nb = project(Xm + (u2 * h)[:, None] * d, f)
            ok = in_view(na) & in_view(nb)
            next_lid += n_new                                                # (ids of rejected candidates are simply not used)
            l_id = np.concatenate([l_id, ids[ok]]); l_X = np.vstack([l_X, Xm[ok]]); l_d = np.vstack([l_d, d[ok]]); l_h = np.concatenate([l_h, h[ok]])
            l_axis = np.concatenate([l_axis, axis[ok]]); a = np.vstack([a, na[ok]]); b = np.vstack([b, nb[ok]])
        for k in range(len(l_id)):
            m = np.zeros(15)
            m[0:2] = a[k, :2] / a[k, 2] + rng.normal(0, 1, 2) * sig
            m[2:4] = b[k, :2] / b[k, 2] + rng.normal(0, 1, 2) * sig
            if l_axis[k] >= 0:
                v = Rc.T @ l_d[k]
                if abs(v[2]) >= 0.05: m[12:15] = v / v[2]
            lns[f][int(l_id[k])] = m
    seq = Sequence()
    seq.n_frames = n_frames
    seq.truth_pose = np.hstack([Ps, Qs]); seq.truth_vel = Vs.copy()
    seq.ba, seq.bg = gt["ba"][rows].copy(), gt["bw"][rows].copy()          # per frame here (the recorded biases drift)
    seq.stamps = t_abs[rows].copy()                                        # absolute stamps: the result file is scored against the recorded rows
    seq.gt_rows = rows
    seq.samples, seq.points, seq.lines = samples, pts, lns
    pose0 = seq.truth_pose[:NF].copy()
    sb0 = np.hstack([Vs[:NF], seq.ba[:NF], seq.bg[:NF]])
    if perturb:       # an imperfect visual-inertial alignment
        for f in range(NF):
            pose0[f, :3] += rng_init.normal(0, 0.02, 3)
            q = quat_mul(pose0[f, 3:], exp_quat(rng_init.normal(0, np.deg2rad(0.5), 3))); pose0[f, 3:] = q / np.linalg.norm(q)
            sb0[f, 0:3] += rng_init.normal(0, 0.05, 3); sb0[f, 3:6] += rng_init.normal(0, 0.01, 3); sb0[f, 6:9] += rng_init.normal(0, 0.001, 3)
    seq.pose0, seq.sb0 = pose0, sb0
    return seq


def save(seq, path):
    out = [MAGIC, float(seq.n_frames)]
    out += list(seq.pose0.ravel()) + list(seq.sb0.ravel())
    for f in range(seq.n_frames):
        out += [float(seq.stamps[f]), float(len(seq.samples[f]))]
        for dt, a, g in seq.samples[f]: out += [dt, *a, *g]
        out.append(float(len(seq.points[f])))
        for i, m in seq.points[f].items(): out += [float(i), *m]
        out.append(float(len(seq.lines[f])))
        for i, m in seq.lines[f].items(): out += [float(i), *m]
    np.asarray(out, dtype=np.float64).tofile(path)


def load_result(path):
    d = np.fromfile(path, dtype=np.float64)
    n = int(d[0])
    r = d[1:1 + 24 * n].reshape(n, 24)
    return dict(frame=r[:, 0].astype(int), flag=r[:, 1].astype(int), P=r[:, 2:5], q=r[:, 5:9], V=r[:, 9:12], ba=r[:, 12:15], bg=r[:, 15:18],
                initial_cost=r[:, 18], final_cost=r[:, 19], iterations=r[:, 20].astype(int), n_points=r[:, 21].astype(int), n_lines=r[:, 22].astype(int),
                status=r[:, 23].astype(int))


def ate(P_est, P_true):
    """RMSE of the positions after the best rigid (rotation + translation, no scale) alignment -- Horn / Umeyama."""
    a, b = P_est - P_est.mean(0), P_true - P_true.mean(0)
    U, _, Vt = np.linalg.svd(a.T @ b)
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = (U @ D @ Vt).T
    e = (R @ a.T).T - b
    return float(np.sqrt((e ** 2).sum(1).mean()))
