"""Synthetic FRAME SEQUENCE for closed-loop replay (the stand-in for BASELINE configs[4], a rosbag through the unchanged front-end).

A smooth trajectory of `n_frames` keyframe candidates with raw 200 Hz IMU messages, world points / lines / Manhattan vanishing
points that enter and leave the field of view, and per-frame feature messages in the layout the reference's front-end publishes
(points: id -> (x, y, z, u, v, vx, vy), estimator_node.cpp:424-451; lines: id -> the 15-vector of feature_manager.h:32-53).
`save()` writes the file `uvs_host_replay_sequence()` (uv-slam_amd/host/host_capi.cpp) reads.
"""
import numpy as np

from . import abi
from .synth import (FOCAL_LENGTH, _cam, _simulate_frames, ex_pose_euroc, exp_quat, quat_mul, quat_to_R)

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
