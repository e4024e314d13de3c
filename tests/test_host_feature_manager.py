"""FeatureManager producers of the host mirror: triangulate (feature_manager.cpp:427-481), triangulateLine (:504-589)
with calcPluckerLine (:827-902).  CPU only: the C hook uvs_host_triangulate runs the mirrored members on a noise-free numpy scene, where
the linear triangulations are exact: depth = z of the point in its start camera, line = the true Pluecker line."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "uv-slam_amd", "libuvs_host.so")


def _quat_R(q):      # xyzw
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _scene(seed, n_pt=40, n_ln=12):
    rng = np.random.default_rng(seed)
    poses = np.zeros((11, 7))
    for i in range(11):
        q = np.array([0.02 * rng.standard_normal(), 0.02 * rng.standard_normal() + 0.01 * i, 0.02 * rng.standard_normal(), 1.0]); q /= np.linalg.norm(q)
        poses[i, :3] = [0.15 * i + 0.02 * rng.standard_normal(), 0.05 * np.sin(i), 0.03 * rng.standard_normal()]
        poses[i, 3:] = q
    qe = np.array([0.01, -0.02, 0.015, 1.0]); qe /= np.linalg.norm(qe)
    ex = np.concatenate([[0.05, -0.02, 0.01], qe])
    Rs = [_quat_R(p[3:]) for p in poses]; ric = _quat_R(qe); tic = ex[:3]
    Rwc = [R @ ric for R in Rs]; twc = [R @ tic + p[:3] for R, p in zip(Rs, poses)]
    cam = lambda i, X: Rwc[i].T @ (X - twc[i])
    pt_start, pt_nobs, pt_obs, depth_true = [], [], [], []
    for _ in range(n_pt):
        s = int(rng.integers(0, 6)); n = int(rng.integers(3, 6))
        X = twc[s] + Rwc[s] @ np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(3, 12)])
        pt_start.append(s); pt_nobs.append(n); depth_true.append(cam(s, X)[2])
        for j in range(s, s + n):
            c = cam(j, X); pt_obs.append(c / c[2])
    ln_start, ln_nobs, ln_sp, ln_ep, plucker_true = [], [], [], [], []
    for _ in range(n_ln):
        s = int(rng.integers(0, 4)); n = int(rng.integers(5, 8))
        A = twc[s] + Rwc[s] @ np.array([rng.uniform(-2, 2), rng.uniform(-1.5, 1.5), rng.uniform(4, 9)])
        d = rng.standard_normal(3); d[2] *= 0.3; d /= np.linalg.norm(d)
        ln_start.append(s); ln_nobs.append(n); plucker_true.append(np.concatenate([np.cross(A, d), d]))
        for j in range(s, s + n):
            a, b = cam(j, A + rng.uniform(-1.0, -0.3) * d), cam(j, A + rng.uniform(0.3, 1.0) * d)     # different endpoints in every frame
            ln_sp.append(a / a[2]); ln_ep.append(b / b[2])
    return dict(poses=poses, ex=ex, pt_start=np.array(pt_start, np.int32), pt_nobs=np.array(pt_nobs, np.int32), pt_obs=np.array(pt_obs),
                ln_start=np.array(ln_start, np.int32), ln_nobs=np.array(ln_nobs, np.int32), ln_sp=np.array(ln_sp), ln_ep=np.array(ln_ep),
                depth_true=np.array(depth_true), plucker_true=np.array(plucker_true))


def _run(sc, depth0=None, orth0=None):
    lib = C.CDLL(LIB)
    dp = lambda a: np.ascontiguousarray(a, np.float64).ctypes.data_as(C.POINTER(C.c_double))
    ip = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data_as(C.POINTER(C.c_int))
    depth = np.full(len(sc["pt_start"]), -1.0) if depth0 is None else depth0.copy()
    orth = np.zeros((len(sc["ln_start"]), 4)) if orth0 is None else orth0.copy()
    keep = [np.ascontiguousarray(sc[k], np.float64) for k in ("poses", "ex", "pt_obs", "ln_sp", "ln_ep")]
    rc = lib.uvs_host_triangulate(dp(keep[0]), dp(keep[1]), len(depth), ip(sc["pt_start"]), ip(sc["pt_nobs"]), dp(keep[2]),
                                  len(orth), ip(sc["ln_start"]), ip(sc["ln_nobs"]), dp(keep[3]), dp(keep[4]),
                                  depth.ctypes.data_as(C.POINTER(C.c_double)), orth.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == 0
    return depth, orth


def _orth_to_plucker(o):
    a, b, c, phi = o
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    U = Rx @ Ry @ Rz
    return np.concatenate([np.cos(phi) * U[:, 0], np.sin(phi) * U[:, 1]])


def test_point_triangulation_recovers_the_start_frame_depth():
    sc = _scene(1)
    depth, _ = _run(sc)
    # points whose start frame is too late are not "used" (start_frame >= WINDOW_SIZE - 2) and keep their value
    used = sc["pt_start"] < 8
    assert np.abs(depth[used] - sc["depth_true"][used]).max() < 1e-8
    assert np.all(depth[~used] == -1.0)


def test_points_with_a_depth_are_left_alone_and_bad_depths_fall_back_to_init_depth():
    sc = _scene(2)
    d0 = np.full(len(sc["pt_start"]), -1.0); d0[:5] = 7.5
    depth, _ = _run(sc, depth0=d0)
    assert np.all(depth[:5] == 7.5)
    # a point BEHIND the start camera triangulates to a negative depth -> INIT_DEPTH (5.0)  (feature_manager.cpp:475-478)
    sc2 = _scene(3, n_pt=1)
    sc2["pt_obs"] = -sc2["pt_obs"]; sc2["pt_obs"][:, 2] = 1.0        # mirrored bearing: intersection behind the cameras
    depth2, _ = _run(sc2)
    assert depth2[0] == 5.0 or depth2[0] >= 0.1


def test_line_triangulation_recovers_the_pluecker_line():
    sc = _scene(4)
    _, orth = _run(sc)
    for o, L in zip(orth, sc["plucker_true"]):
        assert 0.0 <= o[0] <= np.pi + 1e-12          # Eigen's eulerAngles(0,1,2) branch: first angle in [0, pi]
        P = _orth_to_plucker(o)
        Ln = L / np.linalg.norm(L)
        assert min(np.abs(P - Ln).max(), np.abs(P + Ln).max()) < 1e-8
        assert abs(P[:3] @ P[3:]) < 1e-12             # n . d = 0 (Pluecker constraint) by construction of the orthonormal form


def test_lines_with_parameters_are_left_alone():
    sc = _scene(5)
    o0 = np.zeros((len(sc["ln_start"]), 4)); o0[0] = [0.1, 0.2, 0.3, 0.4]
    _, orth = _run(sc, orth0=o0)
    assert np.array_equal(orth[0], o0[0]) and np.all(orth[1:, 3] != 0)
