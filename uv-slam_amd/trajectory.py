"""Trajectory files either side of the replay: the estimator's result file and the EuRoC ground truth, and the ATE between them.

* result file ("TUM format"): one line per solved frame, `stamp x y z qx qy qz qw`, the file the reference appends to in
  pubOdometry (utility/visualization.cpp:195-207: stamp with 9 decimals, the rest with 6; VINS_RESULT_PATH).  The host mirror's replay
  writes it when UVS_VINS_RESULT_PATH is set (host/host_capi.cpp), `write_tum` writes the same lines from Python.
* ground truth: the EuRoC `state_groundtruth_estimate0/data.csv` layout benchmark_publisher parses
  (benchmark_publisher_node.cpp:32-54: `t[ns], p xyz, q wxyz, v xyz, bw xyz, ba xyz`, 17 comma-separated fields per line, header
  lines starting with '#', t scaled by 1e-9; the copies under benchmark_publisher/config/*/data.csv).
* association as the reference does it (benchmark_publisher_node.cpp:67-75): an estimate stamped t is compared with the LAST
  ground-truth sample whose stamp is <= t; estimates after the last ground-truth stamp are dropped.
* ATE: RMSE of the positions after the best rigid alignment (rotation + translation, no scale), `sequence.ate`.
"""
import numpy as np


def write_tum(path, stamps, positions, quats_xyzw, append=False):
    """visualization.cpp:195-207.  quats_xyzw: rows (qx, qy, qz, qw)."""
    with open(path, "a" if append else "w") as f:
        for t, p, q in zip(stamps, positions, quats_xyzw):
            f.write("%.9f %.6f %.6f %.6f %.6f %.6f %.6f %.6f\n" % (t, p[0], p[1], p[2], q[0], q[1], q[2], q[3]))


def read_tum(path):
    """-> stamps [n], positions [n, 3], quaternions xyzw [n, 4].  Accepts blanks or commas, skips '#' lines."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            v = [float(x) for x in line.replace(",", " ").split()]
            if len(v) < 8:
                raise ValueError("result line with %d fields (want stamp x y z qx qy qz qw): %r" % (len(v), line))
            rows.append(v[:8])
    a = np.asarray(rows, dtype=np.float64).reshape(-1, 8)
    return a[:, 0], a[:, 1:4], a[:, 4:8]


def read_euroc_groundtruth(path):
    """benchmark_publisher_node.cpp:32-54 -> dict(t [s], p [n, 3], q_wxyz [n, 4], v [n, 3], bw [n, 3], ba [n, 3]).
    The reference keeps the 16 value fields as `float`; so does this reader (the association and the ATE then see the same numbers)."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            v = line.split(",")
            if len(v) < 17:
                raise ValueError("ground-truth line with %d fields (want 17): %r" % (len(v), line))
            rows.append([float(x) for x in v[:17]])
    a = np.asarray(rows, dtype=np.float64).reshape(-1, 17)
    val = a[:, 1:].astype(np.float32).astype(np.float64)
    return dict(t=a[:, 0] / 1e9, p=val[:, 0:3], q_wxyz=val[:, 3:7], v=val[:, 7:10], bw=val[:, 10:13], ba=val[:, 13:16])


def load_groundtruth_fixture(path):
    """tests/golden/mh05_groundtruth.npz (written by tests/golden/make_mh05_fixture.py from the reference's
    benchmark_publisher/config/MH_05_difficult/data.csv): the same dict `read_euroc_groundtruth` returns for that CSV, bit for bit --
    integer nanosecond stamps and integer micro-unit values, stored as first differences."""
    z = np.load(path)
    t_ns = np.concatenate([z["t0_ns"], z["t0_ns"][0] + np.cumsum(z["dt_ns"].astype(np.int64))])
    micro = np.vstack([z["v0_micro"], z["v0_micro"][0] + np.cumsum(z["dv_micro"].T.astype(np.int64), axis=0)])
    val = (micro / 1e6).astype(np.float32).astype(np.float64)
    return dict(t=t_ns.astype(np.float64) / 1e9, p=val[:, 0:3], q_wxyz=val[:, 3:7], v=val[:, 7:10], bw=val[:, 10:13], ba=val[:, 13:16])


def write_euroc_groundtruth(path, t, p, q_wxyz, v=None, bw=None, ba=None):
    """The same layout (tests and synthetic sequences): stamps in integer nanoseconds, six decimals like the dataset."""
    n = len(t)
    z = np.zeros((n, 3))
    v = z if v is None else v; bw = z if bw is None else bw; ba = z if ba is None else ba
    with open(path, "w") as f:
        f.write("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z [], v_RS_R_x [m s^-1], v_RS_R_y [m s^-1], v_RS_R_z [m s^-1], "
                "b_w_RS_S_x [rad s^-1], b_w_RS_S_y [rad s^-1], b_w_RS_S_z [rad s^-1], b_a_RS_S_x [m s^-2], b_a_RS_S_y [m s^-2], b_a_RS_S_z [m s^-2]\n")
        for i in range(n):
            vals = list(p[i]) + list(q_wxyz[i]) + list(v[i]) + list(bw[i]) + list(ba[i])
            f.write("%d,%s\n" % (int(round(t[i] * 1e9)), ",".join("%.6f" % x for x in vals)))


def associate(est_stamps, gt_stamps):
    """Index of the last ground-truth sample with stamp <= each estimate's stamp (-1: before the first one) and the mask of the
    estimates the reference would publish a ground-truth pose for (benchmark_publisher_node.cpp:67-75)."""
    idx = np.searchsorted(gt_stamps, est_stamps, side="right") - 1
    keep = (idx >= 0) & (est_stamps <= gt_stamps[-1])
    return idx, keep


def align_rigid(P_est, P_true):
    """Best rotation + translation (no scale) taking P_est onto P_true (Horn / Umeyama); returns (R, t)."""
    ma, mb = P_est.mean(0), P_true.mean(0)
    U, _, Vt = np.linalg.svd((P_est - ma).T @ (P_true - mb))
    D = np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))])
    R = (U @ D @ Vt).T
    return R, mb - R @ ma


def ate(est_path, gt_path):
    """ATE of a result file against an EuRoC ground-truth CSV (or the committed fixture of one, *.npz, or the parsed dict): dict(rmse_m, mean_m, max_m, n_matched, n_estimates)."""
    ts, P, _ = read_tum(est_path)
    gt = gt_path if isinstance(gt_path, dict) else load_groundtruth_fixture(gt_path) if str(gt_path).endswith(".npz") else read_euroc_groundtruth(gt_path)
    idx, keep = associate(ts, gt["t"])
    if keep.sum() < 3:
        raise ValueError("fewer than 3 estimates fall inside the ground truth's time span")
    Pe, Pt = P[keep], gt["p"][idx[keep]]
    R, t = align_rigid(Pe, Pt)
    e = np.linalg.norm((R @ Pe.T).T + t - Pt, axis=1)
    return dict(rmse_m=float(np.sqrt((e ** 2).mean())), mean_m=float(e.mean()), max_m=float(e.max()), n_matched=int(keep.sum()), n_estimates=int(len(ts)))


def result_to_tum(result, stamps, path):
    """The rows of `sequence.load_result` (one per solved frame) as a result file; `stamps[frame]` are the sequence's frame stamps."""
    write_tum(path, [stamps[f] for f in result["frame"]], result["P"], result["q"])
