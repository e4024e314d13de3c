"""BASELINE configs[4] as far as this image allows ("Full EuRoC MH_05_difficult replay ... ATE"): the rosbag, the images and the ROS
front-end are absent, but the one evaluation fixture the reference holds is not -- the MH_05_difficult ground truth that
benchmark_publisher replays (benchmark_publisher/config/MH_05_difficult/data.csv: 22 212 rows at 200 Hz, parsed at
benchmark_publisher_node.cpp:32-54, compared at :67-141).  tests/golden/mh05_groundtruth.npz is that file, row for row
(tests/golden/make_mh05_fixture.py); `sequence.make_groundtruth_sequence` synthesises 200 Hz IMU samples and point / line / vanishing-
point messages ALONG that trajectory (10 Hz keyframe candidates whose true poses are the recorded rows), and the mirrored per-frame
state machine (processIMU / processImage / optimization / marginalization / slideWindow) replays all ~108 s of it closed loop:
more than a thousand chained windows, both marginalization kinds, failureDetection() never firing.  The result file (VINS_RESULT_PATH
layout, visualization.cpp:195-207) is scored against the recorded rows with the reference's association rule (trajectory.ate).

CPU: the fixture reproduces the 40 rows of the CSV kept under tests/golden/ and the loader / writer round trip; the generator is
deterministic; the ORACLE-backed state machine tracks a prefix of the trajectory.
GPU: the HIP-backed product library replays the WHOLE trajectory; a prefix is replayed by the oracle-backed library too and must agree.
"""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import uvs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
seqm, traj = uvs.sequence, uvs.trajectory


@pytest.fixture(scope="module")
def gt():
    return traj.load_groundtruth_fixture(os.path.join(GOLDEN, "mh05_groundtruth.npz"))


def _replay(lib_path, seq, tmp_path, tag, monkeypatch, result_path=None):
    lib = C.CDLL(lib_path)
    lib.uvs_host_replay_sequence.argtypes = [C.c_char_p, C.c_char_p]; lib.uvs_host_replay_sequence.restype = C.c_int
    pin, pout = str(tmp_path / ("seq_%s.bin" % tag)), str(tmp_path / ("out_%s.bin" % tag))
    seqm.save(seq, pin)
    if result_path: monkeypatch.setenv("UVS_VINS_RESULT_PATH", result_path)
    rc = lib.uvs_host_replay_sequence(pin.encode(), pout.encode())      # -4: failureDetection() rebooted the estimator
    if result_path: monkeypatch.delenv("UVS_VINS_RESULT_PATH")
    assert rc == 0
    r = seqm.load_result(pout)
    tm = (C.c_double * 4)()
    lib.uvs_host_replay_timing.argtypes = [C.POINTER(C.c_double)]; lib.uvs_host_replay_timing.restype = None
    lib.uvs_host_replay_timing(tm)
    return r, list(tm)


def test_fixture_is_the_recorded_file(gt, tmp_path):
    # the 40 rows of the same CSV kept as text (tests/golden/euroc_gt_head.csv) are rows 0..38 of the fixture
    head = traj.read_euroc_groundtruth(os.path.join(GOLDEN, "euroc_gt_head.csv"))
    n = len(head["t"])
    assert n >= 30
    for k in head:
        assert np.array_equal(head[k], gt[k][:n]), k
    assert len(gt["t"]) == 22212 and abs((gt["t"][-1] - gt["t"][0]) - 111.055) < 1e-3
    assert np.all(np.diff(gt["t"]) > 0.0049) and np.all(np.diff(gt["t"]) < 0.0051)                 # 200 Hz throughout
    assert np.abs(np.linalg.norm(gt["q_wxyz"], axis=1) - 1).max() < 1e-3      # (six printed decimals, and the recording is not exactly normalised)
    # CSV round trip in the layout benchmark_publisher parses
    p = str(tmp_path / "data.csv")
    traj.write_euroc_groundtruth(p, gt["t"][:500], gt["p"][:500], gt["q_wxyz"][:500], gt["v"][:500], gt["bw"][:500], gt["ba"][:500])
    back = traj.read_euroc_groundtruth(p)
    for k in back:
        assert np.array_equal(back[k], gt[k][:500]), k


def test_groundtruth_sequence_is_deterministic_and_on_the_rows(gt, tmp_path):
    a = seqm.make_groundtruth_sequence(gt, t_end=6.0)
    b = seqm.make_groundtruth_sequence(gt, t_end=6.0)
    seqm.save(a, str(tmp_path / "a.bin")); seqm.save(b, str(tmp_path / "b.bin"))
    assert (tmp_path / "a.bin").read_bytes() == (tmp_path / "b.bin").read_bytes()
    assert a.n_frames == 30 and np.all(np.diff(a.gt_rows) == 20)
    # a sequence cut short is a PREFIX of the longer one, message for message (independent noise streams, splines through the whole recording)
    longer = seqm.make_groundtruth_sequence(gt, t_end=9.0)
    seqm.save(longer, str(tmp_path / "l.bin"))
    A, L = np.fromfile(str(tmp_path / "a.bin")), np.fromfile(str(tmp_path / "l.bin"))
    assert longer.n_frames == 60 and np.array_equal(A[2:], L[2:len(A)])
    assert np.array_equal(a.truth_pose[:, :3], gt["p"][a.gt_rows]) and np.array_equal(a.stamps, gt["t"][a.gt_rows])
    assert all(len(s) == 20 for s in a.samples[1:]) and abs(sum(d for s in a.samples[1:] for d, _, _ in s) - (a.stamps[-1] - a.stamps[0])) < 1e-6
    # the IMU samples integrate back onto the recorded rows (midpoint rule, noise-free): position to millimetres over 2.9 s
    c = seqm.make_groundtruth_sequence(gt, t_end=6.0, acc_sigma=0.0, gyr_sigma=0.0)
    from helpers import uvs as _u
    G, q2R, qmul, qexp = _u.synth.G, _u.synth.quat_to_R, _u.synth.quat_mul, _u.synth.exp_quat
    P, V, q = c.truth_pose[0, :3].copy(), c.truth_vel[0].copy(), c.truth_pose[0, 3:].copy()
    a0, g0 = c.samples[0][0][1], c.samples[0][0][2]
    for f in range(1, c.n_frames):
        ba, bg = c.ba[f], c.bg[f]
        for dt, a1, g1 in c.samples[f]:
            R0 = q2R(q)
            q = qmul(q, qexp((0.5 * (g0 + g1) - bg) * dt)); q /= np.linalg.norm(q)
            am = 0.5 * (R0 @ (a0 - ba) - G + q2R(q) @ (a1 - ba) - G)
            P = P + V * dt + 0.5 * am * dt * dt; V = V + am * dt
            a0, g0 = a1, g1
    assert np.linalg.norm(P - c.truth_pose[-1, :3]) < 5e-3 and np.linalg.norm(V - c.truth_vel[-1]) < 5e-3


def test_oracle_backed_state_machine_tracks_a_prefix(gt, tmp_path, monkeypatch):
    seq = seqm.make_groundtruth_sequence(gt, t_end=6.5)      # 35 frames: 25 chained windows of the hand-held excitation at the start of the recording
    res = str(tmp_path / "vins_result.txt")
    r, _ = _replay(os.path.join(ROOT, "oracle", "libuvs_host_oracle.so"), seq, tmp_path, "oracle", monkeypatch, res)
    assert list(r["frame"]) == list(range(10, seq.n_frames)) and np.all(r["status"] == 0)
    assert r["n_points"].min() >= 120 and r["n_lines"].min() >= 15
    # scored the way benchmark_publisher pairs an estimate with the ground truth (the recorded rows, through the CSV layout it parses)
    gcsv = str(tmp_path / "data.csv")
    traj.write_euroc_groundtruth(gcsv, gt["t"], gt["p"], gt["q_wxyz"], gt["v"], gt["bw"], gt["ba"])
    a = traj.ate(res, gcsv)
    assert a["n_matched"] == len(r["frame"]) and a["rmse_m"] < 0.02
    assert np.abs(r["ba"] - seq.ba[r["frame"]]).max() < 0.05 and np.abs(r["bg"] - seq.bg[r["frame"]]).max() < 0.005


@pytest.mark.gpu
def test_hip_backed_replay_of_the_whole_mh05_trajectory(gt, gpu_api, tmp_path, monkeypatch):
    seq = seqm.make_groundtruth_sequence(gt)
    assert seq.n_frames >= 1075
    res = str(tmp_path / "vins_result.txt")
    rg, tm = _replay(os.path.join(ROOT, "uv-slam_amd", "libuvs_host.so"), seq, tmp_path, "hip", monkeypatch, res)
    n = len(rg["frame"])
    assert n >= 1000 and list(rg["frame"]) == list(range(10, seq.n_frames))          # every frame solved, in order: no reboot of the estimator
    assert np.all(rg["status"] == 0)
    kinds = np.bincount(rg["flag"], minlength=2)
    assert kinds[0] >= 200 and kinds[1] >= 200                                         # MARGIN_OLD and MARGIN_SECOND_NEW both in the hundreds
    assert rg["n_points"].min() >= 100 and rg["n_lines"].min() >= 10
    gcsv = str(tmp_path / "data.csv")
    traj.write_euroc_groundtruth(gcsv, gt["t"], gt["p"], gt["q_wxyz"], gt["v"], gt["bw"], gt["ba"])
    a = traj.ate(res, gcsv)
    assert a["n_matched"] == n and a["rmse_m"] < 0.10, a                              # (oracle-backed run of the same file in the build container: 0.032 m; another noise realisation: 0.018 m)
    assert np.abs(rg["ba"] - seq.ba[rg["frame"]]).max() < 0.05 and np.abs(rg["bg"] - seq.bg[rg["frame"]]).max() < 0.005
    raw = np.linalg.norm(rg["P"] - seq.truth_pose[rg["frame"], :3], axis=1)
    assert raw.max() < 1.0                                                             # drift of the un-aligned odometry over 108 s / ~95 m of path: under 1 % (oracle-backed: 0.48 m)
    # ---- the oracle-backed state machine over a prefix of the same file
    n_pre = 150
    pre = seqm.make_groundtruth_sequence(gt, t_end=3.0 + 0.1 * n_pre + 0.05)
    assert pre.n_frames == n_pre + 1
    ro, _ = _replay(os.path.join(ROOT, "oracle", "libuvs_host_oracle.so"), pre, tmp_path, "oracle", monkeypatch)
    m = len(ro["frame"])
    assert list(ro["frame"]) == list(rg["frame"][:m]) and np.mean(ro["flag"] == rg["flag"][:m]) >= 0.97      # (the keyframe test reads the tracks that survived the outlier rule: it may flip where the runs differ)
    dp = np.linalg.norm(rg["P"][:m] - ro["P"], axis=1)
    # Window 0 has no prior: lock step.  From window 1 on each run carries ITS OWN prior; with the small baseline of the hand-held start a 10-iteration LM is not run
    # to convergence (final costs of the two runs differ in the third digit), so they follow different, equally valid paths a fraction of a millimetre apart
    # (measured: <= 1e-4 m typical, 2e-3 m worst over 190 windows; test_sequence_replay.py has the same statement on its short sequences).
    assert dp[0] < 1e-9 and dp[:20].max() < 1e-3 and dp.max() < 1e-2, (dp[0], dp[:20].max(), dp.max())
    Pt = pre.truth_pose[ro["frame"], :3]
    ag, ao = seqm.ate(rg["P"][:m], Pt), seqm.ate(ro["P"], Pt)
    assert abs(ag - ao) < 1e-3
    print("MH_05_difficult ground-truth trajectory, synthetic measurements: %d chained windows (%d MARGIN_OLD, %d MARGIN_SECOND_NEW), ATE %.4f m (mean %.4f, max %.4f), "
          "un-aligned drift max %.3f m; %.3f ms per optimization() (solve %.3f, marginalization %.3f); first %d windows vs oracle backend: max |dP| %.2e m (first six %.1e), ATE %.4f / %.4f m"
          % (n, kinds[0], kinds[1], a["rmse_m"], a["mean_m"], a["max_m"], raw.max(), tm[0], tm[1], tm[2], m, dp.max(), dp[:6].max(), ag, ao))
