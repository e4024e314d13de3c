"""Host C++ mirror of the reference API (uv-slam_amd/host/): record/replay file round trip (CPU) and
Estimator::optimization() end to end on the GPU."""
import ctypes as C
import os
import struct
import tempfile

import numpy as np
import pytest

from helpers import uvs, abi, synth, pose_deltas

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "uv-slam_amd", "libuvs_host.so")


def _host():
    lib = C.CDLL(HOST)
    lib.uvs_host_window_probe.argtypes = [C.c_char_p, abi.c_double_p]; lib.uvs_host_window_probe.restype = C.c_int
    lib.uvs_host_replay_window.argtypes = [C.c_char_p, C.c_char_p, C.c_int]; lib.uvs_host_replay_window.restype = C.c_int
    return lib


def test_window_file_roundtrip(oracle):
    marg = lambda win, flag: oracle.marginalize(win, flag)
    w = synth.make_window(31, n_points=25, n_lines=7, n_tagged=5, with_prior=True, marginalize_fn=marg)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.uvsw")
        w.save(path)
        out = np.zeros(12)
        assert _host().uvs_host_window_probe(path.encode(), out.ctypes.data_as(abi.c_double_p)) == 0
    assert list(out[:6]) == [25, len(w.pt_lm), 7, len(w.ln_lm), 10, w.prior.n]
    assert np.isclose(out[6], w.pose.sum()) and np.isclose(out[7], w.pt_pj.sum()) and np.isclose(out[8], (w.ln_vp + w.ln_sp).sum())
    assert np.isclose(out[9], sum((np.asarray(b["covariance"]) * 1e6 + np.asarray(b["jacobian"])).sum() for b in w.imu))
    assert np.isclose(out[10], w.prior.J0().sum()) and out[11] == (w.pt_lm + 3 * w.pt_fi + 7 * w.pt_fj).sum()


def test_window_file_roundtrip_with_relocalization_section():
    """The optional relocalization section (header flag 2): python writer -> C++ reader, and an old-style file still loads."""
    w = synth.add_relocalization(synth.make_window(33, n_points=30, n_lines=5, n_tagged=3), relo_frame=6, seed=33)
    lib = _host(); lib.uvs_host_window_probe_relo.argtypes = [C.c_char_p, abi.c_double_p]; lib.uvs_host_window_probe_relo.restype = C.c_int
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "w.uvsw"); w.save(path)
        out = np.zeros(4); base = np.zeros(12)
        assert lib.uvs_host_window_probe_relo(path.encode(), out.ctypes.data_as(abi.c_double_p)) == 0
        assert lib.uvs_host_window_probe(path.encode(), base.ctypes.data_as(abi.c_double_p)) == 0
        plain = synth.make_window(33, n_points=30, n_lines=5, n_tagged=3); p2 = os.path.join(d, "p.uvsw"); plain.save(p2)
        out0 = np.ones(4)
        assert lib.uvs_host_window_probe_relo(p2.encode(), out0.ctypes.data_as(abi.c_double_p)) == 0
    assert out[0] == len(w.relo_lm) > 0 and out[1] == 6 and out[3] == w.relo_lm.sum()
    assert np.isclose(out[2], w.relo_pose.sum() + (w.relo_pi + 2.0 * w.relo_pj).sum())
    assert base[1] == len(w.pt_lm) and out0[0] == 0


@pytest.mark.gpu
def test_estimator_optimization_matches_direct_solve(gpu_api):
    """Estimator::optimization() (host mirror: uvs::Problem -> uvs_solve_window -> double2vector -> uvs_marginalize) against
    the same window solved directly through the C ABI."""
    s = gpu_api.Solver(max_batch=2)
    marg = lambda win, flag: s.marginalize(win, flag)
    w = synth.make_window(32, with_prior=True, marginalize_fn=marg)
    st, rep = s.solve(w)
    with tempfile.TemporaryDirectory() as d:
        pin, pout = os.path.join(d, "in.uvsw"), os.path.join(d, "out.bin")
        w.save(pin)
        assert _host().uvs_host_replay_window(pin.encode(), pout.encode(), 0) == 0
        raw = np.fromfile(pout, dtype=np.float64)
    status, iters, c0, c1 = raw[:4]
    # same kernel; the inputs differ by one rounding (R -> q -> R in vector2double, 1/(1/lambda)), hence ~1e-14, not bitwise
    assert status == 0 and iters == rep.num_iterations and abs(c0 - rep.initial_cost) <= 1e-9 * c0 and abs(c1 - rep.final_cost) <= 1e-9 * c1
    fr = raw[4:4 + 11 * 16].reshape(11, 16)
    pose = fr[:, :7]
    # double2vector re-anchors yaw + position of frame 0 to their pre-solve values (estimator.cpp:598-648) ...
    assert np.abs(pose[0, :3] - w.pose[0, :3]).max() < 1e-9
    yaw = lambda q: np.arctan2(synth.quat_to_R(q)[1, 0], synth.quat_to_R(q)[0, 0])
    assert abs(yaw(pose[0, 3:]) - yaw(w.pose[0, 3:])) < 1e-9
    # ... and leaves gauge-invariant quantities equal to the raw solver output
    def rel(P):
        R0 = synth.quat_to_R(P[0, 3:])
        return R0.T @ (P[10, :3] - P[0, :3]), R0.T @ synth.quat_to_R(P[10, 3:])
    (ta, Ra), (tb, Rb) = rel(pose), rel(st.pose)
    assert np.abs(ta - tb).max() < 1e-9 and np.abs(Ra - Rb).max() < 1e-9
    assert np.abs(fr[:, 10:16] - st.speedbias[:, 3:]).max() < 1e-9            # biases are gauge independent
    k = 4 + 11 * 16
    dep = raw[k:k + 2 * 150].reshape(150, 2); k += 300
    assert np.allclose(1.0 / dep[:, 0], st.inv_depth, rtol=1e-8) and set(dep[:, 1]) <= {1.0, 2.0}
    lines = raw[k:k + 5 * 40].reshape(40, 5); k += 200
    okl = lines[:, 4] == 1
    assert np.abs(lines[okl, :4] - st.line_orth[okl]).max() < 1e-8 and okl.sum() >= 30
    pn = int(raw[k]); k += 1
    assert pn == 69          # old prior minus Pose[0] (poses 1..9) + SpeedBias[1] + Ex_Pose; Pose[10] is not linked to frame 0 here
    r0 = raw[k:k + pn]; J0 = raw[k + pn:k + pn + pn * pn].reshape(pn, pn)
    assert np.all(np.isfinite(J0)) and np.linalg.matrix_rank(J0) >= 60
    s.close()


@pytest.mark.gpu
def test_factor_classes_evaluate_through_the_gpu(gpu_api, oracle, tmp_path):
    """SURVEY.md 8b "Factor API surface to keep": ProjectionFactor::Evaluate(parameters, residuals, jacobians) (+ check()), IMUFactor /
    MarginalizationFactor::Evaluate, the LineProjectionFactor / VPProjectionFactor functors.  Each call is one uvs_evaluate() of a
    one-block window; the outputs must equal the per-block entries of an evaluation of the WHOLE window (oracle, no loss), global-size
    row-major with a zero 7th pose column."""
    marg = lambda win, flag: oracle.marginalize(win, flag)
    w = synth.make_window(31, with_prior=True, marginalize_fn=marg)
    assert w.ln_has_vp[0] == 1
    path = str(tmp_path / "w.bin"); w.save(path)
    host = C.CDLL(os.path.join(ROOT, "uv-slam_amd", "libuvs_host.so"))
    host.uvs_host_factor_api_probe.argtypes = [C.c_char_p, abi.c_double_p, C.c_int]; host.uvs_host_factor_api_probe.restype = C.c_int
    buf = np.zeros(20000)
    n = host.uvs_host_factor_api_probe(path.encode(), abi._dp(buf), len(buf))
    assert n > 0, n
    e = oracle.evaluate(w, robust=False)
    pos = [0]
    def take(k):
        a = buf[pos[0]:pos[0] + k]; pos[0] += k; return a
    close = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() <= 1e-9 * max(1.0, np.abs(np.asarray(b)).max())
    # point block 0
    assert close(take(2), e.pt_r[0])
    for b in range(3):
        J = take(14).reshape(2, 7)
        assert close(J[:, :6], e.pt_J[0][:, 6 * b:6 * b + 6]) and np.all(J[:, 6] == 0.0)
    assert close(take(2), e.pt_J[0][:, 18])
    fd_gap = take(1)[0]
    assert 0.0 <= fd_gap < 1e-3 * np.abs(e.pt_J[0]).max()          # ProjectionFactor::check(): analytic vs finite differences (eps 1e-6)
    # line block 0 (+ VP)
    assert close(take(2), e.ln_r[0])
    J = take(14).reshape(2, 7); assert close(J[:, :6], e.ln_J[0][:, :6]) and np.all(J[:, 6] == 0.0)
    assert close(take(8).reshape(2, 4), e.ln_J[0][:, 6:])
    assert close(take(1), e.vp_r[0])
    J = take(7); assert close(J[:6], e.vp_J[0][0, :6]) and J[6] == 0.0
    assert close(take(4), e.vp_J[0][0, 6:])
    # IMU block 0
    assert close(take(15), e.imu_r[0])
    for col0, loc, glob in ((0, 6, 7), (6, 9, 9), (15, 6, 7), (21, 9, 9)):
        J = take(15 * glob).reshape(15, glob)
        assert close(J[:, :loc], e.imu_J[0][:, col0:col0 + loc]) and (glob == loc or np.all(J[:, loc:] == 0.0))
    # prior
    p = w.prior
    assert close(take(p.n), e.prior_r[:p.n])
    size0 = p.block_size[0]; J = take(p.n * size0).reshape(p.n, size0)
    assert close(J[:, :6], p.J0()[:, p.block_idx[0]:p.block_idx[0] + 6]) and np.all(J[:, 6] == 0.0)
    assert pos[0] == n
