"""Synthetic window generator: determinism, shape, and the pre-integration restatement."""
import numpy as np

from helpers import uvs, abi, synth


def test_canonical_shape_and_determinism():
    a, b = synth.make_window(21), synth.make_window(21)
    assert len(a.inv_depth) == 150 and len(a.pt_lm) == 750 and len(a.line_orth) == 40 and len(a.ln_lm) == 280
    assert int(a.ln_has_vp.sum()) == 210 and len(a.imu) == 10
    for name in ("pose", "speedbias", "inv_depth", "pt_pj", "line_orth", "ln_sp", "ln_vp"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    c = synth.make_window(22)
    assert not np.array_equal(a.pose, c.pose)
    # reference selection rules: points start before frame WINDOW_SIZE-2 with >= 2 observations; lines have >= LINE_WINDOW observations
    assert a.pt_fi.max() < abi.WINDOW_SIZE - 2 and np.all(a.pt_fj > a.pt_fi)
    assert np.all(np.bincount(a.ln_lm) >= 5)
    assert np.all(a.ln_vp[a.ln_has_vp == 1][:, 2] == 1.0) and np.all(a.ln_vp[a.ln_has_vp == 0] == 0.0)


def test_preintegration_composes_to_truth():
    """Frame states are defined by composing the deltas, so the raw IMU residual at truth is zero (Appendix C (i))."""
    w = synth.make_window(23, noise=False, perturb=False)
    for b in w.imu:
        i = b["frame_i"]; dt = b["sum_dt"]
        Pi, Qi, Vi = w.pose[i, :3], w.pose[i, 3:], w.speedbias[i, :3]
        Pj, Qj, Vj = w.pose[i + 1, :3], w.pose[i + 1, 3:], w.speedbias[i + 1, :3]
        Ri = synth.quat_to_R(Qi)
        assert np.allclose(Ri.T @ (0.5 * synth.G * dt * dt + Pj - Pi - Vi * dt), b["delta_p"], atol=1e-12)
        assert np.allclose(Ri.T @ (synth.G * dt + Vj - Vi), b["delta_v"], atol=1e-12)
        q = synth.quat_mul(Qi, b["delta_q"]); q /= np.linalg.norm(q)
        assert min(np.abs(q - Qj).max(), np.abs(q + Qj).max()) < 1e-12
        cov = np.asarray(b["covariance"]); assert np.allclose(cov, cov.T, atol=1e-18) and np.linalg.eigvalsh(cov).min() > 0
        assert 0.099 < dt < 0.301


def test_line_parameterisation_roundtrip():
    rng = np.random.default_rng(0)
    for _ in range(20):
        A = rng.normal(size=3) * 3 + np.array([0, 0, 5.0]); d = rng.normal(size=3); d /= np.linalg.norm(d)
        a, b, c, phi = synth.line_to_orth(A, d)
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        Rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        U = Rx @ Ry @ Rz
        n = np.cross(A, d)
        assert np.allclose(np.cos(phi) * U[:, 0] / np.sin(phi), n, atol=1e-9)      # (n, d) up to the common scale sin(phi)
        assert np.allclose(U[:, 1], d, atol=1e-12)


def test_algorithmic_bytes_formula():
    w = synth.make_window(0)
    assert synth.algorithmic_bytes(w) == 158880 - 8 * (75 * 75 + 75 + 86)          # no prior: SURVEY 8d minus the prior term
