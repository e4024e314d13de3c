"""Pins the CPU oracle's residual blocks against an independent torch-autograd restatement (tests/pyref.py).

The reference ships no tests or golden vectors, so this cross-check -- two restatements written
separately, one with hand-coded/Jet Jacobians (C++), one with autograd (torch) -- is what pins the factor math.
"""
import numpy as np
import pytest
import torch

import pyref
from helpers import uvs, abi, synth

T = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))


@pytest.fixture(scope="module")
def win():
    return synth.make_window(11)


@pytest.fixture(scope="module")
def ev(oracle, win):
    return oracle.evaluate(win, robust=False)


def test_truth_is_a_zero_of_every_family(oracle):
    w = synth.make_window(0, noise=False, perturb=False)
    e = oracle.evaluate(w, robust=False)
    assert np.abs(e.pt_r).max() < 1e-9 and np.abs(e.ln_r).max() < 1e-9 and np.abs(e.imu_r).max() < 1e-6
    assert np.abs(e.vp_r).max() < 1e-5        # acos(|c|) near c=1 amplifies rounding: sqrt(eps) * VP_FACTOR
    assert e.cost < 1e-9


def test_point_factor_matches_autograd(win, ev):
    o = abi.default_options()
    for k in [0, 7, 123, 400, 749]:
        fi, fj, lm = int(win.pt_fi[k]), int(win.pt_fj[k]), int(win.pt_lm[k])
        args = (T(win.pose[fi]), T(win.pose[fj]), T(win.ex_pose), T(win.inv_depth[lm]), T(win.pt_pi[k]), T(win.pt_pj[k]), o.point_sqrt_info)
        r = pyref.point_residual(*args).numpy()
        J = pyref.point_jacobian(*args).numpy()
        assert np.allclose(ev.pt_r[k], r, rtol=1e-11, atol=1e-9)
        assert np.allclose(ev.pt_J[k], J, rtol=1e-9, atol=1e-7)


def test_line_and_vp_factors_match_raw_quaternion_autograd(win, ev):
    o = abi.default_options()
    for k in [0, 5, 100, 279]:
        fj, lm = int(win.ln_fj[k]), int(win.ln_lm[k])
        pose, line, ex = T(win.pose[fj]), T(win.line_orth[lm]), T(win.ex_pose)
        f = lambda p, l: pyref.line_residual(p, l, ex, T(win.ln_sp[k]), T(win.ln_ep[k]), o.line_factor)
        assert np.allclose(ev.ln_r[k], f(pose, line).numpy(), rtol=1e-11, atol=1e-9)
        assert np.allclose(ev.ln_J[k], pyref.raw_jacobian(f, pose, line).numpy(), rtol=1e-9, atol=1e-7)
        if win.ln_has_vp[k]:
            g = lambda p, l: pyref.vp_residual(p, l, ex, T(win.ln_vp[k]), o.vp_factor)
            assert np.allclose(ev.vp_r[k], g(pose, line).numpy(), rtol=1e-9, atol=1e-9)
            assert np.allclose(ev.vp_J[k], pyref.raw_jacobian(g, pose, line).numpy(), rtol=1e-7, atol=1e-6)


def test_quirk_D1_raw_quaternion_jacobian_is_twice_the_tangent_one_at_identity(oracle):
    """the line factor's rotation columns are d r / d(qx,qy,qz), i.e. 2x the tangent derivative at q = I."""
    w = synth.make_window(12)
    w.pose[:, 3:] = np.array([0, 0, 0, 1.0])
    e = oracle.evaluate(w, robust=False)
    o = abi.default_options()
    k = 3
    fj, lm = int(w.ln_fj[k]), int(w.ln_lm[k])
    pose, line, ex = T(w.pose[fj]), T(w.line_orth[lm]), T(w.ex_pose)
    tang = pyref.jac(lambda d: pyref.line_residual(pyref.pose_plus_raw(pose, d), line, ex, T(w.ln_sp[k]), T(w.ln_ep[k]), o.line_factor), torch.zeros(6)).numpy()
    assert np.allclose(e.ln_J[k][:, :3], tang[:, :3], rtol=1e-9, atol=1e-7)          # translation columns agree
    assert np.allclose(e.ln_J[k][:, 3:6], 2.0 * tang[:, 3:6], rtol=1e-8, atol=1e-6)  # rotation columns are doubled


def test_imu_factor(oracle):
    w = synth.make_window(13)
    for f in range(abi.NUM_FRAMES):          # make dbg = 0 so the reference's first-order bias Jacobians are exact
        w.speedbias[f, 6:9] = w.imu[0]["linearized_bg"]
    e = oracle.evaluate(w, robust=False)
    G = T(synth.G)
    for b in [0, 4, 9]:
        blk = {k: T(v) if not np.isscalar(v) else v for k, v in w.imu[b].items() if k not in ("frame_i", "skip", "covariance")}
        i = b
        W = oracle.imu_sqrt_info(w.imu[b]["covariance"])
        cov = np.asarray(w.imu[b]["covariance"])
        assert np.allclose(W.T @ W @ cov, np.eye(15), atol=1e-6)                       # W^T W = cov^-1
        assert np.allclose(W, np.triu(W))                                              # upper triangular (LLT.matrixL().transpose())
        raw = pyref.imu_residual_raw(blk, G, T(w.pose[i]), T(w.speedbias[i]), T(w.pose[i + 1]), T(w.speedbias[i + 1])).numpy()
        assert np.allclose(e.imu_r[b], W @ raw, rtol=1e-9, atol=1e-6)
        z6 = torch.zeros(6)
        Ji = pyref.jac(lambda d: pyref.imu_residual_raw(blk, G, pyref.pose_plus_raw(T(w.pose[i]), d), T(w.speedbias[i]), T(w.pose[i + 1]), T(w.speedbias[i + 1])), z6).numpy()
        Jj = pyref.jac(lambda d: pyref.imu_residual_raw(blk, G, T(w.pose[i]), T(w.speedbias[i]), pyref.pose_plus_raw(T(w.pose[i + 1]), d), T(w.speedbias[i + 1])), z6).numpy()
        Jsi = pyref.jac(lambda s: pyref.imu_residual_raw(blk, G, T(w.pose[i]), s, T(w.pose[i + 1]), T(w.speedbias[i + 1])), T(w.speedbias[i])).numpy()
        Jsj = pyref.jac(lambda s: pyref.imu_residual_raw(blk, G, T(w.pose[i]), T(w.speedbias[i]), T(w.pose[i + 1]), s), T(w.speedbias[i + 1])).numpy()
        Jraw = np.hstack([Ji, Jsi, Jj, Jsj])
        ref = W @ Jraw
        assert np.abs(e.imu_J[b] - ref).max() <= 1e-7 * np.abs(ref).max()


def test_cauchy_corrector_scales_by_sqrt_rho_prime(oracle, win):
    er, en = oracle.evaluate(win, robust=True), oracle.evaluate(win, robust=False)
    for name, a in (("pt", 1.0), ("ln", 0.1)):
        r0, r1 = getattr(en, name + "_r"), getattr(er, name + "_r")
        s = (r0 ** 2).sum(axis=1)
        scale = np.sqrt(1.0 / (1.0 + s / a ** 2))
        assert np.allclose(r1, r0 * scale[:, None], rtol=1e-12)
        J0, J1 = getattr(en, name + "_J"), getattr(er, name + "_J")
        assert np.allclose(J1, J0 * scale[:, None, None], rtol=1e-12)
    cost = 0.5 * (np.log1p((en.pt_r ** 2).sum(1)).sum() + 0.01 * np.log1p((en.ln_r ** 2).sum(1) / 0.01).sum()
                  + np.log1p(en.vp_r[:, 0] ** 2)[win.ln_has_vp == 1].sum() + (en.imu_r ** 2).sum())
    assert abs(er.cost - cost) <= 1e-10 * cost


def test_pose_plus(oracle):
    rng = np.random.default_rng(5)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    x = np.concatenate([rng.normal(size=3), q]); d = rng.normal(size=6) * 0.1
    out = oracle.pose_plus(x, d)
    ref = synth.quat_mul(q, np.array([d[3] / 2, d[4] / 2, d[5] / 2, 1.0])); ref /= np.linalg.norm(ref)
    assert np.allclose(out[:3], x[:3] + d[:3]) and np.allclose(out[3:], ref, atol=1e-15)
