"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the CPU oracle).

CPU leg: the oracle still reproduces them (regression pin).  GPU leg: the HIP solver reproduces them through the C ABI
without anything from /root/reference or the oracle library being needed at run time for the comparison itself.
"""
import glob
import importlib.util
import os

import numpy as np
import pytest

from helpers import uvs, abi, pose_deltas

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")) if not p.endswith(("config3_trace.npz", "mh05_groundtruth.npz")))      # (window fixtures; the other two are a trace and a trajectory)


def load(name):
    d = dict(np.load(os.path.join(HERE, "golden", name + ".npz")))
    return d, mg.dict_to_window(d)


def test_fixtures_exist():
    assert set(CASES) >= {"small_noprior", "small_prior", "points_only", "small_relo", "canonical_prior", "canonical_vp_heavy"}
    assert os.path.exists(os.path.join(HERE, "golden", "config3_trace.npz"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(oracle, name):
    d, w = load(name)
    st, rep = oracle.solve(w)
    k = rep.num_iterations + 1
    assert k == len(d["out_cost"]) and int(d["out_termination"]) == rep.termination
    assert list(rep.accepted[:k]) == list(d["out_accepted"])
    assert np.allclose(np.array(rep.cost[:k]), d["out_cost"], rtol=1e-9)
    dp, da = pose_deltas(st.pose, d["out_pose"])
    assert dp < 1e-8 and da < 1e-7
    if "relo_lm" in d: assert np.abs(st.relo_pose - d["out_relo_pose"]).max() < 1e-8
    ev = oracle.evaluate(w, robust=True)
    assert np.allclose(ev.pt_r, d["ev_pt_r"], rtol=1e-10, atol=1e-10) and np.allclose(ev.imu_r, d["ev_imu_r"], rtol=1e-9, atol=1e-7)
    assert np.allclose(ev.ln_J[:4], d["ev_ln_J0"], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_solver_reproduces_golden(gpu_api, name):
    d, w = load(name)
    s = gpu_api.Solver(max_batch=4)
    st, rep = s.solve(w)
    ev = s.evaluate(w, robust=True)
    k = rep.num_iterations + 1
    assert rep.status == 0 and k == len(d["out_cost"]) and list(rep.accepted[:k]) == list(d["out_accepted"])
    assert abs(rep.final_cost - float(d["out_final_cost"])) <= 1e-6 * float(d["out_final_cost"])
    dp, da = pose_deltas(st.pose, d["out_pose"])
    assert dp < 1e-4 and da < 1e-4                      # north_star tolerance: 1e-4 m / 1e-4 rad
    assert np.abs(st.speedbias - d["out_speedbias"]).max() < 1e-4
    if "relo_lm" in d: assert np.abs(st.relo_pose - d["out_relo_pose"]).max() < 1e-4 and not np.array_equal(st.relo_pose, w.relo_pose)
    assert np.abs(st.inv_depth - d["out_inv_depth"]).max() < 1e-4 and (len(d["out_line_orth"]) == 0 or np.abs(st.line_orth - d["out_line_orth"]).max() < 1e-4)
    assert abs(ev.cost - float(d["ev_cost"])) <= 1e-9 * float(d["ev_cost"])
    assert np.allclose(ev.pt_r, d["ev_pt_r"], rtol=1e-9, atol=1e-9) and np.allclose(ev.imu_r, d["ev_imu_r"], rtol=1e-8, atol=1e-6)
    if "marg_n" in d:                                   # marginalization: compare the information form (J0 itself is sign/order ambiguous)
        p = s.marginalize(w.with_state(st), 0)
        assert p.n == int(d["marg_n"])
        A = p.J0().T @ p.J0(); b = p.J0().T @ p.r0()
        assert np.abs(A - d["marg_A"]).max() <= 1e-6 * np.abs(d["marg_A"]).max()
        assert np.abs(b - d["marg_b"]).max() <= 1e-6 * max(1.0, np.abs(d["marg_b"]).max())
    s.close()


# ---- configs[3] (20 000 points + 5 000 lines): generator arguments + expected trace instead of 10 MB of inputs
def _config3():
    d = dict(np.load(os.path.join(HERE, "golden", "config3_trace.npz")))
    idx, n_points, n_lines, n_tagged = (int(v) for v in d["synth_args"])
    w = uvs.synth.make_window(idx, n_points=n_points, n_lines=n_lines, n_tagged=n_tagged)
    chk = np.array([w.pt_pj.sum(), w.ln_sp.sum(), w.inv_depth.sum(), w.line_orth.sum(), w.pose.sum()])
    assert np.allclose(chk, d["in_checksum"], rtol=1e-13), "synth.make_window no longer generates the window the fixture was made from"
    return d, w


def _check_config3(d, st, rep, tol_pose, tol_cost):
    k = rep.num_iterations + 1
    assert rep.status == 0 and k == len(d["out_cost"]) and int(d["out_termination"]) == rep.termination
    assert list(rep.accepted[:k]) == list(d["out_accepted"])
    assert np.allclose(np.array(rep.cost[:k]), d["out_cost"], rtol=tol_cost) and np.allclose(np.array(rep.radius[:k]), d["out_radius"], rtol=1e-6)
    assert abs(rep.final_cost - float(d["out_final_cost"])) <= tol_cost * float(d["out_final_cost"])
    dp, da = pose_deltas(st.pose, d["out_pose"])
    assert dp < tol_pose and da < max(tol_pose, 1e-7), (dp, da)      # (the angle comes out of an arccos near 1: 4e-8 is its floor)
    assert np.abs(st.speedbias - d["out_speedbias"]).max() < tol_pose
    assert np.abs(st.inv_depth[:64] - d["out_inv_depth_head"]).max() < 10 * tol_pose and np.abs(np.asarray(st.line_orth)[:16] - d["out_line_orth_head"]).max() < 100 * tol_pose
    assert abs(st.inv_depth.sum() - d["out_landmark_checksum"][0]) <= 1e-6 * d["out_landmark_checksum"][1]


def test_oracle_reproduces_config3_trace(oracle):
    d, w = _config3()
    st, rep = oracle.solve(w)
    _check_config3(d, st, rep, 1e-9, 1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["step-wise", "fused"])
def test_hip_solver_reproduces_config3_trace(gpu_api, form):
    """BASELINE configs[3] at full size against the COMMITTED trace (no oracle at run time): both forms of the landmark-sharded path."""
    d, w = _config3()
    s = gpu_api.Solver(max_batch=1, max_points=20008, max_point_obs=240000, max_lines=5008, max_line_obs=60000)
    if form == "fused":
        s.large_comm_init(None)
        st, rep, _ = s.large_solve_fused(w)
    else:
        st, rep = s.large_solve(w)
    s.close()
    _check_config3(d, st, rep, 1e-6, 1e-8)
