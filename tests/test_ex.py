"""ESTIMATE_EXTRINSIC path: para_Ex_Pose is a free 6-dof manifold block (estimator.cpp:784-788; d r / d ex_pose of ProjectionFactor,
projection_factor.cpp:143-147).  The HIP path keeps the 176-wide padded reduced system and puts the six dofs into the spare slots of
frames 0..5; the oracle orders them after the frames.  Both are exact solves of the same system."""
import numpy as np
import pytest

from helpers import uvs, abi, synth, pose_deltas

pytestmark = pytest.mark.gpu


def _opts(td=False):
    o = abi.default_options(); o.estimate_extrinsic = 1; o.estimate_td = 1 if td else 0
    return o


def _perturbed_extrinsic(w, seed):
    rng = np.random.default_rng(seed)
    o = w.copy()
    o.ex_pose = o.ex_pose.copy()
    o.ex_pose[:3] += 0.01 * rng.standard_normal(3)
    q = o.ex_pose[3:] + 0.005 * rng.standard_normal(4); o.ex_pose[3:] = q / np.linalg.norm(q)
    return o


@pytest.mark.parametrize("index,td", [(95, False), (96, False), (97, True)])
def test_extrinsic_solve_matches_oracle(gpu_api, oracle, index, td):
    w = _perturbed_extrinsic(synth.make_window(index), index)
    if td:
        w = synth.add_time_offset(w, td_true=0.004)
    o = _opts(td)
    s = gpu_api.Solver(opts=o, max_batch=2)
    sg, rg = s.solve(w)
    s.close()
    so, ro = oracle.solve(w, opts=o)
    assert rg.status == 0 and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    assert abs(rg.initial_cost - ro.initial_cost) <= 1e-10 * ro.initial_cost
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-4 and da < 1e-4, (dp, da)
    assert np.abs(sg.ex_pose - so.ex_pose).max() < 1e-5 and not np.array_equal(sg.ex_pose, w.ex_pose)
    assert abs(rg.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost
    if td:
        assert abs(sg.td - so.td) < 1e-6


def test_extrinsic_first_iteration_step(gpu_api, oracle):
    """First LM step of the two formulations (different variable orders, Jacobi-scaled vs unscaled coordinates)."""
    w = _perturbed_extrinsic(synth.make_window(98), 3)
    o = _opts(); o.max_num_iterations = 1
    s = gpu_api.Solver(opts=o, max_batch=2)
    sg, rg = s.solve(w)
    s.close()
    so, ro = oracle.solve(w, opts=o)
    assert abs(rg.model_cost_change[1] - ro.model_cost_change[1]) <= 1e-8 * abs(ro.model_cost_change[1])
    assert abs(rg.candidate_cost[1] - ro.candidate_cost[1]) <= 1e-8 * abs(ro.candidate_cost[1])
    assert np.abs(sg.ex_pose - so.ex_pose).max() < 1e-9


def test_extrinsic_block_in_the_prior_and_marginalization(gpu_api, oracle):
    o = _opts()
    w = _perturbed_extrinsic(synth.make_window(99), 4)
    s = gpu_api.Solver(opts=o, max_batch=2)
    sg, rg = s.solve(w)
    wg = w.with_state(sg)
    pg = s.marginalize(wg, 0); po = oracle.marginalize(wg, 0, opts=o)
    assert pg.n == po.n and abi.UVS_BLOCK_EX_POSE in [pg.block_kind[b] for b in range(pg.n_blocks)]
    Hg, Ho = pg.J0().T @ pg.J0(), po.J0().T @ po.J0()
    assert np.abs(Hg - Ho).max() <= 1e-6 * np.abs(Ho).max()
    w2 = _perturbed_extrinsic(synth.make_window(100), 5); w2.prior = pg
    s2, r2 = s.solve(w2)
    so, ro = oracle.solve(w2, opts=o)
    s.close()
    assert r2.status == 0 and r2.num_iterations == ro.num_iterations
    assert list(r2.accepted[: r2.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    assert pose_deltas(s2.pose, so.pose)[0] < 1e-4 and np.abs(s2.ex_pose - so.ex_pose).max() < 1e-5
    assert abs(r2.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost


def test_extrinsic_large_window_path(gpu_api, oracle):
    o = _opts()
    w = _perturbed_extrinsic(synth.make_window(101, n_points=600, n_lines=160, n_tagged=120), 6)
    s = gpu_api.Solver(opts=o, max_batch=2)
    sg, rg = s.large_solve(w)
    s.close()
    so, ro = oracle.solve(w, opts=o)
    assert rg.status == 0 and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    assert pose_deltas(sg.pose, so.pose)[0] < 1e-4 and np.abs(sg.ex_pose - so.ex_pose).max() < 1e-5


@pytest.mark.parametrize("which", [1, 14])
def test_td_and_extrinsic_full_chunks(gpu_api, oracle, which):
    """Regression (found by tests/gpu_soak_options.py): with BOTH options on every observation has ten direct gather entries (3 + 3 td
    + 3 ex + the (ex, td) one), the host's chunk-capacity estimate counted nine, and on windows whose chunks were nearly full the lists
    ran past the LDS staging area into the LM state (initial cost inf, UVS_ERR_NUMERIC).  The layout is now also checked after the
    lists are built."""
    rng = np.random.default_rng(23000)
    for i in range(which + 1):
        npt, nln = int(rng.integers(20, 300)), int(rng.integers(0, 80))
        ntag, ptt, lnt = int(rng.integers(0, nln + 1)), int(rng.integers(3, 10)), int(rng.integers(5, 10))
        ex_noise = 0.01 * rng.standard_normal(3); q_noise = 0.005 * rng.standard_normal(4); td_true = float(rng.uniform(-0.01, 0.01))
    w = synth.make_window(23000 + which, n_points=npt, n_lines=nln, n_tagged=ntag, pt_track=ptt, ln_track=lnt).copy()
    w.ex_pose = w.ex_pose.copy(); w.ex_pose[:3] += ex_noise
    q = w.ex_pose[3:] + q_noise; w.ex_pose[3:] = q / np.linalg.norm(q)
    w = synth.add_time_offset(w, td_true=td_true, seed=which)
    o = _opts(td=True)
    s = gpu_api.Solver(opts=o, max_batch=2, max_points=320, max_point_obs=3600, max_lines=100, max_line_obs=1100)
    sg, rg = s.solve(w)
    s.close()
    so, ro = oracle.solve(w, opts=o)
    assert rg.status == 0 and np.isfinite(rg.initial_cost) and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-6 and da < 1e-6 and abs(sg.td - so.td) < 1e-8
