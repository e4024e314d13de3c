"""GPU parity: the HIP solver (through the C ABI) against the CPU oracle on identical synthetic windows.

Tolerances (north_star): pose delta <= 1e-4 m / 1e-4 rad, final cost <= 1e-6 relative.  The residual /
Jacobian comparison is element-wise at 1e-9 relative (hand-derived HIP Jacobians vs the oracle's Jets).
"""
import numpy as np
import pytest

from helpers import uvs, abi, synth, lm_reduced_system, unpad, pose_deltas

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(gpu_api):
    s = gpu_api.Solver(max_batch=512)
    yield s
    s.close()


def _relerr(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


@pytest.mark.parametrize("robust", [True, False])
def test_evaluate_elementwise(solver, oracle, robust):
    w = synth.make_window(3)
    eg = solver.evaluate(w, robust=robust)
    eo = oracle.evaluate(w, robust=robust)
    for name in ("pt_r", "pt_J", "ln_r", "ln_J", "vp_r", "vp_J", "imu_r", "imu_J"):
        a, b = getattr(eg, name), getattr(eo, name)
        assert _relerr(a, b) < 1e-9, name
    assert abs(eg.cost - eo.cost) <= 1e-10 * abs(eo.cost)


def test_first_iteration_system(solver, oracle):
    w = synth.make_window(4)
    eo = oracle.evaluate(w, robust=True)
    ref = lm_reduced_system(w, eo)
    d = solver.debug_first_iteration(w)
    S = unpad(d["S"]); S = S + np.tril(S, -1).T
    assert abs(d["cost"] - eo.cost) <= 1e-10 * eo.cost
    assert _relerr(unpad(d["hd"]), ref["hd"][:165]) < 1e-9
    assert _relerr(unpad(d["g"]), ref["g"]) < 1e-8
    assert np.abs(S - ref["S"]).max() <= 1e-9 * np.abs(ref["S"]).max()
    # block by block as well: the entries of S span fourteen orders of magnitude, and a whole-matrix max norm is blind to a wrong 6x6 pose
    # block of a weakly connected frame pair (an experimental 512-thread build once passed the line above with three such blocks off by O(1))
    Rf = ref["S"]
    for a in range(11):
        for b in range(11):
            ra = Rf[15 * a:15 * a + 15, 15 * b:15 * b + 15]
            for sl in ((slice(0, 6), slice(0, 6)), (slice(0, 15), slice(6, 15)), (slice(6, 15), slice(0, 6))):      # pose-pose, and the speed / bias parts
                den = np.abs(ra[sl]).max()
                if den > 0.0:
                    assert np.abs(S[15 * a:15 * a + 15, 15 * b:15 * b + 15][sl] - ra[sl]).max() <= 1e-7 * den, (a, b)
    assert _relerr(unpad(d["dd"]), ref["dd"][:165]) < 1e-9
    assert d["chol_ok"] == 1.0
    assert _relerr(unpad(d["step"]), ref["step"][:165]) < 1e-6


@pytest.mark.parametrize("index", [0, 1, 5])
def test_solve_matches_oracle(solver, oracle, index):
    w = synth.make_window(index)
    sg, rg = solver.solve(w)
    so, ro = oracle.solve(w)
    assert rg.status == 0
    assert rg.num_iterations == ro.num_iterations
    assert rg.termination == ro.termination
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-4 and da < 1e-4, (dp, da)
    assert np.abs(sg.speedbias - so.speedbias).max() < 1e-4
    assert abs(rg.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost
    assert np.abs(sg.inv_depth - so.inv_depth).max() < 1e-4
    assert np.abs(sg.line_orth - so.line_orth).max() < 1e-4


def test_points_only_known_answer(solver):
    """Noise-free points+IMU window: LM from a perturbed start must return to ground truth (cost -> 0)."""
    w = synth.make_window(2, n_lines=0, n_tagged=0, noise=False, perturb=True)
    opts = abi.default_options(); opts.max_num_iterations = 30
    s = uvs.api.Solver(opts=opts, max_batch=4)
    st, rep = s.solve(w)
    s.close()
    assert rep.final_cost < 1e-12
    T = w.truth
    R0 = synth.quat_to_R(st.pose[0, 3:]); R0t = synth.quat_to_R(T["pose"][0, 3:])
    rel = R0.T @ (st.pose[10, :3] - st.pose[0, :3]); relt = R0t.T @ (T["pose"][10, :3] - T["pose"][0, :3])
    assert np.abs(rel - relt).max() < 1e-6


def test_batch_matches_single(solver, oracle):
    ws = [synth.make_window(i) for i in range(6)]
    solver.upload(ws)
    ms = solver.solve_resident()
    states, reps = solver.download()
    assert ms > 0
    for w, st, rep in zip(ws, states, reps):
        so, ro = oracle.solve(w)
        dp, da = pose_deltas(st.pose, so.pose)
        assert dp < 1e-4 and da < 1e-4
        assert abs(rep.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost
    # deterministic: a second run of the resident batch reproduces the first bit for bit
    solver.solve_resident()
    states2, reps2 = solver.download()
    for a, b in zip(states, states2):
        assert np.array_equal(a.pose, b.pose) and np.array_equal(a.inv_depth, b.inv_depth) and np.array_equal(a.line_orth, b.line_orth)


def test_large_window_path_matches_oracle(solver, oracle):
    """configs[3] code path (grid of landmark chunks + reduce + single-workgroup solve + grid back-substitution) on a window the
    oracle still solves in seconds."""
    w = synth.make_window(40, n_points=900, n_lines=240, n_tagged=180)
    sg, rg = solver.large_solve(w)
    so, ro = oracle.solve(w)
    assert rg.status == 0 and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-4 and da < 1e-4, (dp, da)
    assert abs(rg.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost
    assert np.abs(sg.inv_depth - so.inv_depth).max() < 1e-4 and np.abs(sg.line_orth - so.line_orth).max() < 1e-4


def test_large_path_equals_persistent_kernel(solver):
    w = synth.make_window(41)
    s1, r1 = solver.solve(w)
    s2, r2 = solver.large_solve(w)
    assert r1.num_iterations == r2.num_iterations and list(r1.accepted[:11]) == list(r2.accepted[:11])
    assert abs(r1.final_cost - r2.final_cost) <= 1e-9 * r1.final_cost
    assert pose_deltas(s1.pose, s2.pose)[0] < 1e-8


def test_config3_full_size_matches_oracle(gpu_api, oracle):
    """BASELINE configs[3] AT FULL SIZE (20 000 point + 5 000 line landmarks, 135 000 observations): the landmark-sharded grid path
    against the CPU oracle (which needs ~7 s for it), plus the size-independent properties: bitwise reproducibility of the
    deterministic two-level reductions, and landmark shards summing to the unsharded reduced system (one rank holding everything
    equals the single-call path exactly)."""
    w = synth.make_window(70, n_points=20000, n_lines=5000, n_tagged=3750)
    assert len(w.pt_lm) == 100000 and len(w.ln_lm) == 35000
    s = gpu_api.Solver(max_batch=1, max_points=20008, max_point_obs=240000, max_lines=5008, max_line_obs=60000)
    sg, rg = s.large_solve(w)
    so, ro = oracle.solve(w)
    assert rg.status == 0 and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-6 and da < 1e-6, (dp, da)                          # north star: 1e-4 m / 1e-4 rad
    assert abs(rg.final_cost - ro.final_cost) <= 1e-8 * ro.final_cost and abs(rg.initial_cost - ro.initial_cost) <= 1e-10 * ro.initial_cost
    assert np.abs(sg.speedbias - so.speedbias).max() < 1e-6
    assert np.abs(sg.inv_depth - so.inv_depth).max() < 1e-6 and np.abs(sg.line_orth - so.line_orth).max() < 1e-5
    s2, r2 = s.large_solve(w)                                         # same handle, same window: bit for bit
    assert np.array_equal(sg.pose, s2.pose) and np.array_equal(sg.inv_depth, s2.inv_depth) and np.array_equal(sg.line_orth, s2.line_orth)
    assert r2.final_cost == rg.final_cost and list(r2.cost[:11]) == list(rg.cost[:11])
    # the FUSED loop -- the path bench.py times for this configuration (510 balanced chunks on persistent workgroups, the frame-terms
    # workgroup, trust-region control and re-damping on the device) -- against the same oracle solve, then bit for bit against itself
    s.large_comm_init(None)
    sf, rf, loop_ms = s.large_solve_fused(w)
    n = ro.num_iterations
    assert rf.status == 0 and rf.num_iterations == n and rf.termination == ro.termination
    assert list(rf.accepted[: n + 1]) == list(ro.accepted[: n + 1])
    assert np.allclose(np.array(rf.cost[: n + 1]), np.array(ro.cost[: n + 1]), rtol=1e-8) and np.allclose(np.array(rf.radius[: n + 1]), np.array(ro.radius[: n + 1]), rtol=1e-6)
    dpf, daf = pose_deltas(sf.pose, so.pose)
    assert dpf < 1e-6 and daf < 1e-6, (dpf, daf)
    assert abs(rf.final_cost - ro.final_cost) <= 1e-8 * ro.final_cost and abs(rf.initial_cost - ro.initial_cost) <= 1e-10 * ro.initial_cost
    assert np.abs(sf.speedbias - so.speedbias).max() < 1e-6
    assert np.abs(sf.inv_depth - so.inv_depth).max() < 1e-6 and np.abs(sf.line_orth - so.line_orth).max() < 1e-5
    sf2, rf2, _ = s.large_solve_fused(w)
    assert np.array_equal(sf.pose, sf2.pose) and np.array_equal(sf.inv_depth, sf2.inv_depth) and np.array_equal(sf.line_orth, sf2.line_orth)
    assert rf2.final_cost == rf.final_cost and list(rf2.cost[:11]) == list(rf.cost[:11])
    s.close()
    print("configs[3] full size: max |dp| %.2e m, |dtheta| %.2e rad vs oracle; final cost %.10g | %.10g; fused loop %.2e m, %.10g, %.3f ms" % (dp, da, rg.final_cost, ro.final_cost, dpf, rf.final_cost, loop_ms))
