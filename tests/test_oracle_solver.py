"""Known-answer and invariant tests of the oracle's Levenberg-Marquardt restatement."""
import numpy as np
import pytest

from helpers import uvs, abi, synth, lm_reduced_system, pose_deltas


def _rel_pose(pose, a=0, b=10):
    Ra = synth.quat_to_R(pose[a, 3:]); Rb = synth.quat_to_R(pose[b, 3:])
    return Ra.T @ (pose[b, :3] - pose[a, :3]), Ra.T @ Rb


def test_known_answer_points_imu(oracle):
    """Noise-free points + IMU: from a perturbed start LM must return to the ground truth (gauge-invariant quantities)."""
    w = synth.make_window(2, n_lines=0, n_tagged=0, noise=False, perturb=True)
    o = abi.default_options(); o.max_num_iterations = 40
    st, rep = oracle.solve(w, o)
    assert rep.final_cost < 1e-12 * rep.initial_cost and rep.final_cost < 1e-10
    t, R = _rel_pose(st.pose); t0, R0 = _rel_pose(w.truth["pose"])
    assert np.abs(t - t0).max() < 1e-7 and np.abs(R - R0).max() < 1e-8
    assert np.abs(st.speedbias[:, 3:] - w.truth["speedbias"][:, 3:]).max() < 1e-6   # biases are observable


def test_schur_equals_dense_solve(oracle):
    w = synth.make_window(1)
    s0, r0 = oracle.solve(w, linear_mode=0)
    s1, r1 = oracle.solve(w, linear_mode=1)
    assert r0.num_iterations == r1.num_iterations
    assert abs(r0.final_cost - r1.final_cost) <= 1e-9 * r0.final_cost
    assert pose_deltas(s0.pose, s1.pose)[0] < 1e-9
    assert np.abs(s0.line_orth - s1.line_orth).max() < 1e-7


def test_trace_invariants(oracle):
    w = synth.make_window(3)
    st, rep = oracle.solve(w)
    t = rep.trace()
    assert rep.num_iterations == 10 and rep.termination == 0      # NO_CONVERGENCE: the iteration cap stops it (B.6)
    acc = t["accepted"]
    for k in range(1, rep.num_iterations + 1):
        if acc[k] == 1:
            assert t["cost"][k] < t["cost"][k - 1]                 # monotone over accepted steps
            assert t["relative_decrease"][k] > 1e-3
            assert t["radius"][k] >= t["radius"][k - 1] / 1.0000001 or t["relative_decrease"][k] < 0.75
        else:
            assert t["cost"][k] == t["cost"][k - 1]
            assert t["radius"][k] < t["radius"][k - 1]
        assert t["model_cost_change"][k] > 0
    assert rep.final_cost == t["cost"][rep.num_iterations]


def test_first_step_matches_numpy_normal_equations(oracle):
    """One LM iteration of the oracle == numpy solve of the damped normal equations built from the evaluation dump."""
    w = synth.make_window(4)
    ev = oracle.evaluate(w, robust=True)
    ref = lm_reduced_system(w, ev)
    o = abi.default_options(); o.max_num_iterations = 1
    st, rep = oracle.solve(w, o)
    step = ref["step"]
    F = 165
    for f in range(abi.NUM_FRAMES):
        assert np.allclose(st.pose[f, :3] - w.pose[f, :3], step[15 * f:15 * f + 3], rtol=1e-4, atol=1e-7)
        assert np.allclose(st.speedbias[f] - w.speedbias[f], step[15 * f + 6:15 * f + 15], rtol=1e-4, atol=1e-7)
    assert np.allclose(st.inv_depth - w.inv_depth, step[F:F + len(w.inv_depth)], rtol=1e-4, atol=1e-7)
    assert np.allclose((st.line_orth - w.line_orth).ravel(), step[F + len(w.inv_depth):], rtol=1e-4, atol=1e-7)
    g, H, dd = ref["gfull"], ref["H"], ref["dd"]
    mcc = -(step @ g) - 0.5 * step @ H @ step
    assert abs(rep.model_cost_change[1] - mcc) <= 1e-8 * abs(mcc)
    assert abs(0.5 * (step @ (dd * step) - step @ g) - mcc) <= 1e-8 * abs(mcc)     # closed form used on the device


def test_max_iterations_and_zero_iterations(oracle):
    w = synth.make_window(5)
    o = abi.default_options(); o.max_num_iterations = 0
    st, rep = oracle.solve(w, o)
    assert rep.num_iterations == 0 and np.array_equal(st.pose, w.pose) and rep.final_cost == rep.initial_cost


def test_function_tolerance_terminates_without_taking_the_step(oracle):
    w = synth.make_window(3)
    o = abi.default_options(); o.max_num_iterations = 60; o.function_tolerance = 5e-3
    st, rep = oracle.solve(w, o)
    assert rep.termination == 3           # FUNCTION_TOL
    k = rep.num_iterations
    assert rep.accepted[k] == 0 and rep.final_cost == rep.cost[k - 1]      # Ceres order: tolerance check precedes accept (B.4)
    o.function_tol_keeps_candidate = 1
    st2, rep2 = oracle.solve(w, o)
    assert rep2.termination == 3 and rep2.final_cost <= rep.final_cost


def test_marginalization_prior_is_consistent(oracle):
    """J0^T J0 and J0^T r0 reproduce the Schur complement of the marginalised normal equations (marginalization_factor.cpp:295-296)."""
    w = synth.make_window(6)
    p = oracle.marginalize(w, flag=0)
    # frame 0 is linked to frames 1..6 by 6-frame point tracks / 7-frame line tracks: 6 poses + speed-bias 1 + extrinsic
    assert p.n == 6 * 6 + 9 + 6 and p.n_blocks == 8
    kinds = list(p.block_kind[:p.n_blocks]); frames = list(p.block_frame[:p.n_blocks])
    assert kinds == [abi.BLOCK_POSE] * 6 + [abi.BLOCK_SPEEDBIAS, abi.BLOCK_EX_POSE] and frames[:7] == list(range(6)) + [0]
    J0, r0 = p.J0(), p.r0()
    A = J0.T @ J0
    assert np.allclose(A, A.T)
    ev_ = np.linalg.eigvalsh(A)
    assert ev_.min() > -1e-6 * ev_.max()
    # linearization point = the post-solve state of frames 1..10 shifted to 0..9
    x0 = np.array(p.x0[:7]); assert np.allclose(x0, w.pose[1])
    # using the prior in the next window: residual at x0 is r0, cost 0.5|r0|^2
    w2 = synth.make_window(6)
    w2.pose[:10] = w.pose[1:11]; w2.speedbias[0] = w.speedbias[1]; w2.prior = p
    e = oracle.evaluate(w2, robust=True)
    assert np.allclose(e.prior_r[:p.n], r0, atol=1e-9)


def test_marginalize_second_new(oracle):
    marg = lambda win, flag: oracle.marginalize(win, flag)
    w2 = synth.make_window(7, with_prior=True, marginalize_fn=marg)
    p = w2.prior
    q = oracle.marginalize(w2, flag=1)           # prior touches Pose[9] -> it is dropped, nothing else changes frame
    assert q.n == p.n - 6 and q.n_blocks == p.n_blocks - 1
    assert 9 not in [q.block_frame[b] for b in range(q.n_blocks) if q.block_kind[b] == abi.BLOCK_POSE]
    w3 = synth.make_window(7); w3.prior = q
    r = oracle.marginalize(w3, flag=1)           # no Pose[9] any more -> prior returned unchanged
    assert r.n == q.n and np.array_equal(r.J0(), q.J0())


def test_prior_window_solves(oracle):
    marg = lambda win, flag: oracle.marginalize(win, flag)
    w = synth.make_window(8, with_prior=True, marginalize_fn=marg)
    assert w.prior is not None and w.prior.n == 75
    assert synth.algorithmic_bytes(w) == 158880                      # SURVEY.md section 8d canonical figure
    st, rep = oracle.solve(w)
    assert rep.final_cost < rep.initial_cost * 1e-3 and rep.num_successful >= 3
    s2, r2 = oracle.solve(w, linear_mode=1)
    assert abs(rep.final_cost - r2.final_cost) <= 1e-8 * rep.final_cost


def test_sym_eig(oracle):
    rng = np.random.default_rng(3)
    M = rng.normal(size=(40, 40)); A = M @ M.T
    ev_, V = oracle.sym_eig(A)
    assert np.allclose(V @ np.diag(ev_) @ V.T, A, atol=1e-10) and np.allclose(V.T @ V, np.eye(40), atol=1e-12)
