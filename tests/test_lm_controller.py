"""The C++ oracle's Levenberg-Marquardt against an independent numpy controller (tests/pyref_lm.py: dense normal equations in Ceres'
Jacobi-scaled coordinates, no Schur complement, written from SURVEY.md Appendix B and sharing no solver code with the oracle).
SURVEY.md 8c(5): "two independent restatements agreeing ... <= 1e-9 on final states".  This is the strongest pin of the Ceres
semantics available without Ceres itself ("parity unpinned" stays in force)."""
import numpy as np
import pytest

import pyref_lm
from helpers import abi, synth, pose_deltas


def _compare(oracle, w, opts=None, state_tol=1e-9, cost_rtol=1e-8, cost_atol=1e-15, radius_rtol=1e-7, mcc_rtol=1e-6):
    opts = opts or abi.default_options()
    st, rep = oracle.solve(w, opts)
    x, tr = pyref_lm.solve(w, lambda win: oracle.evaluate(win, robust=True, opts=opts), opts)
    n = rep.num_iterations
    assert tr.num_iterations == n, (tr.num_iterations, n)
    assert tr.termination == rep.termination, (abi.TERM_NAMES[tr.termination], abi.TERM_NAMES[rep.termination])
    assert list(tr.accepted[:n + 1]) == list(rep.accepted[:n + 1]), (tr.accepted, list(rep.accepted[:n + 1]))
    assert np.allclose(tr.radius[:n + 1], np.array(rep.radius[:n + 1]), rtol=radius_rtol), (tr.radius, list(rep.radius[:n + 1]))
    assert np.allclose(tr.cost[:n + 1], np.array(rep.cost[:n + 1]), rtol=cost_rtol, atol=cost_atol)
    assert np.allclose(tr.model_cost_change[1:n + 1], np.array(rep.model_cost_change[1:n + 1]), rtol=mcc_rtol)
    assert abs(tr.final_cost - rep.final_cost) <= cost_rtol * rep.final_cost + cost_atol      # (a noise-free window ends at round-off level, ~1e-18)
    dp, dq = pose_deltas(x.pose, st.pose)
    scale = lambda a: max(1.0, np.abs(a).max())
    assert dp <= state_tol * scale(st.pose[:, :3]) and dq <= max(1e-7, state_tol), (dp, dq)
    assert np.abs(x.speedbias - st.speedbias).max() <= state_tol * scale(st.speedbias)
    assert np.abs(x.inv_depth - st.inv_depth).max() <= 10 * state_tol * scale(st.inv_depth)
    return rep, tr


@pytest.mark.parametrize("index", range(16))
def test_default_options_traces_agree(oracle, index):
    """16 canonical windows (W10-P150-L40-V3, noisy, perturbed start): same accept / reject pattern, radii, costs and final state.
    (line parameters are compared through the cost: an un-tagged, nearly unobservable line may differ in one angle at equal cost)"""
    rep, tr = _compare(oracle, synth.make_window(300 + index))
    assert rep.num_iterations == 10


def test_windows_with_prior_agree(oracle):
    marg = lambda win, flag: oracle.marginalize(win, flag)
    rejected = 0
    for index in (400, 401, 402, 403):
        rep, tr = _compare(oracle, synth.make_window(index, with_prior=True, marginalize_fn=marg))
        rejected += sum(1 for a in tr.accepted if a == 0)
    assert rejected >= 1              # the comparison covers rejected steps (radius halving, diagonal reuse)


@pytest.mark.parametrize("kw", [dict(n_lines=0, n_tagged=0), dict(n_points=60, n_lines=12, n_tagged=12, pt_track=4), dict(n_points=220, n_lines=60, n_tagged=0)])
def test_other_shapes_agree(oracle, kw):
    _compare(oracle, synth.make_window(77, **kw))


def test_function_tolerance_both_variants(oracle):
    """B.4: the tolerance tests precede accept / reject; the candidate is dropped (Ceres) or kept (option)."""
    w = synth.make_window(3)
    for keep in (0, 1):
        o = abi.default_options(); o.max_num_iterations = 60; o.function_tolerance = 5e-3; o.function_tol_keeps_candidate = keep
        rep, tr = _compare(oracle, w, o, state_tol=1e-8)
        assert rep.termination == abi.TERM_NAMES.index("FUNCTION_TOL")


def test_long_runs_and_convergence(oracle):
    """Noise-free window to convergence (many accepted steps, growing radius) and a noisy one with 30 iterations."""
    o = abi.default_options(); o.max_num_iterations = 30
    _compare(oracle, synth.make_window(9), o, state_tol=1e-8)
    w = synth.make_window(2, n_lines=0, n_tagged=0, noise=False, perturb=True)
    o = abi.default_options(); o.max_num_iterations = 40
    # the cost falls from 1e10 to 1e-18: its last values are round-off of the residuals, compared absolutely
    # (no prior => the 4-dof gauge is free and only held by the damping: 40 steps drift by ~1e-7 m along it at equal cost)
    rep, tr = _compare(oracle, w, o, state_tol=1e-6, cost_rtol=1e-6, cost_atol=1e-9)
    assert rep.final_cost < 1e-10


def test_without_jacobi_scaling_and_small_radius(oracle):
    # unscaled, the full 475 x 475 system has condition ~1e20: the dense solve and the oracle's Schur solve agree to ~1e-6 in the step,
    # which shows in rho and, through (2 rho - 1)^3, in the radius
    o = abi.default_options(); o.jacobi_scaling = 0
    _compare(oracle, synth.make_window(21), o, state_tol=1e-5, cost_rtol=1e-4, radius_rtol=1e-3, mcc_rtol=1e-4)
    o = abi.default_options(); o.initial_trust_region_radius = 1e-2        # heavy damping from the start
    _compare(oracle, synth.make_window(22), o)
