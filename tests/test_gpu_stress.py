"""Stress margins of LM-trace parity, as an assertion (round-3 verdict item 3; the measurements are printed by tests/gpu_soak_rejections.py and
tests/cpu_soak_oracle_variants.py).

What the soaks established: on heavily perturbed windows (0.5 m / 50 % start errors, 30 iterations, runs of rejected steps) the LM iteration amplifies
round-off exponentially -- two CPU builds of the SAME oracle sources (`-O2 -ffp-contract=off` vs `-O3 -march=native`) part ways on the same windows
and in the same manner as the HIP solver does from the oracle: the relative difference of the accepted cost grows from 1e-16 through 1e-11 ... 1e-4
and only THEN a Ceres decision (rho > min_relative_decrease, estimator.cpp:982-994 -> Appendix B.5) comes out differently, with rho far from its
threshold on both sides.  So "identical traces" cannot be asserted there; what CAN be asserted, and what a real defect (wrong accept rule, wrong
damping after a rejection, a discontinuity) would violate, is:
  (a) both solvers agree to round-off over the first iterations (nothing has been amplified yet);
  (b) a decision differs only after the trajectories have separated (cost drift >= 1e-6 at the iteration before) or on the knife edge of the threshold;
  (c) the drift grows smoothly (no jump by more than 1e4 between consecutive iterations once it is above the noise floor);
  (d) where the traces ARE identical the end states agree to the stated tolerance (1e-4 m / 1e-4 rad, BASELINE.json north_star);
  (e) both solvers converge (final cost < 1e-6 of the initial cost) whatever path they took."""
import numpy as np
import pytest

from helpers import abi, pose_deltas, first_divergence
from gpu_soak_rejections import stressed_window


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["persistent", "fused"])
def test_stress_windows_diverge_from_the_oracle_only_by_amplified_round_off(gpu_api, oracle, form):
    opts = abi.default_options(); opts.max_num_iterations = 30
    s = gpu_api.Solver(opts=opts, max_batch=1)
    rng = np.random.default_rng(77)
    n_diff = 0; worst_dp = worst_dq = 0.0
    for i in range(60):
        w, amp = stressed_window(i, s, rng)
        so, ro = oracle.solve(w, opts)
        if form == "fused": sg, rg, _ = s.large_solve_fused(w)
        else: sg, rg = s.solve(w)
        d = first_divergence(rg, ro, opts)
        k = d["k"] if d is not None else min(rg.num_iterations, ro.num_iterations)
        drift = [abs(rg.cost[q] - ro.cost[q]) / abs(ro.cost[q]) for q in range(k + 1)]
        assert max(drift[:3]) <= 1e-8, (i, drift[:3])                                                   # (a)
        for q in range(1, len(drift)):                                                                   # (c)
            assert drift[q] <= 1e4 * max(drift[q - 1], 1e-12), (i, q, drift)
        assert rg.final_cost < 1e-6 * rg.initial_cost and ro.final_cost < 1e-6 * ro.initial_cost, i      # (e)
        if d is not None:
            n_diff += 1
            if d["kind"] == "accept":
                assert drift[k - 1] >= 1e-6 or d["rel_margin"] < 1e-6, (i, d, drift)                     # (b)
            continue
        dp, dq = pose_deltas(sg.pose, so.pose)
        worst_dp, worst_dq = max(worst_dp, dp), max(worst_dq, dq)
    assert worst_dp <= 1e-4 and worst_dq <= 1e-4, (worst_dp, worst_dq)                                   # (d)
    print("stress windows (%s): LM traces differ from the oracle's in %d of 60" % (form, n_diff))
    # 7 of 60 in rounds 3 - 5 -- the rate at which two CPU builds of the oracle itself part ways on these windows (4 of the 30 prior-free ones:
    # tests/test_cpu_oracle_variants.py asserts that); a jump would mean the amplification started earlier than round-off explains
    assert n_diff <= 9, n_diff
    s.close()
