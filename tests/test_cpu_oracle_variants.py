"""The claim behind tests/test_gpu_stress.py, checked instead of narrated: on the heavily perturbed STRESS windows two CPU builds of the ONE sequential
oracle program (`-O2 -ffp-contract=off`, the checker of this suite, and `-O3 -march=native`: fused multiply-adds, other summation orders) part ways
in the same manner, and about as often, as the HIP solver does from the oracle -- so a differing LM trace there is a property of the window
(exponential amplification of round-off far from the optimum), not of a solver.  CPU only; the script form with the per-window print-out is
tests/cpu_soak_oracle_variants.py.  The assertions (a) - (c), (e) are the ones test_gpu_stress.py makes for the HIP solver."""
import os
import subprocess

import numpy as np

from helpers import abi, synth, pose_deltas, first_divergence
from oracle_binding import Oracle, ROOT


def test_two_cpu_builds_of_the_oracle_diverge_on_stress_windows_like_the_hip_solver_does(tmp_path):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native", "OUT=" + str(tmp_path)])
    a, b = Oracle(), Oracle(os.path.join(str(tmp_path), "liboracle_native.so"))
    opts = abi.default_options(); opts.max_num_iterations = 30
    rng = np.random.default_rng(77)
    n_solved = n_diff = 0; worst_dp = worst_dq = 0.0
    for i in range(60):
        # the random stream of gpu_soak_rejections.stressed_window (every window draws; the prior-free half needs no GPU to generate)
        w = synth.make_window(9000 + i).copy()
        amp = float(rng.choice([0.05, 0.2, 0.5]))
        w.pose[2:, :3] += amp * rng.standard_normal((9, 3)); w.inv_depth *= np.exp(amp * rng.standard_normal(len(w.inv_depth)))
        w.line_orth += 0.3 * amp * rng.standard_normal(w.line_orth.shape)
        if i % 2: continue
        n_solved += 1
        sa, ra = a.solve(w, opts); sb, rb = b.solve(w, opts)
        d = first_divergence(rb, ra, opts)
        k = d["k"] if d is not None else min(rb.num_iterations, ra.num_iterations)
        drift = [abs(rb.cost[q] - ra.cost[q]) / abs(ra.cost[q]) for q in range(k + 1)]
        assert max(drift[:3]) <= 1e-8, (i, drift[:3])                                                   # (a) nothing amplified yet
        for q in range(1, len(drift)):                                                                   # (c) smooth growth
            assert drift[q] <= 1e4 * max(drift[q - 1], 1e-12), (i, q, drift)
        assert rb.final_cost < 1e-6 * rb.initial_cost and ra.final_cost < 1e-6 * ra.initial_cost, i      # (e) both converge
        if d is not None:
            n_diff += 1
            if d["kind"] == "accept":
                assert drift[k - 1] >= 1e-6 or d["rel_margin"] < 1e-6, (i, d, drift)                     # (b) a decision flips only after the paths separated
            continue
        dp, dq = pose_deltas(sb.pose, sa.pose)
        worst_dp, worst_dq = max(worst_dp, dp), max(worst_dq, dq)
    assert worst_dp <= 1e-4 and worst_dq <= 1e-4, (worst_dp, worst_dq)                                   # (d) identical traces => identical end states
    print("%d prior-free stress windows, two CPU builds of the oracle: LM traces differ in %d (HIP vs oracle on all 60: 7)" % (n_solved, n_diff))
    # 4 of 30 in the build container (what -march=native fuses depends on the host, so a band, not a number): the same rate as HIP-vs-oracle (7 of 60)
    assert 1 <= n_diff <= 8, n_diff
