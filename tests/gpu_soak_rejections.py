"""Soak for the re-damping path and for the stress margins of LM-trace parity (run by hand on the GPU box): heavily perturbed windows, 30 LM
iterations, so that runs of consecutive rejected steps occur (every rejection re-damps the stored linearization relative to the previous one).
Both single-window forms against the oracle.  For every window whose trace differs from the oracle's the script reports WHERE the two part ways
(first differing decision), HOW CLOSE both sides were to the threshold of the test that decided there (helpers.first_divergence) and where the two
solves END anyway (final poses / cost): a divergence is benign when it happens on the knife edge of a Ceres threshold and both end states are
within the stated tolerance of each other (1e-4 m / 1e-4 rad, BASELINE.json north_star).
python tests/gpu_soak_rejections.py [n windows] [first seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, abi, synth, pose_deltas, first_divergence
from oracle_binding import Oracle


def stressed_window(i, s, rng):
    w = synth.make_window(9000 + i, with_prior=bool(i % 2), marginalize_fn=(lambda win, flag: s.marginalize(win, flag)) if i % 2 else None).copy()
    amp = float(rng.choice([0.05, 0.2, 0.5]))
    w.pose[2:, :3] += amp * rng.standard_normal((9, 3)); w.inv_depth *= np.exp(amp * rng.standard_normal(len(w.inv_depth)))
    w.line_orth += 0.3 * amp * rng.standard_normal(w.line_orth.shape)
    return w, amp


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    o = Oracle()
    opts = abi.default_options(); opts.max_num_iterations = 30
    s = uvs.api.Solver(opts=opts, max_batch=1)
    rng = np.random.default_rng(77)
    stats = {"persistent": [0, 0.0, 0.0], "fused": [0, 0.0, 0.0]}; rej_total = 0; longest = 0; t0 = time.time()
    div = []; growth = []
    for i in range(N):
        w, amp = stressed_window(i, s, rng)
        so, ro = o.solve(w, opts)
        acc = list(ro.accepted[1:ro.num_iterations + 1]); rej_total += sum(1 for a in acc if a != 1)
        run = 0
        for a in acc:
            run = run + 1 if a != 1 else 0; longest = max(longest, run)
        for form in ("persistent", "fused"):
            if form == "fused": sg, rg, _ = s.large_solve_fused(w)
            else: sg, rg = s.solve(w)
            d = first_divergence(rg, ro, opts)
            dp, dq = pose_deltas(sg.pose, so.pose)
            dc = abs(rg.final_cost - ro.final_cost) / max(ro.final_cost, 1e-300)
            if d is not None:
                stats[form][0] += 1
                # how the two solvers drift apart BEFORE their first differing decision: relative difference of the accepted cost per iteration
                growth.append((form, i, d["k"], [abs(rg.cost[q] - ro.cost[q]) / abs(ro.cost[q]) for q in range(0, d["k"] + 1)], [int(ro.accepted[q]) for q in range(0, d["k"] + 1)]))
                div.append((form, i, amp, d, dp, dq, dc, rg.final_cost, ro.final_cost, float(np.abs(sg.inv_depth - so.inv_depth).max()), float(np.abs(sg.line_orth - so.line_orth).max())))
            else: stats[form][1] = max(stats[form][1], dp); stats[form][2] = max(stats[form][2], dc)
    print("%d windows, %d rejected / invalid steps in the oracle's traces (longest run %d), %.1f s" % (N, rej_total, longest, time.time() - t0))
    for form, (bad, dp, dc) in stats.items():
        print("  %-10s trace differences %d; worst over identical traces: dp %.2e m, relative final cost %.2e" % (form, bad, dp, dc))
    print("windows whose LM trace differs from the oracle's: first differing decision, distance of both sides from the threshold there, and the end states")
    print("  %-10s %4s %4s %3s %-9s %-12s %-12s %-10s %-10s | %-9s %-9s %-9s %-9s %-9s" % ("form", "win", "amp", "k", "test", "gpu", "oracle", "margin_gpu", "margin_orc", "end dp[m]", "dq[rad]", "rel cost", "invd", "line"))
    for form, i, amp, d, dp, dq, dc, cg, co, di, dl in div:
        g_ = ("rho %.6g" % d["rho_gpu"]) if d["kind"] == "accept" else str(d["gpu"]); o_ = ("rho %.6g" % d["rho_oracle"]) if d["kind"] == "accept" else str(d["oracle"])
        print("  %-10s %4d %4.2f %3d %-9s %-12s %-12s %-10.2e %-10.2e | %-9.2e %-9.2e %-9.2e %-9.2e %-9.2e" % (form, i, amp, d["k"], d["kind"], g_, o_, d["margin_gpu"], d["margin_oracle"], dp, dq, dc, di, dl))
    print("conditioning of the deciding quantity at those iterations: |model cost change| / cost on both sides, the round-off noise of rho that follows (4e-16 cost / |mcc|),")
    print("relative difference of the two solvers' cost and candidate cost THERE (identical decisions up to that point), iterations / termination of both")
    for form, i, amp, d, dp, dq, dc, cg, co, di, dl in div:
        if d["kind"] == "accept":
            print("  %-10s %4d k %2d  mcc/cost %.1e | %.1e   rho noise %.1e | %.1e   cost diff %.1e  candidate diff %.1e   final cost %.9g | %.9g" % (form, i, d["k"], d["mcc_over_cost"][0], d["mcc_over_cost"][1], d["rho_noise"][0], d["rho_noise"][1], d["cost_rel_diff"], d["cand_rel_diff"], cg, co))
        else: print("  %-10s %4d k %2d  %s" % (form, i, d["k"], {k_: v for k_, v in d.items() if k_ not in ("k", "kind")}))
    print("drift of the accepted cost (relative difference GPU vs oracle) iteration by iteration up to the first differing decision [oracle's accept flags]:")
    for form, i, k, g, a in growth:
        print("  %-10s %4d  " % (form, i) + " ".join("%.0e%s" % (v, "" if f == 1 else "r") for v, f in zip(g, a)))
    if div:
        print("  worst relative threshold margin at a divergence %.2e; worst end-state difference over diverging windows: dp %.2e m, dq %.2e rad, relative final cost %.2e"
              % (max(d[3]["rel_margin"] for d in div), max(d[4] for d in div), max(d[5] for d in div), max(d[6] for d in div)))
