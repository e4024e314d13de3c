"""Soak for the re-damping path (run by hand on the GPU box): heavily perturbed windows, 30 LM iterations, so that runs of consecutive rejected
steps occur (every rejection re-damps the stored linearization relative to the previous one).  Both single-window forms against the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, abi, synth, pose_deltas
from oracle_binding import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
o = Oracle()
opts = abi.default_options(); opts.max_num_iterations = 30
s = uvs.api.Solver(opts=opts, max_batch=1)
rng = np.random.default_rng(77)
stats = {"persistent": [0, 0.0, 0.0], "fused": [0, 0.0, 0.0]}; rej_total = 0; longest = 0; t0 = time.time()
for i in range(N):
    w = synth.make_window(9000 + i, with_prior=bool(i % 2), marginalize_fn=(lambda win, flag: s.marginalize(win, flag)) if i % 2 else None).copy()
    amp = float(rng.choice([0.05, 0.2, 0.5]))
    w.pose[2:, :3] += amp * rng.standard_normal((9, 3)); w.inv_depth *= np.exp(amp * rng.standard_normal(len(w.inv_depth)))
    w.line_orth += 0.3 * amp * rng.standard_normal(w.line_orth.shape)
    so, ro = o.solve(w, opts)
    acc = list(ro.accepted[1:ro.num_iterations + 1]); rej_total += sum(1 for a in acc if a != 1)
    run = 0
    for a in acc:
        run = run + 1 if a != 1 else 0; longest = max(longest, run)
    for form in ("persistent", "fused"):
        if form == "fused": sg, rg, _ = s.large_solve_fused(w)
        else: sg, rg = s.solve(w)
        same = rg.num_iterations == ro.num_iterations and list(rg.accepted[:rg.num_iterations + 1]) == list(ro.accepted[:ro.num_iterations + 1]) and rg.termination == ro.termination
        dp, dq = pose_deltas(sg.pose, so.pose)
        if not same: stats[form][0] += 1; print("TRACE DIFF", form, i, rg.num_iterations, ro.num_iterations, list(rg.accepted[:12]), list(ro.accepted[:12]))
        else: stats[form][1] = max(stats[form][1], dp); stats[form][2] = max(stats[form][2], abs(rg.final_cost - ro.final_cost) / max(ro.final_cost, 1e-300))
print("%d windows, %d rejected / invalid steps in the oracle's traces (longest run %d), %.1f s" % (N, rej_total, longest, time.time() - t0))
for form, (bad, dp, dc) in stats.items():
    print("  %-10s trace differences %d; worst over identical traces: dp %.2e m, relative final cost %.2e" % (form, bad, dp, dc))
