"""Randomised parity soak (run by hand on the GPU box): N random windows of varying shape (landmark counts, track lengths, noise,
with / without prior, VP share) solved by the HIP library and by the oracle; reports the worst pose / landmark / cost deviation and
every window whose iteration count or accept pattern differs.   python tests/gpu_soak.py [N] [seed0] [persistent|fused]
(the third argument picks the single-window form: the persistent kernel, default, or the multi-workgroup fused loop)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, synth, pose_deltas
from oracle_binding import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
fused = len(sys.argv) > 3 and sys.argv[3] == "fused"
o = Oracle(); s = uvs.api.Solver(max_batch=2, max_points=400, max_point_obs=4400, max_lines=120, max_line_obs=1320)
rng = np.random.default_rng(seed0)
worst = dict(dp=0.0, dq=0.0, cost=0.0, invd=0.0, line=0.0); mism = []; t0 = time.time()
for i in range(N):
    npt = int(rng.integers(8, 320)); nln = int(rng.integers(0, 90)); ntag = int(rng.integers(0, nln + 1))
    ptt = int(rng.integers(2, 11)); lnt = int(rng.integers(5, 11))
    kw = dict(n_points=npt, n_lines=nln, n_tagged=ntag, pt_track=ptt, ln_track=lnt, noise=bool(rng.integers(0, 4)), pixel_sigma=float(rng.choice([0.2, 0.5, 1.5])))
    prior = bool(rng.integers(0, 2))
    try:
        w = synth.make_window(seed0 + i, with_prior=prior, marginalize_fn=(lambda win, flag: s.marginalize(win, flag)) if prior else None, **kw)
    except Exception as e:
        print("gen failed", i, kw, e); continue
    if fused: sg, rg, _ = s.large_solve_fused(w)
    else: sg, rg = s.solve(w)
    so, ro = o.solve(w)
    same = rg.num_iterations == ro.num_iterations and list(rg.accepted[:rg.num_iterations + 1]) == list(ro.accepted[:ro.num_iterations + 1]) and rg.termination == ro.termination
    dp, dq = pose_deltas(sg.pose, so.pose)
    dc = abs(rg.final_cost - ro.final_cost) / max(ro.final_cost, 1e-300)
    di = np.abs(sg.inv_depth - so.inv_depth).max() if npt else 0.0
    dl = np.abs(sg.line_orth - so.line_orth).max() if nln else 0.0
    if not same:
        mism.append((i, kw, prior, rg.num_iterations, ro.num_iterations, list(rg.accepted[:11]), list(ro.accepted[:11]), rg.termination, ro.termination, dp, dq))
    else:
        worst["dp"] = max(worst["dp"], dp); worst["dq"] = max(worst["dq"], dq); worst["cost"] = max(worst["cost"], dc); worst["invd"] = max(worst["invd"], di); worst["line"] = max(worst["line"], dl)
    if dl > 1e-5:
        k = int(np.argmax(np.abs(sg.line_orth - so.line_orth).max(axis=1)))
        nob = int((w.ln_lm == k).sum())
        print("LINE", i, kw, prior, "dl %.2e line %d (%d obs) gpu %s oracle %s" % (dl, k, nob, sg.line_orth[k], so.line_orth[k]))
    if dp > 1e-6 or dq > 1e-6:
        print("LARGE", i, kw, prior, "dp %.2e dq %.2e" % (dp, dq), "same" if same else "DIFF", rg.num_iterations, ro.num_iterations)
print("%d windows in %.1f s; identical LM trace in %d; worst over those: %s" % (N, time.time() - t0, N - len(mism), {k: "%.2e" % v for k, v in worst.items()}))
for m in mism[:20]: print("TRACE DIFF", m)
