"""Stage-by-stage GPU vs oracle comparison (run by hand on the GPU box: python tests/gpu_debug.py)."""
import sys, os, time, importlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import uvs, abi, synth, lm_reduced_system, unpad, pose_deltas
from oracle_binding import Oracle

o = Oracle()
s = uvs.api.Solver(max_batch=1024)
w = synth.make_window(0)
rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max())
for robust in (True, False):
    eg = s.evaluate(w, robust=robust); eo = o.evaluate(w, robust=robust)
    print("evaluate robust=%d cost gpu %.15g oracle %.15g" % (robust, eg.cost, eo.cost))
    for name in ("pt_r", "pt_J", "ln_r", "ln_J", "vp_r", "vp_J", "imu_r", "imu_J"):
        print("   %-6s relerr %.3e" % (name, rel(getattr(eg, name), getattr(eo, name))))
eo = o.evaluate(w, robust=True)
ref = lm_reduced_system(w, eo)
d = s.debug_first_iteration(w)
S = unpad(d["S"]); S = S + np.tril(S, -1).T
print("first-iter: cost %.15g (oracle %.15g) chol_ok %g" % (d["cost"], eo.cost, d["chol_ok"]))
print("   hd  relerr %.3e" % rel(unpad(d["hd"]), ref["hd"][:165]))
print("   dd  relerr %.3e" % rel(unpad(d["dd"]), ref["dd"][:165]))
print("   g   relerr %.3e" % rel(unpad(d["g"]), ref["g"]))
print("   S   relerr %.3e" % (np.abs(S - ref["S"]).max() / np.abs(ref["S"]).max()))
print("   step relerr %.3e" % rel(unpad(d["step"]), ref["step"][:165]))
if rel(S, ref["S"]) > 1e-6:
    E = np.abs(S - ref["S"]) / np.abs(ref["S"]).max()
    bad = np.argwhere(E > 1e-6)
    print("   bad S entries (first 20):", bad[:20].tolist())
t = time.time(); sg, rg = s.solve(w); tg = time.time() - t
t = time.time(); so, ro = o.solve(w); to = time.time() - t
print("solve: gpu %.4fs oracle %.4fs" % (tg, to))
print("  iters", rg.num_iterations, ro.num_iterations, "term", rg.termination, ro.termination, "status", rg.status)
tg_, to_ = rg.trace(), ro.trace()
for k in range(max(rg.num_iterations, ro.num_iterations) + 1):
    print("  %2d cost %.12g | %.12g  cand %.12g | %.12g  radius %.6g | %.6g acc %d | %d  mcc %.6g | %.6g" % (
        k, rg.cost[k], ro.cost[k], rg.candidate_cost[k], ro.candidate_cost[k], rg.radius[k], ro.radius[k], rg.accepted[k], ro.accepted[k],
        rg.model_cost_change[k], ro.model_cost_change[k]))
print("  pose deltas", pose_deltas(sg.pose, so.pose), "sb", np.abs(sg.speedbias - so.speedbias).max(),
      "invd", np.abs(sg.inv_depth - so.inv_depth).max(), "line", np.abs(sg.line_orth - so.line_orth).max())
ws = [synth.make_window(i) for i in range(256)]
s.upload(ws)
for it in range(3):
    ms = s.solve_resident()
    print("batch 256: %.3f ms -> %.0f solves/s" % (ms, 256 / (ms * 1e-3)))
s.upload(ws[:1])
for it in range(3):
    ms = s.solve_resident()
    print("single window: %.3f ms -> %.0f solves/s" % (ms, 1 / (ms * 1e-3)))
d = s.debug_first_iteration(w)
tot = sum(d["cycles"].values())
print("phase cycles (whole solve, thread 0 of the workgroup):")
for k, v in d["cycles"].items():
    print("   %-9s %12.0f  %5.1f%%" % (k, v, 100 * v / tot))
print("   total %.0f cycles" % tot)
print("sub-timers", d["sub_timers"])
