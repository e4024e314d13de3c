"""Randomised parity soak of the option / path variants (run by hand on the GPU box): ESTIMATE_TD, ESTIMATE_EXTRINSIC, both, and the
large-window grid path, each against the oracle on random windows.   python tests/gpu_soak_options.py [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, abi, synth, pose_deltas
from oracle_binding import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
o = Oracle()

def run(tag, td, ex, large, seed0):
    opts = abi.default_options(); opts.estimate_td = int(td); opts.estimate_extrinsic = int(ex)
    s = uvs.api.Solver(opts=opts, max_batch=2, max_points=3100, max_point_obs=34000, max_lines=820, max_line_obs=9000)
    rng = np.random.default_rng(seed0)
    bad = 0; worst = [0.0, 0.0, 0.0]
    for i in range(N):
        if large: npt, nln = int(rng.integers(300, 3000)), int(rng.integers(50, 800))
        else: npt, nln = int(rng.integers(20, 300)), int(rng.integers(0, 80))
        w = synth.make_window(seed0 + i, n_points=npt, n_lines=nln, n_tagged=int(rng.integers(0, nln + 1)), pt_track=int(rng.integers(3, 10)), ln_track=int(rng.integers(5, 10)))
        if ex:
            w = w.copy(); w.ex_pose = w.ex_pose.copy(); w.ex_pose[:3] += 0.01 * rng.standard_normal(3)
            q = w.ex_pose[3:] + 0.005 * rng.standard_normal(4); w.ex_pose[3:] = q / np.linalg.norm(q)
        if td: w = synth.add_time_offset(w, td_true=float(rng.uniform(-0.01, 0.01)), seed=i)
        sg, rg = (s.large_solve(w) if large else s.solve(w)); so, ro = o.solve(w, opts=opts)
        same = rg.num_iterations == ro.num_iterations and list(rg.accepted[:rg.num_iterations + 1]) == list(ro.accepted[:ro.num_iterations + 1])
        dp, dq = pose_deltas(sg.pose, so.pose); dc = abs(rg.final_cost - ro.final_cost) / max(ro.final_cost, 1e-300)
        if not same or dp > 1e-6 or dq > 1e-6:
            bad += 1; print("  ", tag, "window", i, (npt, nln), "same trace" if same else "TRACE DIFF", "dp %.2e dq %.2e" % (dp, dq), rg.num_iterations, ro.num_iterations, list(rg.accepted[:11]), list(ro.accepted[:11]))
        else: worst = [max(worst[0], dp), max(worst[1], dq), max(worst[2], dc)]
        if td and same: worst[0] = max(worst[0], abs(sg.td - so.td))
    s.close()
    print("%-22s %d windows, %d flagged; worst dp|dtd %.2e m, dq %.2e rad, cost %.2e" % (tag, N, bad, *worst))

t0 = time.time()
run("estimate_td", True, False, False, 21000)
run("estimate_extrinsic", False, True, False, 22000)
run("td + extrinsic", True, True, False, 23000)
run("large path", False, False, True, 24000)
print("%.1f s" % (time.time() - t0))
