"""The benchmarked configuration, every window against the oracle (run by hand on the GPU box): the 256 canonical windows of bench.py (W10-P150-L40-V3, each with
the n = 75 prior built by the product's own marginalization) solved as ONE resident batch, then one by one by the CPU oracle (one window per core).
python tests/gpu_soak_batch256.py [first seed]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from helpers import uvs, abi, synth, pose_deltas
from oracle_binding import Oracle

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0      # bench.py times window indices 0..255 (rank 0)
s = uvs.api.Solver(max_batch=256)
marg = lambda win, flag: s.marginalize(win, flag)
t0 = time.time()
ws = [synth.make_window(seed0 + i, with_prior=True, marginalize_fn=marg) for i in range(256)]
s.upload(ws); s.solve_resident(); states, reps = s.download()
o = Oracle()
jobs = [o.prepare(w) for w in ws]
with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex: rcs = list(ex.map(o.solve_prepared, jobs))
assert all(rc == 0 for rc in rcs)
for j in jobs: j[3].from_c(j[4])
refs = [(j[3], j[5]) for j in jobs]
same = 0; worst = dict(dp=0.0, dq=0.0, cost=0.0, invd=0.0, line=0.0); diffs = []
for i, (w, sg, rg, (so, ro)) in enumerate(zip(ws, states, reps, refs)):
    ok = rg.status == 0 and rg.num_iterations == ro.num_iterations and list(rg.accepted[:rg.num_iterations + 1]) == list(ro.accepted[:ro.num_iterations + 1]) and rg.termination == ro.termination
    if not ok: diffs.append((i, rg.status, rg.num_iterations, ro.num_iterations, list(rg.accepted[:11]), list(ro.accepted[:11]))); continue
    same += 1
    dp, dq = pose_deltas(sg.pose, so.pose)
    for k, v in (("dp", dp), ("dq", dq), ("cost", abs(rg.final_cost - ro.final_cost) / ro.final_cost), ("invd", np.abs(sg.inv_depth - so.inv_depth).max()), ("line", np.abs(sg.line_orth - so.line_orth).max())):
        worst[k] = max(worst[k], v)
print("256 benchmark windows (seeds %d..%d, n = 75 prior each) as one resident batch vs the oracle: identical LM trace in %d; worst over those: %s  [%.0f s]"
      % (seed0, seed0 + 255, same, {k: "%.2e" % v for k, v in worst.items()}, time.time() - t0))
for d in diffs[:10]: print("TRACE DIFF", d)
