"""Per-phase cycles of a window WITH the n = 75 marginalization prior (the bench workload); run by hand on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import uvs, synth
s = uvs.api.Solver(max_batch=4)
w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
print("prior n", w.prior.n if w.prior is not None else 0)
d = s.debug_first_iteration(w)
tot = sum(d["cycles"].values())
for k, v in d["cycles"].items():
    print("   %-11s %10.0f %5.1f%%" % (k, v, 100 * v / tot))
print("   total %.0f" % tot)
s.upload([w])
print("single window ms", [round(s.solve_resident(), 3) for _ in range(4)])
st, rep = s.solve(w)
print("iters", rep.num_iterations, "accepted", list(rep.accepted[:11]), "final", rep.final_cost, "initial", rep.initial_cost)
print("sub-timers", d["sub_timers"])
