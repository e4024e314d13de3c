import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import uvs, synth
s = uvs.api.Solver(max_batch=4)
w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
d = s.debug_first_iteration(w)
print("gather phase", d["cycles"]["gather"], "per-wave point-gather busy", d["sub_timers"]["chol_busy_per_wave"], "total", sum(d["cycles"].values()))
