import sys, os
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from helpers import uvs, synth
s = uvs.api.Solver(max_batch=4)
w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
d = s.debug_first_iteration(w)
tot = sum(d["cycles"].values())
print("chol_diag", d["cycles"]["chol_diag"], "total", tot)
print("timers", d["sub_timers"])
