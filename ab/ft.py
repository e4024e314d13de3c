import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import uvs, synth
s = uvs.api.Solver(max_batch=4)
w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
d = s.debug_first_iteration(w)
st = d["sub_timers"]["cost_phase"]
print("lines pass A (thread 0, 2 chunks x 11): loads %.0f  geom %.0f  line residual+J %.0f  vp+stores %.0f ; obs phase total %.0f" % (st["stage_dx"], st["prior_residual"], st["observations"], st["imu"], d["cycles"]["obs"]))
