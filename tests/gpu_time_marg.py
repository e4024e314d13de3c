"""Wall-clock of uvs_marginalize / uvs_solve_window on the canonical window (run by hand on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, synth
s = uvs.api.Solver(max_batch=2)
w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
st, rep = s.solve(w)
post = w.with_state(st)
for flag in (0, 1):
    ts = []
    for _ in range(10):
        t = time.perf_counter(); p = s.marginalize(post, flag); ts.append(time.perf_counter() - t)
    print("marginalize flag %d: n %d  median %.3f ms  min %.3f ms" % (flag, p.n, 1e3 * np.median(ts), 1e3 * min(ts)))
ts = []
for _ in range(10):
    t = time.perf_counter(); s.solve(w); ts.append(time.perf_counter() - t)
print("solve (PCIe inclusive): median %.3f ms" % (1e3 * np.median(ts)))
if os.environ.get("UVS_MARG_PROFILE"):
    pass
ts = []
for _ in range(20):
    t = time.perf_counter(); s.upload([w]); ts.append(time.perf_counter() - t)
print("upload (pack + H2D) alone: median %.3f ms" % (1e3 * np.median(ts)))
ts = []
for _ in range(20):
    t = time.perf_counter(); s.solve_resident(); ts.append(time.perf_counter() - t)
print("solve_resident wall: median %.3f ms" % (1e3 * np.median(ts)))
ts = []
for _ in range(20):
    t = time.perf_counter(); s.download(); ts.append(time.perf_counter() - t)
print("download alone: median %.3f ms" % (1e3 * np.median(ts)))
