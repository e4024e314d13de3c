import sys, os, time, ctypes as C
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np
from helpers import uvs, synth, abi
s = uvs.api.Solver(max_batch=2, max_points=1000, max_point_obs=11000, max_lines=1000, max_line_obs=11000)
w = synth.make_window(5, n_points=195, n_lines=38, n_tagged=28, with_prior=True, marginalize_fn=lambda win, f: s.marginalize(win, f))
wc, keep = w.to_c()
st = abi.State(len(w.inv_depth), len(w.line_orth)); sc = st.alloc_c(); rep = abi.Report(); lm = C.c_float(0)
L = uvs.api.lib()
for k in range(5): L.uvs_large_solve_fused(s._h, C.byref(wc), C.byref(sc), C.byref(rep), C.byref(lm))
ts, ls = [], []
for k in range(30):
    t = time.perf_counter(); L.uvs_large_solve_fused(s._h, C.byref(wc), C.byref(sc), C.byref(rep), C.byref(lm)); ts.append(time.perf_counter() - t); ls.append(lm.value)
print("195 points / 38 lines, n = 75 prior: uvs_large_solve_fused wall %.3f ms (min %.3f), device loop (events) %.3f ms, iterations %d" % (1e3 * np.median(ts), 1e3 * min(ts), np.median(ls), rep.num_iterations))
os.environ["UVS_PACK_PROFILE"] = "1"
