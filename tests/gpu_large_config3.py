"""BASELINE configs[3] at full size on ONE GPU: a 10-KF window with 20 000 point + 5 000 line landmarks through the landmark-sharded
path (k_large_chunks grid + reduce + single-workgroup reduced solve + grid back-substitution).  Run by hand on the GPU box:
    python tests/gpu_large_config3.py [n_points n_lines]
Prints timing and convergence; the 2/4/8-GPU runs are the driver's (api.Solver.large_solve(dist=...))."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import uvs, synth
import numpy as np
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
nln = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
t = time.time(); w = synth.make_window(70, n_points=npts, n_lines=nln, n_tagged=(3 * nln) // 4); tg = time.time() - t
print("window: %d points / %d obs, %d lines / %d obs (generated in %.1f s)" % (len(w.inv_depth), len(w.pt_lm), len(w.line_orth), len(w.ln_lm), tg))
s = uvs.api.Solver(max_batch=1, max_points=npts + 8, max_point_obs=12 * npts, max_lines=nln + 8, max_line_obs=12 * nln)
for rep_i in range(3):
    t = time.time(); st, rep = s.large_solve(w); dt = time.time() - t
    print("large_solve: %.1f ms  iterations %d  accepted %s  cost %.6g -> %.6g  status %d" % (dt * 1e3, rep.num_iterations, list(rep.accepted[:11]), rep.initial_cost, rep.final_cost, rep.status))

# split: host packing + upload (uvs_large_begin) vs the resident LM loop (kernels + one small read-back per iteration) vs download
import ctypes as C
from helpers import abi
L = uvs.api.lib()
wc, keep = w.to_c(); st2 = abi.State(len(w.inv_depth), len(w.line_orth)); sc = st2.alloc_c(); rep2 = abi.Report()
for rep_i in range(2):
    t0 = time.perf_counter(); assert L.uvs_large_begin(s._h, C.byref(wc)) == 0; t1 = time.perf_counter()
    while not L.uvs_large_done(s._h):
        if L.uvs_large_need_linearize(s._h): assert L.uvs_large_linearize(s._h) == 0
        assert L.uvs_large_step(s._h) == 0
        assert L.uvs_large_decide(s._h) == 0
    t2 = time.perf_counter(); L.uvs_large_finish(s._h, C.byref(sc), C.byref(rep2)); t3 = time.perf_counter()
    print("begin (pack + upload) %.2f ms | resident LM loop %.2f ms (%d iterations) | finish (download) %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, rep2.num_iterations, (t3 - t2) * 1e3))
