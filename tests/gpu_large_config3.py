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
