import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import uvs, synth
from oracle_binding import Oracle
import numpy as np
w = synth.make_window(40, n_points=900, n_lines=240, n_tagged=180)
s = uvs.api.Solver(max_batch=2)
o = Oracle()
so, ro = o.solve(w)
s1, r1 = s.solve(w)
s2, r2 = s.large_solve(w)
print("oracle ", ro.num_iterations, list(ro.accepted[:11]), [float(c) for c in ro.cost[:5]])
print("k_solve", r1.num_iterations, list(r1.accepted[:11]), [float(c) for c in r1.cost[:5]])
print("large  ", r2.num_iterations, list(r2.accepted[:11]), [float(c) for c in r2.cost[:5]])
