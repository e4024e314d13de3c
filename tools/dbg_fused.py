import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np
from helpers import uvs, synth
w = synth.make_window(54, n_points=900, n_lines=200, n_tagged=150)
for rep_ in range(2):
    s = uvs.api.Solver(max_batch=2, max_points=1000, max_point_obs=12000, max_lines=256, max_line_obs=3000)
    st0, rep0 = s.large_solve(w)
    st0b, rep0b = s.large_solve(w)
    s.large_comm_init(None)
    st1, rep1, ms = s.large_solve_fused(w)
    st2, rep2, ms2 = s.large_solve_fused(w)
    s.close()
    n = rep0.num_iterations
    print("host", np.array(rep0.cost[:n+1]))
    print("host again == ", np.array_equal(np.array(rep0.cost[:n+1]), np.array(rep0b.cost[:n+1])))
    print("fused", np.array(rep1.cost[:n+1]))
    print("fused again ==", np.array_equal(np.array(rep1.cost[:n+1]), np.array(rep2.cost[:n+1])))
    print("rel", np.abs(np.array(rep1.cost[:n+1]) - np.array(rep0.cost[:n+1])) / np.array(rep0.cost[:n+1]))
    print(list(rep0.accepted[:n+1]), list(rep1.accepted[:n+1]))
