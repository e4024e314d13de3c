import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import uvs, synth
s = uvs.api.Solver(max_batch=4)
w = synth.make_window(7, n_points=60, n_lines=16, n_tagged=12)
st, rep = s.solve(w)
print("status", rep.status, "its", rep.num_iterations, "cost", rep.initial_cost, "->", rep.final_cost, list(rep.accepted[:11]))
