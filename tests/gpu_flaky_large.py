import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import uvs, abi, synth
import numpy as np
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
w53 = synth.make_window(53, n_points=400, n_lines=100, n_tagged=75)
w41 = synth.make_window(41)
def run(tag, s):
    st, rep = s.large_solve(w53)
    print(tag, rep.num_iterations, repr(rep.final_cost), [repr(float(c)) for c in rep.cost[:4]], flush=True)
if mode == "fresh":
    s = uvs.api.Solver(max_batch=2); run("fresh", s); s.close()
elif mode == "solve_then":
    s0 = uvs.api.Solver(max_batch=2); s0.solve(w41); s0.close()
    s = uvs.api.Solver(max_batch=2); run("after solve+close", s); s.close()
elif mode == "large_then":
    s0 = uvs.api.Solver(max_batch=2); s0.large_solve(w41); s0.close()
    s = uvs.api.Solver(max_batch=2); run("after large41+close", s); s.close()
elif mode == "both_then":
    s0 = uvs.api.Solver(max_batch=2); s0.solve(w41); s0.large_solve(w41); s0.close()
    s = uvs.api.Solver(max_batch=2); run("after solve+large41+close", s); s.close()
elif mode == "both_noclose":
    s0 = uvs.api.Solver(max_batch=2); s0.solve(w41); s0.large_solve(w41)
    s = uvs.api.Solver(max_batch=2); run("after solve+large41 (no close)", s); s.close()
elif mode == "same":
    s0 = uvs.api.Solver(max_batch=2); s0.solve(w41); s0.large_solve(w41); run("same solver", s0); run("same solver again", s0)
