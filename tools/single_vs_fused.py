"""One canonical window: the persistent single-workgroup kernel (k_solve) against the multi-workgroup fused loop of the large-window path
(landmark chunks on many compute units, one workgroup for the reduced solve).  Run on the GPU box."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
uvs = importlib.import_module("uv-slam_amd"); synth, api = uvs.synth, uvs.api
s = api.Solver(device=0, max_batch=1)
marg = lambda win, flag: s.marginalize(win, flag)
for prior in (False, True):
    w = synth.make_window(3, with_prior=prior, marginalize_fn=marg if prior else None)
    s.upload([w]); s.solve_resident()
    k = float(np.median([s.solve_resident() for _ in range(10)]))
    s.large_comm_init(None)
    st, rep, ms = s.large_solve_fused(w)
    loop = []
    for _ in range(8):
        st, rep, ms = s.large_solve_fused(w); wall = s.last_solve_ms
        loop.append((ms, wall))
    st2, rep2 = s.solve(w); st2, rep2 = s.solve(w); ksw = s.last_solve_ms
    print("prior %d: k_solve %.3f ms (uvs_solve_window call %.3f ms) | fused loop %.3f ms (uvs_large_solve_fused call %.3f ms) | iterations %d vs %d, final cost %.12g vs %.12g, max pose diff %.2e"
          % (prior, k, ksw, float(np.median([a for a, b in loop])), float(np.median([b for a, b in loop])), rep.num_iterations, rep2.num_iterations, rep.final_cost, rep2.final_cost,
             float(np.abs(st.pose - st2.pose).max())))
