"""By hand on the GPU box: relocalization blocks on windows far larger than the canonical one (dozens of landmark chunks, every landmark
matched), HIP vs oracle.   python tests/gpu_check_big_relo.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, synth, pose_deltas, quat_angle
from oracle_binding import Oracle
o = Oracle(); s = uvs.api.Solver(max_batch=2, max_points=1100, max_point_obs=12000, max_lines=320, max_line_obs=3400)
for kw in (dict(n_points=1000, n_lines=300, n_tagged=200, pt_track=10, ln_track=10), dict(n_points=600, n_lines=0, n_tagged=0, pt_track=11), dict(n_points=8, n_lines=0, n_tagged=0, pt_track=2)):
    w = synth.add_relocalization(synth.make_window(77, **kw), relo_frame=9, fraction=1.0, pixel_sigma=0.5, seed=1)
    sg, rg = s.solve(w); so, ro = o.solve(w)
    same = rg.num_iterations == ro.num_iterations and list(rg.accepted[:rg.num_iterations + 1]) == list(ro.accepted[:ro.num_iterations + 1])
    dp, dq = pose_deltas(sg.pose, so.pose)
    print(kw, "relo blocks", len(w.relo_lm), "status", rg.status, "trace same", same, "dp %.2e dq %.2e relo %.2e / %.2e cost rel %.2e" % (
        dp, dq, np.abs(sg.relo_pose[:3] - so.relo_pose[:3]).max(), quat_angle(sg.relo_pose[3:], so.relo_pose[3:]), abs(rg.final_cost - ro.final_cost) / ro.final_cost))
