"""Randomised parity soak of the relocalization blocks (run by hand on the GPU box): random windows (shape, noise, prior) with a random
relocalization frame / match fraction / pose offset / match noise, HIP library vs oracle.   python tests/gpu_soak_relo.py [N] [seed0] [td|ex|tdex|-] [persistent|fused]
(third argument: every window also estimates the camera / IMU time offset (ESTIMATE_TD), the extrinsic (ESTIMATE_EXTRINSIC: relo_Pose is then a second-level
block of the persistent kernel), or both)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, abi, synth, pose_deltas, quat_angle
from oracle_binding import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 31000
mode = sys.argv[3] if len(sys.argv) > 3 else ""
form = sys.argv[4] if len(sys.argv) > 4 else "persistent"      # "fused": the multi-workgroup fused loop (uvs_large_solve_fused; with ex: relo_Pose as a second-level block of k_large_solve, round 6)
with_td = "td" in mode; with_ex = "ex" in mode
opts = abi.default_options(); opts.estimate_td = 1 if with_td else 0; opts.estimate_extrinsic = 1 if with_ex else 0
o = Oracle(); s = uvs.api.Solver(opts=opts, max_batch=2, max_points=400, max_point_obs=4800, max_lines=120, max_line_obs=1320)
rng = np.random.default_rng(seed0)
s_prior = uvs.api.Solver(max_batch=2, max_points=400, max_point_obs=4800, max_lines=120, max_line_obs=1320)      # the previous window's marginalization (no time offset there)
worst = dict(dp=0.0, dq=0.0, cost=0.0, invd=0.0, relo_p=0.0, relo_q=0.0); mism = []; t0 = time.time(); done = 0
for i in range(N):
    npt = int(rng.integers(8, 320)); nln = int(rng.integers(0, 90)); ntag = int(rng.integers(0, nln + 1))
    ptt = int(rng.integers(2, 11)); lnt = int(rng.integers(5, 11))
    kw = dict(n_points=npt, n_lines=nln, n_tagged=ntag, pt_track=ptt, ln_track=lnt, noise=bool(rng.integers(0, 4)), pixel_sigma=float(rng.choice([0.2, 0.5, 1.5])))
    rk = dict(relo_frame=int(rng.integers(0, 10)), fraction=float(rng.uniform(0.05, 1.0)), offset=(float(rng.uniform(0.0, 0.5)), float(rng.uniform(0.0, 8.0))),
              pixel_sigma=float(rng.choice([0.0, 0.5, 2.0])))
    prior = bool(rng.integers(0, 2))
    try:
        w = synth.make_window(seed0 + i, with_prior=prior, marginalize_fn=(lambda win, flag: s_prior.marginalize(win, flag)) if prior else None, **kw)
        w = synth.add_relocalization(w, seed=seed0 + i, **rk)
        if with_td: w = synth.add_time_offset(w)
    except Exception as e:
        print("gen failed", i, kw, e); continue
    if len(w.relo_lm) == 0: continue
    done += 1
    if form == "fused": sg, rg, _ = s.large_solve_fused(w)
    else: sg, rg = s.solve(w)
    so, ro = o.solve(w, opts=opts)
    same = rg.num_iterations == ro.num_iterations and list(rg.accepted[:rg.num_iterations + 1]) == list(ro.accepted[:ro.num_iterations + 1]) and rg.termination == ro.termination
    dp, dq = pose_deltas(sg.pose, so.pose)
    dc = abs(rg.final_cost - ro.final_cost) / max(ro.final_cost, 1e-300)
    di = np.abs(sg.inv_depth - so.inv_depth).max()
    rp = np.abs(sg.relo_pose[:3] - so.relo_pose[:3]).max(); rq = quat_angle(sg.relo_pose[3:], so.relo_pose[3:])
    if not same or rg.status != 0:
        mism.append((i, kw, rk, prior, len(w.relo_lm), rg.status, rg.num_iterations, ro.num_iterations, list(rg.accepted[:11]), list(ro.accepted[:11]), dp, rp))
    else:
        for k, v in (("dp", dp), ("dq", dq), ("cost", dc), ("invd", di), ("relo_p", rp), ("relo_q", rq)): worst[k] = max(worst[k], v)
    if same and (dp > 1e-6 or rp > 1e-6):
        print("LARGE", i, kw, rk, prior, "n_relo %d dp %.2e relo %.2e %.2e" % (len(w.relo_lm), dp, rp, rq))
print("%d relocalization windows%s (%s form) in %.1f s; identical LM trace in %d; worst over those: %s" % (done, (" with ESTIMATE_TD" if with_td else "") + (" with ESTIMATE_EXTRINSIC" if with_ex else ""), form, time.time() - t0, done - len(mism), {k: "%.2e" % v for k, v in worst.items()}))
for m in mism[:20]: print("TRACE DIFF", m)
