"""More randomised soaks (by hand on the GPU box): (1) marginalization of random solved windows vs the oracle (both kinds), (2) one
heterogeneous batch vs the same windows solved one by one (bitwise), (3) extreme shapes (tiny / big / skipped IMU blocks / no lines).
    python tests/gpu_soak_more.py [N]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, abi, synth, pose_deltas, prior_information, marginalization_reference
from oracle_binding import Oracle

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
o = Oracle()
s = uvs.api.Solver(max_batch=64, max_points=1300, max_point_obs=14000, max_lines=420, max_line_obs=4700)
rng = np.random.default_rng(777)

def rand_window(i, big=False, prior=None):
    npt = int(rng.integers(5, 1200 if big else 300)); nln = int(rng.integers(0, 400 if big else 80))
    return synth.make_window(31000 + i, n_points=npt, n_lines=nln, n_tagged=int(rng.integers(0, nln + 1)), pt_track=int(rng.integers(2, 11)), ln_track=int(rng.integers(5, 11)),
                             with_prior=bool(prior), marginalize_fn=(lambda win, flag: s.marginalize(win, flag)) if prior else None)

# ---- (1) marginalization
t0 = time.time(); worstH = worstb = 0.0; bad = 0
for i in range(N):
    w = rand_window(i, prior=bool(i % 2))
    st, rep = s.solve(w); post = w.with_state(st)
    for flag in (0, 1):
        pg, po = s.marginalize(post, flag), o.marginalize(post, flag)
        if pg.n != po.n or list(pg.block_kind) != list(po.block_kind) or list(pg.block_frame) != list(po.block_frame): bad += 1; print("  STRUCT DIFF", i, flag, pg.n, po.n); continue
        if pg.n == 0: continue
        Hg, bg, _ = prior_information(pg) if flag == 0 else (pg.J0().T @ pg.J0(), pg.J0().T @ pg.r0(), None)
        Ho, bo, _ = prior_information(po) if flag == 0 else (po.J0().T @ po.J0(), po.J0().T @ po.r0(), None)
        eH, eb = np.abs(Hg - Ho).max() / np.abs(Ho).max(), np.abs(bg - bo).max() / max(np.abs(bo).max(), 1e-300)
        worstH, worstb = max(worstH, eH), max(worstb, eb)
        if eH > 1e-5 or eb > 1e-4:      # (1e-6 / 1e-5 is the float64 floor of ill-conditioned windows: both sides are that far from an extended-precision Schur complement)
            bad += 1; msg = ""
            if flag == 0:      # who is right?  extended-precision Schur complement from the oracle's evaluation of the same state
                Ar, br, kp = marginalization_reference(post, o.evaluate(post, robust=True))
                _, _, cols = prior_information(pg); perm = [cols.index(c2) for c2 in kp]
                f = lambda H, b: (np.abs(H[np.ix_(perm, perm)] - Ar).max() / np.abs(Ar).max(), np.abs(b[perm] - br).max() / np.abs(br).max())
                msg = "  vs longdouble: hip H %.2e b %.2e | oracle H %.2e b %.2e" % (*f(Hg, bg), *f(Ho, bo))
            print("  MARG DIFF", i, flag, "H %.2e b %.2e n %d" % (eH, eb, pg.n), msg)
print("(1) marginalization: %d windows x 2 kinds, %d flagged, worst H %.2e b %.2e (relative to the largest entry)  [%.1f s]" % (N, bad, worstH, worstb, time.time() - t0))

# ---- (2) heterogeneous batch vs singles
t0 = time.time()
ws = [rand_window(1000 + i, prior=bool(i % 3 == 0)) for i in range(48)]
s.upload(ws); s.solve_resident(); states, reps = s.download()
nd = 0
for w, stb, rb in zip(ws, states, reps):
    st1, r1 = s.solve(w)
    if not (np.array_equal(st1.pose, stb.pose) and np.array_equal(st1.inv_depth, stb.inv_depth) and np.array_equal(st1.line_orth, stb.line_orth) and r1.final_cost == rb.final_cost): nd += 1
print("(2) heterogeneous batch of 48 vs one by one: %d windows differ bitwise  [%.1f s]" % (nd, time.time() - t0))

# ---- (3) extreme shapes against the oracle
t0 = time.time(); bad = 0; worst = 0.0
cases = []
for i in range(N // 2): cases.append(("big", rand_window(2000 + i, big=True)))
for i in range(N // 4):
    w = rand_window(3000 + i); k = int(rng.integers(0, 10))
    for b in w.imu[:]:
        if b["frame_i"] in (k, (k + 3) % 10): b["sum_dt"] = 11.0; b["skip"] = 1
    cases.append(("skipped imu", w))
for i in range(N // 4): cases.append(("tiny", synth.make_window(4000 + i, n_points=int(rng.integers(4, 12)), n_lines=int(rng.integers(0, 3)), n_tagged=0, pt_track=int(rng.integers(2, 6)), ln_track=5)))
for tag, w in cases:
    sg, rg = s.solve(w); so, ro = o.solve(w)
    same = rg.num_iterations == ro.num_iterations and list(rg.accepted[:rg.num_iterations + 1]) == list(ro.accepted[:ro.num_iterations + 1]) and rg.status == ro.status
    dp, dq = pose_deltas(sg.pose, so.pose)
    if not same or dp > 1e-6 or dq > 1e-6:
        bad += 1; print("  ", tag, len(w.inv_depth), len(w.line_orth), "same" if same else "TRACE DIFF", "dp %.2e dq %.2e" % (dp, dq), rg.status, ro.status, rg.num_iterations, ro.num_iterations, list(rg.accepted[:11]), list(ro.accepted[:11]))
    else: worst = max(worst, dp)
print("(3) extreme shapes: %d windows, %d flagged, worst dp %.2e m  [%.1f s]" % (len(cases), bad, worst, time.time() - t0))
