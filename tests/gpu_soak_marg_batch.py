"""Randomised soak of uvs_marginalize_batch against the one-window call (run by hand on the GPU box): random window shapes, with / without a prior, both marginalization kinds,
in batches of 32.   python tests/gpu_soak_marg_batch.py [N] [seed0]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, abi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 192
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 41000
s = uvs.api.Solver(max_batch=2, max_points=400, max_point_obs=4800, max_lines=120, max_line_obs=1320)
rng = np.random.default_rng(seed0)
wins, flags = [], []
t0 = time.time()
for i in range(N):
    kw = dict(n_points=int(rng.integers(8, 320)), n_lines=int(rng.integers(0, 90)), pt_track=int(rng.integers(2, 11)), ln_track=int(rng.integers(5, 11)), noise=bool(rng.integers(0, 4)))
    kw["n_tagged"] = int(rng.integers(0, kw["n_lines"] + 1))
    prior = bool(rng.integers(0, 4))
    try: w = synth.make_window(seed0 + i, with_prior=prior, marginalize_fn=(lambda win, f: s.marginalize(win, f)) if prior else None, **kw)
    except Exception as e: print("gen failed", i, kw, e); continue
    st, rep = s.solve(w)
    wins.append(w.with_state(st)); flags.append(int(rng.integers(0, 2)) if prior else 0)
worst = dict(H=0.0, b=0.0); bad = 0; n_cmp = 0; n_fb = 0
for b0 in range(0, len(wins), 32):
    ws, fl = wins[b0:b0 + 32], flags[b0:b0 + 32]
    single = [s.marginalize(w, f) for w, f in zip(ws, fl)]
    batch, status = s.marginalize_batch(ws, fl)
    for k, (p1, pb, stc) in enumerate(zip(single, batch, status)):
        if stc != 0 or pb.n != p1.n or pb.n_blocks != p1.n_blocks or list(pb.block_idx[:p1.n_blocks]) != list(p1.block_idx[:p1.n_blocks]) or not np.array_equal(np.asarray(pb.x0[:9 * p1.n_blocks]), np.asarray(p1.x0[:9 * p1.n_blocks])):
            bad += 1; print("MISMATCH (tables)", b0 + k, fl[k], stc, pb.n, p1.n); continue
        if p1.n == 0: continue
        H1, Hb = p1.J0().T @ p1.J0(), pb.J0().T @ pb.J0(); b1, bb = p1.J0().T @ p1.r0(), pb.J0().T @ pb.r0()
        eH, eb = np.abs(Hb - H1).max() / np.abs(H1).max(), np.abs(bb - b1).max() / max(1.0, np.abs(b1).max())
        n_cmp += 1; worst["H"] = max(worst["H"], eH); worst["b"] = max(worst["b"], eb)
        if eH > 1e-6 or eb > 1e-5: bad += 1; print("MISMATCH (values)", b0 + k, fl[k], "%.2e %.2e" % (eH, eb))
print("%d random windows (%d MARGIN_SECOND_NEW) through uvs_marginalize_batch in batches of 32 vs the one-window call in %.1f s: %d compared, %d flagged; worst H %.2e, b %.2e (relative)" % (
    len(wins), sum(flags), time.time() - t0, n_cmp, bad, worst["H"], worst["b"]))
