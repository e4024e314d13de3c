"""uvs_marginalize_batch against one-window calls (run by hand on the GPU box): wall clock for N post-solve canonical windows, both marginalization kinds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = uvs.api.Solver(max_batch=n)
base = [synth.make_window(1000 + i, with_prior=True, marginalize_fn=lambda w, f: s.marginalize(w, f)) for i in range(min(n, 32))]
wins = []
for i in range(n):
    w = base[i % len(base)]
    if i < len(base):
        st, _ = s.solve(w); base[i] = w.with_state(st)
    wins.append(base[i % len(base)])
for flag in (0, 1):
    flags = [flag] * n
    s.marginalize_batch(wins, flags)      # buffers, threads
    ts = []
    for _ in range(5):
        t = time.perf_counter(); pri, st = s.marginalize_batch(wins, flags); ts.append(time.perf_counter() - t)
    t1 = time.perf_counter()
    for w in wins[:32]: s.marginalize(w, flag)
    one = (time.perf_counter() - t1) / 32
    print("flag %d: uvs_marginalize_batch of %d windows: median %.3f ms (%.1f us per window, status ok %d); one-window uvs_marginalize: %.3f ms per window (python binding included in both)"
          % (flag, n, 1e3 * np.median(ts), 1e6 * np.median(ts) / n, sum(1 for x in st if x == 0), 1e3 * one))
