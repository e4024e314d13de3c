"""Phase cycles of ONE window's solve while the other 255 compute units run a resident batch (a second solver handle on its own stream,
driven from a background thread): which phases stretch when 256 workgroups share the memory system?  Run on the GPU box."""
import importlib, os, sys, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
uvs = importlib.import_module("uv-slam_amd"); synth, api = uvs.synth, uvs.api
bg = api.Solver(device=0, max_batch=256); fg = api.Solver(device=0, max_batch=1)
wins = [synth.make_window(i) for i in range(1, 256)]
w = synth.make_window(0)
def phases(n=6):
    acc = None
    for _ in range(n):
        d = fg.debug_first_iteration(w)
        v = np.array(list(d["cycles"].values())); acc = v if acc is None else acc + v
    return list(d["cycles"].keys()), acc / n
names, quiet = phases()
bg.upload(wins)
stop = False
def loop():
    while not stop: bg.solve_resident()
t = threading.Thread(target=loop); t.start()
import time; time.sleep(0.2)
_, busy = phases(12)
stop = True; t.join()
print("%-10s %12s %12s %8s" % ("phase", "alone", "255 busy", "ratio"))
for n, a, b in zip(names, quiet, busy):
    if a > 0: print("%-10s %12.0f %12.0f %8.3f" % (n, a, b, b / a))
print("%-10s %12.0f %12.0f %8.3f" % ("total", quiet.sum(), busy.sum(), busy.sum() / quiet.sum()))
