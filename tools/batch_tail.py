"""Where does the gap between one resident window (one workgroup) and the 256-window launch come from: the slowest window of the
batch (the launch ends with it) or contention between 256 concurrently running workgroups?  Run on the GPU box."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
uvs = importlib.import_module("uv-slam_amd"); synth, api = uvs.synth, uvs.api
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
solver = api.Solver(device=0, max_batch=B)
marg = lambda win, flag: solver.marginalize(win, flag)
wins = [synth.make_window(i, with_prior=True, marginalize_fn=marg) for i in range(B)]
def timed(ws, n=5):
    solver.upload(ws); solver.solve_resident(); solver.solve_resident()
    return float(np.median([solver.solve_resident() for _ in range(n)]))
full = timed(wins, 10)
single = np.array([timed([w], 3) for w in wins])
print(f"batch {B}: {full:.4f} ms")
print(f"single windows: mean {single.mean():.4f}  min {single.min():.4f}  max {single.max():.4f}  p50 {np.median(single):.4f}  p90 {np.percentile(single, 90):.4f} ms")
order = np.argsort(single)
for n in (8, 32, 64, 128):
    sub = [wins[i] for i in order[-n:]]      # the n slowest windows together: same tail, less contention
    print(f"  the {n} slowest together: {timed(sub):.4f} ms   (slowest alone {single[order[-1]]:.4f})")
sub = [wins[order[-1]]] * B
print(f"  the slowest window x {B}: {timed(sub):.4f} ms")
sub = [wins[order[B // 2]]] * B
print(f"  the median window x {B}: {timed(sub):.4f} ms  (alone {single[order[B // 2]]:.4f})")
