"""Per-workgroup timeline of the LAST active k_large_chunks launch of a fused solve (UVS_LARGE_PROF, uvs_large_kernel.h: g_large_prof).  Run on the GPU box:
python tools/large_timeline.py [canonical|config3]"""
import importlib, os, sys, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
out = "/tmp/uvs_large_prof.bin"; os.environ["UVS_LARGE_PROF"] = out
os.environ["UVS_REDAMP"] = "0"      # every pass linearizes: the recorded launch (the last active one) is then a full linearization, not the re-damping after a rejected step
uvs = importlib.import_module("uv-slam_amd"); synth, api = uvs.synth, uvs.api
which = sys.argv[1] if len(sys.argv) > 1 else "config3"
if which == "canonical":
    s = api.Solver(max_batch=1); w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
else:
    w = synth.make_window(70, n_points=20000, n_lines=5000, n_tagged=3750)
    s = api.Solver(device=0, max_batch=1, max_points=20008, max_point_obs=240000, max_lines=5008, max_line_obs=60000)
for _ in range(3): st, rep, ms = s.large_solve_fused(w)
raw = open(out, "rb").read(); n_wg, n_chunks = struct.unpack("ii", raw[:8]); t = np.frombuffer(raw[8:], dtype=np.int64).reshape(-1, 8)[:n_wg].astype(float)
t0 = t[:, 0].min(); us = lambda v: (v - t0) / 100.0
cw = t[:-1]; fw = t[-1]
print("%s: loop %.3f ms, %d iterations; last active k_large_chunks launch: %d chunk workgroups for %d chunks + the frame workgroup" % (which, ms, rep.num_iterations, n_wg - 1, n_chunks))
print("  start of the workgroups: %.1f .. %.1f us after the first" % (us(cw[:, 0]).min(), us(cw[:, 0]).max()))
names = ["prologue (state, rotations)", "first chunk", "remaining chunks", "part sums", "canonical image + partial write"]
for k, nm in enumerate(names):
    d = (cw[:, k + 1] - cw[:, k]) / 100.0
    print("  %-32s mean %6.1f us   min %6.1f   max %6.1f" % (nm, d.mean(), d.min(), d.max()))
tot = (cw[:, 5] - cw[:, 0]) / 100.0
print("  workgroup total                  mean %6.1f us   min %6.1f   max %6.1f ; chunks per workgroup %d .. %d" % (tot.mean(), tot.min(), tot.max(), int(cw[:, 7].min()), int(cw[:, 7].max())))
print("  last chunk workgroup ends %.1f us after the first start; frame workgroup: start %.1f, end %.1f (%.1f us)" % (us(cw[:, 5]).max(), us(fw[0]), us(fw[5]), (fw[5] - fw[0]) / 100.0))
