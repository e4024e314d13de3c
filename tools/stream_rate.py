"""End-to-end rate of uvs_batch_stream WITHOUT torch in the process (bench.py imports torch first, and with it torch's bundled HIP / HSA runtime; this script lets the
library load the system's): `python tools/stream_rate.py [batches] [per_batch]`.  Prints ms per batch and solves/s of the second of two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pkg = importlib.import_module("uv-slam_amd")
api, synth = pkg.api, pkg.synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if os.environ.get("WITH_TORCH"):
    import torch      # noqa: F401  (its bundled runtime gets loaded first)
    torch.cuda.init()
s = api.Solver(max_batch=per)
windows = [synth.make_window(i, with_prior=True, marginalize_fn=lambda w, f: s.marginalize(w, f)) for i in range(per)]
s.stream(windows * 6, per, want_states=False)      # (every buffer set twice: the first batch of a set sizes its pinned buffer, the second packs in place)
for rep in range(int(os.environ.get('REPS', '3'))):
    _, reps, wall_ms = s.stream(windows * nb, per, want_states=False)
    print("torch loaded: %s   %.3f ms per batch   %.0f solves/s" % ("torch" in sys.modules, wall_ms / nb, nb * per / (wall_ms * 1e-3)))
with open("/proc/self/maps") as f:
    libs = sorted({l.split()[-1] for l in f if "libamdhip64" in l or "libhsa-runtime" in l})
print("\n".join(libs))
