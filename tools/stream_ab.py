"""Which configuration of uvs_batch_stream should be the default?  ONE process, the same 256 windows, the configurations ALTERNATED run by run (so that a busy
minute of the shared host hits all of them), >= 8 runs each; prints every run, then median / quartiles per configuration.

    python tools/stream_ab.py [rounds = 8] [batches per run = 32] [windows per batch = 256]

The knobs (UVS_STREAM_CHAIN, UVS_STREAM_SETS, UVS_PACK_THREADS, UVS_PACK_PIN) are read by the library per call, which is what lets one process alternate them.
No torch in the process (tools/stream_rate.py explains why that matters)."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("uv-slam_amd")
api, synth = pkg.api, pkg.synth
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
per = int(sys.argv[3]) if len(sys.argv) > 3 else 256
CONFIGS = [("default (un-chained, 32 packing threads)", {"UVS_STREAM_CHAIN": "0"}), ("UVS_STREAM_CHAIN=1", {"UVS_STREAM_CHAIN": "1"}), ("UVS_PACK_THREADS=64", {"UVS_STREAM_CHAIN": "0", "UVS_PACK_THREADS": "64"}),
           ("UVS_PACK_THREADS=16", {"UVS_STREAM_CHAIN": "0", "UVS_PACK_THREADS": "16"}), ("UVS_STREAM_SETS=4", {"UVS_STREAM_CHAIN": "0", "UVS_STREAM_SETS": "4"}), ("UVS_STREAM_CHAIN=1 UVS_PACK_THREADS=64", {"UVS_STREAM_CHAIN": "1", "UVS_PACK_THREADS": "64"})]
if os.environ.get("UVS_AB_EXTRA"):      # e.g. UVS_AB_EXTRA="UVS_PACK_PIN=1;UVS_PACK_PIN=1,UVS_STREAM_CHAIN=0"
    for spec in os.environ["UVS_AB_EXTRA"].split(";"):
        CONFIGS.append((spec.replace(",", " "), dict(kv.split("=") for kv in spec.split(","))))
s = api.Solver(max_batch=per)
windows = [synth.make_window(i, with_prior=True, marginalize_fn=lambda w, f: s.marginalize(w, f)) for i in range(per)]
os.environ["UVS_STREAM_SETS"] = "4"; s.stream(windows * 8, per, want_states=False); del os.environ["UVS_STREAM_SETS"]      # every buffer set twice: the first batch of a set sizes its pinned buffer, the second packs in place
rates = {name: [] for name, _ in CONFIGS}
for r in range(rounds):
    order = CONFIGS[r % len(CONFIGS):] + CONFIGS[:r % len(CONFIGS)]      # rotate who goes first
    for name, env in order:
        for k, v in env.items(): os.environ[k] = v
        _, reps, wall_ms = s.stream(windows * nb, per, want_states=False)
        for k in env: del os.environ[k]
        rate = nb * per / (wall_ms * 1e-3)
        rates[name].append(rate)
        print("round %2d  %-42s %.3f ms per batch   %.0f solves/s" % (r, name, wall_ms / nb, rate), flush=True)
print("\n%-42s %10s %10s %10s %10s %10s   (solves/s over %d runs of %d batches x %d windows)" % ("configuration", "median", "q25", "q75", "min", "max", rounds, nb, per))
for name, v in sorted(rates.items(), key=lambda kv: -np.median(kv[1])):
    v = np.asarray(v)
    print("%-42s %10.0f %10.0f %10.0f %10.0f %10.0f" % (name, np.median(v), np.percentile(v, 25), np.percentile(v, 75), v.min(), v.max()))
print("host: %d hardware threads, load average %s" % (os.cpu_count(), open("/proc/loadavg").read().strip()))
