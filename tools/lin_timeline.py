"""Prints the per-wave step log of the linearizations of one solve (UVS_DEBUG_LIN_TIMELINE=<file> python tests/gpu_debug_prior.py; debug == 5 in k_solve).
usage: python tools/lin_timeline.py <file> [linearization index, default 1 (the second: caches warm)]
stamps: 1 chunk start, 2 after the entry barrier, 3 pass A done (this wave), 4 after barrier, 5 pass B done, 6 after barrier, 7 after the last barrier of the
evaluation half, 8 gather walk starts, 10 gather walk done; 512-thread build: 20 lin_prep done, 21 chunk loop done, 22 IMU tiles, 23 S zeroed, 24 part-0 rows in,
25 IMU tiles in, 26 prior in, 27 damping / norms done; 30..37 the solve and candidate phases of the iteration that follows."""
import sys, numpy as np
TL = 4096
a = np.fromfile(sys.argv[1], dtype=np.int64).reshape(8, TL, 2)
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
names = {1: "start", 2: "B0", 3: "passA", 4: "B1", 5: "passB", 6: "B2", 7: "B3", 8: "gather>", 10: "gather<", 20: "prep", 21: "chunks", 22: "imu", 23: "zero", 24: "part0", 25: "imuadd", 26: "prior", 27: "finish",
         50: "ds>", 51: "ds<", 52: "E1", 53: "xchg", 54: "stored", 55: "fold", 56: "imuB3", 40: "imu:zeroed", 41: "imu:B", 42: "imu:raw", 43: "imu:B2", 30: "solve>", 31: "chol", 32: "trsv", 33: "backsub", 34: "staged", 35: "priorq", 36: "cost", 37: "reduced"}
waves = [w for w in range(8) if a[w, 0, 0] != 0]
ev = [w for w in waves if (a[w, :, 0] == 1).any()]      # waves that log evaluation stamps
# linearization boundaries on the first evaluator wave: stamp 1 that follows a stamp 10 / 7 by a long gap -> use chunk count: count stamps "1"
w0 = ev[0]
ids = a[w0, :, 0]; n0 = int((ids != 0).sum())
starts = [i for i in range(n0) if ids[i] == 1]
# chunks per linearization: number of "1" stamps until the gap between consecutive chunk starts exceeds 3x the median
gaps = np.diff([a[w0, i, 1] for i in starts])
med = np.median(gaps)
lin_first = [0] + [k + 1 for k, g in enumerate(gaps) if g > 3 * med]
lo = starts[lin_first[which]]; hi = starts[lin_first[which + 1]] if which + 1 < len(lin_first) else n0
t0 = a[w0, lo, 1]; t1 = a[w0, hi, 1] if hi < n0 else a[w0, n0 - 1, 1]
print(f"linearization {which}: {len(lin_first)} found; {(t1 - t0)} cycles from its first chunk start to the next linearization's")
for w in waves:
    sel = [(int(a[w, i, 0]), int(a[w, i, 1] - t0)) for i in range(TL) if a[w, i, 0] != 0 and t0 <= a[w, i, 1] < t1]
    print(f"wave {w}: " + " ".join(f"{names.get(i, i)}@{t}" for i, t in sel))
