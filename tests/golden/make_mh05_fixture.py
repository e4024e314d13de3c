"""Build-container only: turns the one evaluation fixture the reference holds -- the EuRoC MH_05_difficult ground truth that
benchmark_publisher replays (benchmark_publisher/config/MH_05_difficult/data.csv, parsed at benchmark_publisher_node.cpp:32-54 /
:67-141: `t[ns], p xyz, q wxyz, v xyz, bw xyz, ba xyz`) -- into tests/golden/mh05_groundtruth.npz.

The CSV prints six decimals, so every value is an exact multiple of 1e-6: the fixture stores the stamps (int64 ns) and the 16 value
columns as integer micro-units, first differences along time (they deflate to ~0.4 MB instead of 3.8 MB).  No row is dropped or
resampled; `uv-slam_amd.trajectory.load_groundtruth_fixture` rebuilds exactly the numbers `read_euroc_groundtruth` parses from the CSV
(checked below), and `write_euroc_groundtruth` prints the CSV back in the layout the reference parses.

    python tests/golden/make_mh05_fixture.py [/root/reference]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = os.path.join(ref, "benchmark_publisher", "config", "MH_05_difficult", "data.csv")
    stamps, rows = [], []
    with open(src) as f:
        for line in f:
            line = line.strip()
            if not line or line.startswith("#"):
                continue
            v = line.split(",")
            stamps.append(int(v[0]))
            rows.append([float(x) for x in v[1:17]])
    t = np.asarray(stamps, dtype=np.int64)
    val = np.asarray(rows, dtype=np.float64)
    micro = np.rint(val * 1e6).astype(np.int64)
    assert np.abs(micro / 1e6 - val).max() < 1e-9, "the CSV is expected to print six decimals"
    out = os.path.join(HERE, "mh05_groundtruth.npz")
    np.savez_compressed(out, source=np.array("benchmark_publisher/config/MH_05_difficult/data.csv (EuRoC MAV dataset ground truth, state_groundtruth_estimate0)"),
                        t0_ns=t[:1], dt_ns=np.diff(t).astype(np.int32), v0_micro=micro[:1], dv_micro=np.diff(micro, axis=0).astype(np.int32).T.copy())
    # the fixture must give back what the reference-layout reader parses from the CSV itself
    import importlib
    traj = importlib.import_module("uv-slam_amd.trajectory")
    a, b = traj.load_groundtruth_fixture(out), traj.read_euroc_groundtruth(src)
    for k in b:
        assert np.array_equal(a[k], b[k]), k
    print("%s: %d rows, %.3f s, %d bytes" % (out, len(t), (t[-1] - t[0]) / 1e9, os.path.getsize(out)))


if __name__ == "__main__":
    main()
