#!/usr/bin/env python3
"""ATE of an estimator result file against an EuRoC ground-truth CSV.

    python tools/ate.py <result.txt> <data.csv | tests/golden/mh05_groundtruth.npz>

result.txt: the file the reference's pubOdometry appends to (utility/visualization.cpp:195-207; the host mirror's replay writes it when
UVS_VINS_RESULT_PATH is set); data.csv: the layout benchmark_publisher parses (benchmark_publisher_node.cpp:32-54), e.g. the reference's
benchmark_publisher/config/MH_05_difficult/data.csv -- or the committed fixture of that file, tests/golden/mh05_groundtruth.npz.  Association and alignment: uv-slam_amd/trajectory.py.
"""
import importlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) != 3:
        print(__doc__); return 2
    traj = importlib.import_module("uv-slam_amd.trajectory")
    print(json.dumps(traj.ate(sys.argv[1], sys.argv[2])))
    return 0


if __name__ == "__main__":
    sys.exit(main())
