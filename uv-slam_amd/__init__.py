"""uv-slam_amd -- MI355X-native sliding-window back-end (UV-SLAM `Estimator::optimization()` hot path).

Layout:
  csrc/      hand-written HIP (gfx950) kernels + the C ABI of include/uvs_solver.h  -> libuvs_solver.so
  host/      C++ mirror of the reference's Estimator / factor API above the C ABI   -> libuvs_host.so
  abi.py     ctypes mirror of the C ABI structs
  api.py     thin python binding of libuvs_solver.so (used by tests / bench.py)
  synth.py   synthetic "W10-P150-L40-V3" window generator
  trajectory.py  result file (TUM lines, visualization.cpp:195-207), EuRoC ground-truth CSV (benchmark_publisher_node.cpp:32-54), ATE

The directory name contains a hyphen (it is the name the build contract asks for), so import it with
`importlib.import_module("uv-slam_amd")`.
"""
from . import abi, synth, dist, sequence, trajectory  # noqa: F401


def __getattr__(name):
    if name == "api":
        import importlib
        return importlib.import_module(__name__ + ".api")
    raise AttributeError(name)
