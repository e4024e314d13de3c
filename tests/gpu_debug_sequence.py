"""Per-frame comparison of the HIP-backed and the oracle-backed sequence replay (run by hand on the GPU box)."""
import sys, os, ctypes as C, tempfile, pathlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs
from test_sequence_replay import _replay, _oracle_replay, ROOT, seqm
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 36
seq = seqm.make_sequence(seed, n_frames=n)
tmp = pathlib.Path(tempfile.mkdtemp())
ro = _oracle_replay(seq, tmp)
rg = _replay(os.path.join(ROOT, "uv-slam_amd", "libuvs_host.so"), seq, tmp, "hip")
for i in range(len(ro["frame"])):
    print("%3d flag %d/%d it %2d/%2d pts %3d/%3d ln %2d/%2d  cost0 %.9g | %.9g  cost %.9g | %.9g  dP %.3e" % (
        ro["frame"][i], rg["flag"][i], ro["flag"][i], rg["iterations"][i], ro["iterations"][i], rg["n_points"][i], ro["n_points"][i], rg["n_lines"][i], ro["n_lines"][i],
        rg["initial_cost"][i], ro["initial_cost"][i], rg["final_cost"][i], ro["final_cost"][i], np.linalg.norm(rg["P"][i] - ro["P"][i])))
