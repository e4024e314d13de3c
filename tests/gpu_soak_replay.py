"""Closed-loop replay soak (by hand on the GPU box): several synthetic sequences through the HIP-backed and the oracle-backed state machine;
reports ATE against the ground truth for both and the largest per-frame deviation.   python tests/gpu_soak_replay.py [n_seeds] [n_frames]"""
import sys, os, pathlib, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs
from test_sequence_replay import _replay, _oracle_replay, ROOT, seqm
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tmp = pathlib.Path(tempfile.mkdtemp())
for seed in range(100, 100 + ns):
    seq = seqm.make_sequence(seed, n_frames=nf)
    t0 = time.time(); rg = _replay(os.path.join(ROOT, "uv-slam_amd", "libuvs_host.so"), seq, tmp, "hip"); tg = time.time() - t0
    t0 = time.time(); ro = _oracle_replay(seq, tmp); to = time.time() - t0
    Pt = seq.truth_pose[rg["frame"], :3]
    dp = np.linalg.norm(rg["P"] - ro["P"], axis=1)
    print("seed %d: %d frames, non-keyframes %d | ATE hip %.4f m oracle %.4f m | max |dP| %.2e (first 6: %.1e) | status ok %s | %.1f ms/frame hip, %.1f oracle" % (
        seed, len(rg["frame"]), int(rg["flag"].sum()), seqm.ate(rg["P"], Pt), seqm.ate(ro["P"], Pt), dp.max(), dp[:6].max(), bool(np.all(rg["status"] == 0)),
        1e3 * tg / len(rg["frame"]), 1e3 * to / len(ro["frame"])))
