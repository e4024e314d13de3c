"""Stage times of the marginalizations of a closed-loop replay (run by hand on the GPU box):
    UVS_MARG_PROFILE=1 python tests/gpu_time_marg_replay.py [frames] 2> marg_stages.txt
prints the replay's per-call times; the library's stage lines go to stderr."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
seqm, traj = uvs.sequence, uvs.trajectory
gt = traj.load_groundtruth_fixture(os.path.join(ROOT, "tests", "golden", "mh05_groundtruth.npz"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seq = seqm.make_groundtruth_sequence(gt, t_start=20.0, t_end=20.0 + 0.1 * n + 0.05)      # from the take-off on: mostly MARGIN_OLD
seqm.save(seq, "/tmp/marg_seq.bin")
lib = C.CDLL(os.path.join(ROOT, "uv-slam_amd", "libuvs_host.so"))
lib.uvs_host_replay_sequence.argtypes = [C.c_char_p, C.c_char_p]; lib.uvs_host_replay_sequence.restype = C.c_int
assert lib.uvs_host_replay_sequence(b"/tmp/marg_seq.bin", b"/tmp/marg_out.bin") == 0
r = seqm.load_result("/tmp/marg_out.bin")
tm = (C.c_double * 4)(); lib.uvs_host_replay_timing.argtypes = [C.POINTER(C.c_double)]; lib.uvs_host_replay_timing(tm)
print("windows %d (MARGIN_OLD %d, MARGIN_SECOND_NEW %d): optimization %.3f ms per call, solve %.3f, marginalization on the critical path %.3f" % (
    len(r["frame"]), int((r["flag"] == 0).sum()), int((r["flag"] == 1).sum()), tm[0], tm[1], tm[2]))
