"""Open-loop check of every window of an oracle-backed sequence replay: HIP solve / marginalize vs oracle solve / marginalize on the SAME
recorded window (UVS_DUMP_WINDOWS record hook).  Run by hand on the GPU box."""
import sys, os, tempfile, pathlib, glob
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from helpers import uvs, abi, pose_deltas
from oracle_binding import Oracle
from test_sequence_replay import _oracle_replay, seqm
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 36
seq = seqm.make_sequence(seed, n_frames=n)
tmp = pathlib.Path(tempfile.mkdtemp())
os.environ["UVS_DUMP_WINDOWS"] = str(tmp)
ro = _oracle_replay(seq, tmp)
del os.environ["UVS_DUMP_WINDOWS"]
o = Oracle(); s = uvs.api.Solver(max_batch=2)
for k, path in enumerate(sorted(glob.glob(str(tmp / "window_*.bin")))):
    w = abi.Window.load(path)
    sg, rg = s.solve(w); so, rr = o.solve(w)
    dp, dq = pose_deltas(sg.pose, so.pose)
    flag = int(ro["flag"][k])
    wg, wo = w.with_state(sg), w.with_state(so)
    pg, po = s.marginalize(wo, flag), o.marginalize(wo, flag)
    Hg, Ho = pg.J0().T @ pg.J0(), po.J0().T @ po.J0()
    bg, bo = pg.J0().T @ pg.r0(), po.J0().T @ po.r0()
    lam, Vv = np.linalg.eigh(Ho); cdiff = Vv.T @ (bg - bo); top = np.argsort(-np.abs(cdiff))[:3]
    print("      b diff by eigen-direction of H: " + ", ".join("lam %.2e c %.2e (v.b %.2e)" % (lam[t], cdiff[t], (Vv.T @ bo)[t]) for t in top) + "  |bo|max %.3g" % np.abs(bo).max())
    print("%2d n_prior %2d it %2d/%2d acc %s|%s cost %.9g|%.9g dP %.2e dq %.2e | marg n %d/%d H %.2e b %.2e r0r0 %.9g|%.9g" % (
        k, w.prior.n if w.prior is not None else 0, rg.num_iterations, rr.num_iterations, "".join(str(int(a)) for a in rg.accepted[:rg.num_iterations + 1]),
        "".join(str(int(a)) for a in rr.accepted[:rr.num_iterations + 1]), rg.final_cost, rr.final_cost, dp, dq, pg.n, po.n,
        np.abs(Hg - Ho).max() / np.abs(Ho).max(), np.abs(bg - bo).max() / max(1e-300, np.abs(bo).max()), pg.r0() @ pg.r0(), po.r0() @ po.r0()))
