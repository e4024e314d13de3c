"""The DENSE path of the 512-thread persistent kernel (UVS_DENSE_SCHUR=1 at uvs_create; csrc/uvs_layout.h: UVS_DS_*): the landmark Schur complement AND the direct J^T J /
J^T r terms go through v_mfma_f64_16x16x4_f64 -- no gather lists in the blob, no list walk.  It replaces the same Ceres stage as the list walk (the Schur eliminator behind
ceres::Solve, estimator.cpp:982-994) with another summation order, so parity is by the north-star tolerance (pose 1e-4 m / 1e-4 rad, final cost 1e-6) plus the LM trace on
well-conditioned windows, against the ORACLE and against the list-walk instantiation of the same library.  Opt-in: it is parity-green and slower than the list walk on MI355X
(DESIGN.md 5.00000), so the default stays the list walk."""
import importlib, os, sys
import ctypes as C
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
uvs = importlib.import_module("uv-slam_amd")
synth, abi = uvs.synth, uvs.abi
from helpers import pose_deltas      # noqa: E402
pytestmark = pytest.mark.gpu


def _solver(dense, max_batch=8, redamp=None):
    old = {k: os.environ.get(k) for k in ("UVS_DENSE_SCHUR", "UVS_REDAMP")}
    os.environ["UVS_DENSE_SCHUR"] = "1" if dense else "0"
    if redamp is not None: os.environ["UVS_REDAMP"] = redamp
    try:
        return uvs.api.Solver(device=0, max_batch=max_batch)
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def _trace(rep):
    n = rep.num_iterations
    return [int(rep.accepted[i]) for i in range(n + 1)], rep.termination


def _windows(oracle):
    marg = lambda win, flag: oracle.marginalize(win, flag)
    ws = [synth.make_window(i, with_prior=True, marginalize_fn=marg) for i in range(4)]                      # the benchmark workload (n = 75 prior)
    ws += [synth.make_window(20 + i) for i in range(2)]                                                      # no prior
    rng = np.random.default_rng(17)
    ws += [synth.make_window(5200 + i, n_points=int(rng.integers(20, 320)), n_lines=int(rng.integers(0, 70)), n_tagged=0, pt_track=int(rng.integers(2, 10)),
                             ln_track=int(rng.integers(3, 10))) for i in range(6)]                          # ragged shapes: chunk counts 1..n, short tracks
    ws.append(synth.make_window(5300, n_points=150, n_lines=0, n_tagged=0))                                  # no line chunk
    ws.append(synth.make_window(5301, n_points=12, n_lines=3, n_tagged=1))                                   # chunks far smaller than a wave
    ws.append(synth.make_window(5302, n_points=0, n_lines=30, n_tagged=20))                                  # no point chunk
    return ws


def test_dense_path_matches_oracle_and_the_list_walk(oracle):
    ws = _windows(oracle)
    sd = _solver(True, max_batch=len(ws)); sl = _solver(False, max_batch=len(ws))
    sd.upload(ws); sd.solve_resident(); std, repd = sd.download()
    sl.upload(ws); sl.solve_resident(); stl, repl = sl.download()
    for i, w in enumerate(ws):
        so, ro = oracle.solve(w)
        assert repd[i].status == 0 and _trace(repd[i]) == _trace(ro) == _trace(repl[i]), i
        dp, dq = pose_deltas(std[i].pose, so.pose)
        assert dp < 1e-6 and dq < 1e-6, (i, dp, dq)                                                          # (north star: 1e-4 / 1e-4)
        assert abs(repd[i].final_cost - ro.final_cost) <= 1e-6 * ro.final_cost, i
        assert np.abs(std[i].inv_depth - so.inv_depth).max(initial=0.0) < 1e-6 and np.abs(std[i].speedbias - so.speedbias).max() < 1e-6
        dp2, dq2 = pose_deltas(std[i].pose, stl[i].pose)
        assert dp2 < 1e-7 and dq2 < 1e-7 and abs(repd[i].final_cost - repl[i].final_cost) <= 1e-8 * repl[i].final_cost      # another summation order of the same system
    sd.close(); sl.close()


def test_dense_first_reduced_system_element_wise(oracle):
    """The damped, landmark-reduced frame system of the FIRST linearization (S, g, diag(J^T J), step) out of the dense path against the list walk's, element by element:
    everything the matrix-core products and the C-buffer hand-over produce."""
    w = synth.make_window(0, with_prior=True, marginalize_fn=lambda win, flag: oracle.marginalize(win, flag))
    out = []
    for dense in (True, False):
        s = _solver(dense, max_batch=1)
        out.append(s.debug_first_iteration(w)); s.close()
    a, b = out
    for key, tol in (("S", 1e-9), ("g", 1e-9), ("hd", 1e-10), ("step", 1e-7)):
        ref = np.abs(np.asarray(b[key])).max()
        assert np.abs(np.asarray(a[key]) - np.asarray(b[key])).max() <= tol * ref, key


def test_dense_redamping_equals_relinearizing(oracle):
    """After a rejected step the dense path takes the Schur complement of the old damping out of the stored C buffer and puts the new one in (redamp_dense_*); UVS_REDAMP=0
    linearizes again instead.  Same traces, costs and states on windows whose traces contain rejections, one and two in a row."""
    rejected = 0
    for index in (3, 5, 11, 14):
        marg = (lambda win, flag: oracle.marginalize(win, flag)) if index % 2 == 1 else None
        w = synth.make_window(index, with_prior=(index % 2 == 1), marginalize_fn=marg)
        out = []
        for flag in (None, "0"):
            s = _solver(True, max_batch=1, redamp=flag)
            out.append(s.solve(w)); s.close()
        (sa, ra), (sb, rb) = out
        n = ra.num_iterations + 1
        assert _trace(ra) == _trace(rb)
        rejected += sum(1 for a in list(ra.accepted[1:n]) if a != 1)
        assert np.allclose(list(ra.cost[:n]), list(rb.cost[:n]), rtol=1e-9, atol=0.0)
        dp, dr = pose_deltas(sa.pose, sb.pose)
        assert dp < 1e-8 and dr < 1e-7
    assert rejected >= 4


def test_dense_batch_is_bitwise_reproducible_and_equals_single(oracle):
    """One writer per C-buffer entry and fixed step order: two launches agree bit for bit, a window in a batch equals the window alone."""
    marg = lambda win, flag: oracle.marginalize(win, flag)
    ws = [synth.make_window(i, with_prior=True, marginalize_fn=marg) for i in range(24)]
    s = _solver(True, max_batch=len(ws))
    s.upload(ws); s.solve_resident(); s1, r1 = s.download()
    s.upload(ws); s.solve_resident(); s2, r2 = s.download()
    for a, b, ra, rb in zip(s1, s2, r1, r2):
        assert ra.status == 0 and np.array_equal(a.pose, b.pose) and np.array_equal(a.inv_depth, b.inv_depth) and ra.final_cost == rb.final_cost
    alone, ralone = s.solve(ws[13])
    assert np.array_equal(alone.pose, s1[13].pose) and ralone.final_cost == r1[13].final_cost
    s.close()


def test_dense_blob_carries_no_gather_lists():
    """Host-visible half of the path: the dense packing is smaller (no lists, no group table) and its chunks leave room for the C buffer at the end of the staging area."""
    lib = uvs.api.lib()
    lib.uvs_debug_pack_layout.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowC), C.POINTER(C.c_int32)]
    w = synth.make_window(1000)
    o = abi.default_options(); wc, keep = w.to_c()
    info = {}
    for dense in ("0", "1"):
        os.environ["UVS_DEBUG_PACK_DENSE"] = dense
        try:
            buf = (C.c_int32 * 12)(); assert lib.uvs_debug_pack_layout(C.byref(o), C.byref(wc), buf) == abi.UVS_OK
        finally:
            os.environ.pop("UVS_DEBUG_PACK_DENSE", None)
        info[dense] = list(buf)
    assert info["1"][0] < info["0"][0] - 16 * 1024          # blob bytes: the gather lists are ~28 KB of a canonical window's blob
    assert info["1"][7] <= info["1"][8] < info["0"][8]      # fullest chunk <= its staging capacity < the list walk's capacity (C buffer reserved)
