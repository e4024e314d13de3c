"""Marginalization prior (a12: marginalization_factor.cpp:174-297) against an EXTENDED-PRECISION restatement.

`helpers.marginalization_reference` assembles the MARGIN_OLD normal equations from an evaluation dump in numpy longdouble and forms
the Schur complement by pivoted Gaussian elimination -- no eigen-decomposition, no float64 round-off worth mentioning.  A_mm is
graded over ten orders of magnitude (pose information 1e7..1e8, inverse-depth information 1e0..1e2) and its SMALL eigenvalues are
the ones that get inverted, so this is the check that tells a sloppy eigen-solver (absolute stopping rule: 1e-5 relative errors in
J0^T r0) from a careful one.  CPU: the oracle.  GPU: the product (block elimination of the landmark blocks + QL for the n x n factor).
"""
import numpy as np
import pytest

from helpers import uvs, abi, synth, marginalization_reference, prior_information


def _check(evaluate, marginalize, w, tol_H, tol_b):
    ev = evaluate(w)
    Ar, br, kp = marginalization_reference(w, ev)
    p = marginalize(w)
    H, b, cols = prior_information(p)
    assert sorted(cols) == sorted(kp) and p.n == len(kp)
    perm = [cols.index(c) for c in kp]
    H, b = H[np.ix_(perm, perm)], b[perm]
    eH, eb = np.abs(H - Ar).max() / np.abs(Ar).max(), np.abs(b - br).max() / np.abs(br).max()
    assert eH < tol_H and eb < tol_b, (eH, eb)
    return eH, eb


@pytest.mark.parametrize("index,with_prior", [(70, False), (71, True), (72, True)])
def test_oracle_prior_matches_extended_precision(oracle, index, with_prior):
    w = synth.make_window(index, with_prior=with_prior, marginalize_fn=lambda win, flag: oracle.marginalize(win, flag))
    st, _ = oracle.solve(w)
    _check(lambda x: oracle.evaluate(x, robust=True), lambda x: oracle.marginalize(x, 0), w.with_state(st), 5e-7, 1e-8)


@pytest.mark.gpu
@pytest.mark.parametrize("index,with_prior", [(70, False), (71, True), (72, True)])
def test_hip_prior_matches_extended_precision(gpu_api, index, with_prior):
    s = gpu_api.Solver(max_batch=2)
    w = synth.make_window(index, with_prior=with_prior, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
    st, _ = s.solve(w)
    eH, eb = _check(lambda x: s.evaluate(x, robust=True), lambda x: s.marginalize(x, 0), w.with_state(st), 5e-7, 1e-8)
    print("HIP prior vs longdouble Schur complement: H %.2e, b %.2e (relative to the largest entry)" % (eH, eb))
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("index", [41, 42, 43])
def test_device_marginalization_equals_host_path(gpu_api, oracle, index):
    """MARGIN_OLD on the device (round 3): the factors of frame 0 as a window of their own, ONE linearization with an infinite radius by the solver's kernels
    (= assembly + elimination of every dropped landmark block), frame-0 elimination and the n x n factorization on the host -- against the host path
    (UVS_MARG_HOST=1: k_evaluate + assembly + block elimination in csrc/uvs_marg.h) and against the oracle, in the information form."""
    import os
    marg = lambda win, flag: oracle.marginalize(win, flag)
    w = synth.make_window(index, with_prior=True, marginalize_fn=marg)
    s = gpu_api.Solver(max_batch=2)
    st, rep = s.solve(w)
    ws = w.with_state(st)
    pd = s.marginalize(ws, 0)
    os.environ["UVS_MARG_HOST"] = "1"
    try: ph = s.marginalize(ws, 0)
    finally: os.environ.pop("UVS_MARG_HOST")
    s.close()
    po = oracle.marginalize(ws, 0)
    assert pd.n == ph.n == po.n and pd.n_blocks == ph.n_blocks and list(pd.block_kind[:pd.n_blocks]) == list(ph.block_kind[:ph.n_blocks]) and list(pd.block_idx[:pd.n_blocks]) == list(ph.block_idx[:ph.n_blocks])
    Ad, Ah, Ao = pd.J0().T @ pd.J0(), ph.J0().T @ ph.J0(), po.J0().T @ po.J0()
    bd, bh, bo = pd.J0().T @ pd.r0(), ph.J0().T @ ph.r0(), po.J0().T @ po.r0()
    sc = np.abs(Ao).max()
    assert np.abs(Ad - Ah).max() <= 1e-7 * sc and np.abs(Ad - Ao).max() <= 1e-6 * sc
    assert np.abs(bd - bh).max() <= 1e-6 * max(1.0, np.abs(bo).max()) and np.abs(bd - bo).max() <= 1e-5 * max(1.0, np.abs(bo).max())
    assert np.array_equal(np.asarray(pd.x0[:80]), np.asarray(ph.x0[:80]))


@pytest.mark.gpu
@pytest.mark.parametrize("flag", [0, 1])
def test_marginalization_in_two_halves_equals_the_one_call_form(gpu_api, flag):
    """uvs_marginalize_resident_begin / uvs_marginalize_wait (ABI v6: the marginalization on a worker thread of the handle, beside the caller's work between
    two frames) delivers bit for bit the prior of uvs_marginalize_resident, for both marginalization kinds; a second begin before the wait, and a wait with
    nothing in flight, are errors and leave the job alone."""
    s = gpu_api.Solver(max_batch=2)
    w = synth.make_window(73, with_prior=True, marginalize_fn=lambda win, f: s.marginalize(win, f))
    st, _ = s.solve(w)
    ws = w.with_state(st)
    ref = s.marginalize(ws, flag, resident=True)
    with pytest.raises(Exception):
        s.marginalize_wait()
    st2, _ = s.solve(w)                                   # the factors of w resident again
    s.marginalize_begin(ws, flag)
    with pytest.raises(Exception):
        s.marginalize_begin(ws, flag)
    p = s.marginalize_wait()
    assert p.n == ref.n and p.n_blocks == ref.n_blocks
    assert np.array_equal(np.asarray(p.J0()), np.asarray(ref.J0())) and np.array_equal(np.asarray(p.r0()), np.asarray(ref.r0()))
    s.close()


@pytest.mark.gpu
def test_batched_marginalization_equals_the_one_window_call(gpu_api, oracle):
    """uvs_marginalize_batch (ABI v7, round 6): both marginalization kinds, windows with and without a prior, one window without any factor of frame 0 -- the frame-block
    elimination, the Schur complement and the eigen-decomposition of ALL windows in one launch (csrc/uvs_marg_kernel.h: parallel cyclic Jacobi) against the one-window
    call (host finish: Cholesky + tridiagonal QL) and against the oracle, in the information form the next solve reads (H = J0^T J0, b = J0^T r0), plus the block tables
    and linearization points bit for bit; and against the extended-precision Schur complement of test_hip_prior_matches_extended_precision."""
    s = gpu_api.Solver(max_batch=2)
    marg = lambda win, flag: s.marginalize(win, flag)
    wins, flags = [], []
    for k, (index, with_prior) in enumerate([(70, False), (71, True), (72, True), (41, True), (42, True), (43, True), (73, True), (74, True)]):
        w = synth.make_window(index, with_prior=with_prior, marginalize_fn=marg)
        st, _ = s.solve(w)
        ws_ = w.with_state(st)
        wins.append(ws_); flags.append(0)
        if with_prior: wins.append(ws_); flags.append(1)
    single = [s.marginalize(w, f) for w, f in zip(wins, flags)]
    batch, status = s.marginalize_batch(wins, flags)
    assert status == [0] * len(wins)
    worst = [0.0, 0.0, 0.0, 0.0, 0.0]
    for k, (w, f, p1, pb) in enumerate(zip(wins, flags, single, batch)):
        assert pb.n == p1.n and pb.n_blocks == p1.n_blocks, (k, f)
        nb = p1.n_blocks
        for fld in ("block_kind", "block_frame", "block_size", "block_idx", "x0_off"):
            assert list(getattr(pb, fld)[:nb]) == list(getattr(p1, fld)[:nb]), (k, fld)
        assert np.array_equal(np.asarray(pb.x0[:9 * nb]), np.asarray(p1.x0[:9 * nb]))
        if p1.n == 0: continue
        po = oracle.marginalize(w, f)
        Hb, H1, Ho = pb.J0().T @ pb.J0(), p1.J0().T @ p1.J0(), po.J0().T @ po.J0()
        bb, b1, bo = pb.J0().T @ pb.r0(), p1.J0().T @ p1.r0(), po.J0().T @ po.r0()
        sc, sb = np.abs(Ho).max(), max(1.0, np.abs(bo).max())
        e = [np.abs(Hb - H1).max() / sc, np.abs(bb - b1).max() / sb, np.abs(Hb - Ho).max() / sc, np.abs(bb - bo).max() / sb]
        assert e[0] <= 1e-7 and e[1] <= 1e-6 and e[2] <= 1e-6 and e[3] <= 1e-5, (k, f, e)
        # the constant of the prior's cost, r0^T r0 = b^T H^+ b, which the termination tests of the next solve see
        # (it weighs the components of b along the SMALLEST kept eigenvalues by their inverses: the tridiagonal QL of the one-window call resolves an eigenvalue to ~1e-16 ||H||
        # absolute, the Jacobi of the batch kernel to ~1e-16 relative, so the two -- and the oracle -- agree to 1e-6 .. 1e-4 here, not to round-off)
        c_b, c_1, c_o = float(pb.r0() @ pb.r0()), float(p1.r0() @ p1.r0()), float(po.r0() @ po.r0())
        e.append(abs(c_b - c_1) / max(1.0, abs(c_o)))
        assert e[4] <= 2e-5 and abs(c_b - c_o) <= 5e-4 * max(1.0, abs(c_o)), (k, c_b, c_1, c_o)      # (one-window call vs oracle on the same windows: up to 6e-5)
        worst = [max(a, b_) for a, b_ in zip(worst, e)]
    # extended precision (MARGIN_OLD of the first three windows)
    for k in (0, 1, 3):
        w = wins[k]
        ev = s.evaluate(w, robust=True)
        Ar, br, kp = marginalization_reference(w, ev)
        H, b, cols = prior_information(batch[k])
        perm = [cols.index(c) for c in kp]
        H, b = H[np.ix_(perm, perm)], b[perm]
        assert np.abs(H - Ar).max() / np.abs(Ar).max() < 5e-7 and np.abs(b - br).max() / np.abs(br).max() < 1e-8
    print("batched marginalization of %d windows vs one-window calls: H %.1e, b %.1e; vs oracle: H %.1e, b %.1e; cost constant %.1e (relative)" % (len(wins), *worst))
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("td,ex", [(1, 0), (0, 1), (1, 1)])
def test_batched_marginalization_with_time_offset_and_free_extrinsic(gpu_api, td, ex):
    """uvs_marginalize_batch under ESTIMATE_TD / ESTIMATE_EXTRINSIC (the time-offset and extrinsic blocks are KEPT blocks of the prior: n = 76 with td), windows with
    relocalization blocks (not marginalized: the sub-window leaves them out) and a window without a prior in the same batch: block tables and x0 bit for bit, information
    form to rounding against the one-window call."""
    o = abi.default_options(); o.estimate_td = td; o.estimate_extrinsic = ex
    s = gpu_api.Solver(opts=o, max_batch=2)
    wins, flags = [], []
    for index, with_prior, relo in [(80, True, False), (81, True, True), (82, False, False), (83, True, False)]:
        w = synth.make_window(index)
        if td: w = synth.add_time_offset(w)
        if with_prior:      # a prior that carries the option's blocks: the marginalization of another window under the same options (tests/test_td.py does the same)
            prev = synth.make_window(index + 100)
            if td: prev = synth.add_time_offset(prev)
            stp, _ = s.solve(prev)
            w.prior = s.marginalize(prev.with_state(stp), 0)
        if relo: w = synth.add_relocalization(w, relo_frame=4, fraction=0.5, seed=index)
        st, _ = s.solve(w)
        ws_ = w.with_state(st)
        wins.append(ws_); flags.append(0)
        if with_prior: wins.append(ws_); flags.append(1)
    single = [s.marginalize(w, f) for w, f in zip(wins, flags)]
    batch, status = s.marginalize_batch(wins, flags)
    assert status == [0] * len(wins)
    for k, (p1, pb) in enumerate(zip(single, batch)):
        assert pb.n == p1.n and pb.n_blocks == p1.n_blocks and p1.n >= 30, k
        nb = p1.n_blocks
        for fld in ("block_kind", "block_frame", "block_size", "block_idx", "x0_off"):
            assert list(getattr(pb, fld)[:nb]) == list(getattr(p1, fld)[:nb]), (k, fld)
        assert np.array_equal(np.asarray(pb.x0[:9 * nb]), np.asarray(p1.x0[:9 * nb]))
        if td: assert abi.UVS_BLOCK_TD in list(pb.block_kind[:nb])
        Hb, H1 = pb.J0().T @ pb.J0(), p1.J0().T @ p1.J0()
        bb, b1 = pb.J0().T @ pb.r0(), p1.J0().T @ p1.r0()
        assert np.abs(Hb - H1).max() <= 1e-7 * np.abs(H1).max() and np.abs(bb - b1).max() <= 1e-6 * max(1.0, np.abs(b1).max()), k
    # errors of one window do not take the others down: the per-window status says which one
    bad = wins[0].copy(); bad.pt_lm = bad.pt_lm[::-1].copy()
    with pytest.raises(RuntimeError):
        s.marginalize_batch([wins[1], bad], [0, 0])
    pri, st = s.marginalize_batch([wins[1], bad, wins[0]], [flags[1], 0, 0], check=False)
    assert st[0] == 0 and st[1] == abi.UVS_ERR_INVALID_ARG and st[2] == 0
    assert np.array_equal(np.asarray(pri[0].J0()), np.asarray(batch[1].J0())) and np.array_equal(np.asarray(pri[2].J0()), np.asarray(batch[0].J0()))      # (and the batch is reproducible bit for bit)
    s.close()


@pytest.mark.gpu
def test_batched_marginalization_argument_errors(gpu_api):
    """Misuse of uvs_marginalize_batch is an error code, never a crash: an empty batch is fine, null arrays / a null window / a flag outside {0, 1} are UVS_ERR_INVALID_ARG with a message,
    and a marginalization begun with uvs_marginalize_resident_begin must be waited for first."""
    import ctypes as C
    s = gpu_api.Solver(max_batch=2)
    L = gpu_api.lib(); L.uvs_marginalize_batch.restype = C.c_int
    assert L.uvs_marginalize_batch(s._h, 0, None, None, None, None) == abi.UVS_OK
    w = synth.make_window(75)
    st, _ = s.solve(w); ws_ = w.with_state(st)
    wc, keep = ws_.to_c()
    arr = (C.POINTER(abi.WindowC) * 2)(C.pointer(wc), None)
    pri = (abi.Prior * 2)(); stc = (C.c_int * 2)()
    assert L.uvs_marginalize_batch(s._h, 2, arr, (C.c_int * 2)(0, 0), pri, stc) == abi.UVS_ERR_INVALID_ARG                 # a null window
    arr[1] = C.pointer(wc)
    assert L.uvs_marginalize_batch(s._h, 2, arr, (C.c_int * 2)(0, 2), pri, stc) == abi.UVS_ERR_INVALID_ARG                 # a flag that is neither MARGIN_OLD nor MARGIN_SECOND_NEW
    assert b"flag" in L.uvs_last_error(s._h)
    assert L.uvs_marginalize_batch(s._h, 2, None, (C.c_int * 2)(0, 0), pri, stc) == abi.UVS_ERR_INVALID_ARG
    assert L.uvs_marginalize_batch(s._h, 2, arr, (C.c_int * 2)(0, 0), None, stc) == abi.UVS_ERR_INVALID_ARG
    assert L.uvs_marginalize_batch(None, 2, arr, (C.c_int * 2)(0, 0), pri, stc) == abi.UVS_ERR_INVALID_ARG
    s.solve(w); s.marginalize_begin(ws_, 0)
    assert L.uvs_marginalize_batch(s._h, 2, arr, (C.c_int * 2)(0, 0), pri, stc) == abi.UVS_ERR_INVALID_ARG                 # the worker thread owns the handle until the wait
    s.marginalize_wait()
    assert L.uvs_marginalize_batch(s._h, 2, arr, (C.c_int * 2)(0, 0), pri, None) == abi.UVS_OK and pri[0].n == pri[1].n > 0  # status may be NULL
    assert np.array_equal(np.asarray(pri[0].J0()), np.asarray(pri[1].J0()))
    s.close()
