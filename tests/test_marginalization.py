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
