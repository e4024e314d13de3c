"""A window of the canonical size through BOTH single-window forms of the solver: the persistent one-workgroup kernel (uvs_solve_window) and
the multi-workgroup fused loop (uvs_large_solve_fused: landmark chunks on many compute units, reduced solve in one workgroup, trust-region
control on the device).  Same LM controller, different summation orders: identical accept / reject traces, costs and states to rounding."""
import numpy as np
import pytest

from helpers import uvs, abi, synth, pose_deltas

pytestmark = pytest.mark.gpu


def _same_solve(s, w, cost_rtol=1e-9, pose_tol=1e-8):
    sk, rk = s.solve(w)
    sf, rf, loop_ms = s.large_solve_fused(w)
    assert loop_ms > 0.0
    assert rf.num_iterations == rk.num_iterations and rf.termination == rk.termination and rf.num_successful == rk.num_successful
    n = rk.num_iterations + 1
    assert list(rf.accepted[:n]) == list(rk.accepted[:n])
    assert np.allclose(list(rf.cost[:n]), list(rk.cost[:n]), rtol=cost_rtol, atol=0.0)
    assert np.allclose(list(rf.radius[:n]), list(rk.radius[:n]), rtol=1e-6, atol=0.0)
    assert abs(rf.final_cost - rk.final_cost) <= cost_rtol * rk.final_cost
    dp, dr = pose_deltas(sf.pose, sk.pose)
    assert dp < pose_tol and dr < max(pose_tol, 1e-7)      # the angle comes out of an arccos near 1: 3e-8 is its resolution
    assert np.abs(sf.speedbias - sk.speedbias).max() < 1e-7
    assert np.abs(sf.inv_depth - sk.inv_depth).max() < 1e-7 and np.abs(sf.line_orth - sk.line_orth).max() < 1e-7
    return rk


@pytest.mark.parametrize("index", [0, 1, 2, 3])
def test_canonical_window_both_forms(gpu_api, index):
    s = gpu_api.Solver(max_batch=1)
    rep = _same_solve(s, synth.make_window(index))
    assert rep.num_iterations >= 3
    s.close()


def test_with_the_products_own_prior(gpu_api):
    s = gpu_api.Solver(max_batch=1)
    w = synth.make_window(7, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
    assert w.prior is not None and w.prior.n > 0
    _same_solve(s, w)
    s.close()


def test_rejected_steps_and_early_termination(gpu_api):
    """Heavy perturbation (rejected steps on the way) and a nearly converged start (terminates on a tolerance before max_num_iterations)."""
    s = gpu_api.Solver(max_batch=1)
    w = synth.make_window(11).copy(); rng = np.random.default_rng(5)
    w.pose[3:, :3] += 0.4 * rng.standard_normal((8, 3)); w.inv_depth *= np.exp(0.5 * rng.standard_normal(len(w.inv_depth)))
    _same_solve(s, w, cost_rtol=1e-7, pose_tol=1e-6)
    o = abi.default_options(); o.max_num_iterations = 40; o.function_tolerance = 5e-3
    s2 = gpu_api.Solver(opts=o, max_batch=1)
    rk = _same_solve(s2, synth.make_window(12), cost_rtol=1e-7, pose_tol=1e-6)
    assert rk.num_iterations < 40 and rk.termination == abi.TERM_NAMES.index("FUNCTION_TOL")
    o = abi.default_options(); o.max_num_iterations = 0
    s3 = gpu_api.Solver(opts=o, max_batch=1)
    assert _same_solve(s3, synth.make_window(12)).num_iterations == 0
    s3.close()
    s.close(); s2.close()


def test_time_offset_and_free_extrinsic(gpu_api):
    o = abi.default_options(); o.estimate_td = 1
    s = gpu_api.Solver(opts=o, max_batch=1)
    _same_solve(s, synth.add_time_offset(synth.make_window(81), td_true=0.006))
    s.close()
    o = abi.default_options(); o.estimate_extrinsic = 1
    s = gpu_api.Solver(opts=o, max_batch=1)
    w = synth.make_window(98).copy(); w.ex_pose[:3] += [0.01, -0.008, 0.005]
    _same_solve(s, w, cost_rtol=1e-7, pose_tol=1e-6)
    s.close()

def test_repeated_calls_alternating_forms_and_batch_sizes(gpu_api):
    """The two forms share the handle's staging buffers (pinned upload / download, device blobs): interleaved calls must not disturb each other."""
    s = gpu_api.Solver(max_batch=8)
    ws = [synth.make_window(20 + i) for i in range(8)]
    ref = [s.solve(w)[1].final_cost for w in ws]
    s.upload(ws); s.solve_resident(); _, reps = s.download()
    assert [r.final_cost for r in reps] == ref
    for i in (3, 0, 7):
        _, rf, _ = s.large_solve_fused(ws[i])
        assert abs(rf.final_cost - ref[i]) <= 1e-9 * ref[i]
        assert s.solve(ws[i])[1].final_cost == ref[i]
    s.upload(ws[:3]); s.solve_resident(); _, reps = s.download()
    assert [r.final_cost for r in reps] == ref[:3]
    s.close()


@pytest.mark.parametrize("form", ["persistent", "fused"])
def test_redamping_equals_relinearizing(gpu_api, form):
    """After a rejected step the solver re-damps the stored linearization (Schur complement and reduced gradient updated from the stored E rows)
    instead of linearizing again; UVS_REDAMP=0 switches that off.  Same traces, costs and states on windows whose traces contain rejections."""
    import os
    s = gpu_api.Solver(max_batch=1)
    rejected = 0
    for index in (3, 5, 11, 14):
        w = synth.make_window(index, with_prior=(index % 2 == 1), marginalize_fn=(lambda win, flag: s.marginalize(win, flag)) if index % 2 == 1 else None)
        out = []
        for flag in (None, "0"):
            if flag is None: os.environ.pop("UVS_REDAMP", None)
            else: os.environ["UVS_REDAMP"] = flag
            try:
                if form == "fused": st, rep, _ = s.large_solve_fused(w)
                else: st, rep = s.solve(w)
            finally:
                os.environ.pop("UVS_REDAMP", None)
            out.append((st, rep))
        (sa, ra), (sb, rb) = out
        n = ra.num_iterations + 1
        assert ra.num_iterations == rb.num_iterations and list(ra.accepted[:n]) == list(rb.accepted[:n]) and ra.termination == rb.termination
        rejected += sum(1 for a in list(ra.accepted[1:n]) if a != 1)
        assert np.allclose(list(ra.cost[:n]), list(rb.cost[:n]), rtol=1e-10, atol=0.0)
        assert np.allclose(list(ra.model_cost_change[:n]), list(rb.model_cost_change[:n]), rtol=1e-7, atol=1e-12)
        dp, dr = pose_deltas(sa.pose, sb.pose)
        assert dp < 1e-9 and dr < 1e-7
        assert np.abs(sa.inv_depth - sb.inv_depth).max() < 1e-9 and np.abs(sa.line_orth - sb.line_orth).max() < 1e-7
    assert rejected >= 4      # the comparison is only worth something if steps WERE rejected
    s.close()
