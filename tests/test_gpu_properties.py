"""Size-independent properties of the HIP solve path (MI355X, through the C ABI), checked WITHOUT the oracle: the gauge of a visual-inertial window
(a global translation: whole solves; a rotation about gravity: residuals and start cost), the order of the landmarks, and the fixed point of a converged solve.  They hold for the problem, so they
must hold for every form of the solver at every size -- the canonical window and the 20 000-point window of BASELINE configs[3] alike.

"""
import numpy as np
import pytest

from helpers import uvs, abi, synth, pose_deltas

pytestmark = pytest.mark.gpu
FORMS = ("persistent", "fused")


def _solve(s, w, form):
    if form == "fused":
        st, rep, _ = s.large_solve_fused(w)
        return st, rep
    return s.solve(w)


def _Rz(deg):
    a = np.radians(deg)
    return np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])


def _orth_to_line(o):
    a, b, c, phi = o
    Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    Ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    Rzc = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    U = Rx @ Ry @ Rzc
    nhat, d = U[:, 0], U[:, 1]
    nn = np.cos(phi) / np.sin(phi)
    return np.cross(d, nhat) * nn, d      # the point of the line closest to the origin, its direction


def _move_world(w, R, t):
    """The same window expressed in a world frame moved by (R about gravity, t): poses, velocities, world lines and the prior's linearization point / Jacobian."""
    o = w.copy()
    o.pose = w.pose.copy(); o.speedbias = w.speedbias.copy(); o.line_orth = w.line_orth.copy()
    for i in range(abi.NUM_FRAMES):
        o.pose[i, :3] = R @ w.pose[i, :3] + t
        o.pose[i, 3:] = synth.R_to_quat(R @ synth.quat_to_R(w.pose[i, 3:]))
        o.speedbias[i, :3] = R @ w.speedbias[i, :3]
    for k in range(len(w.line_orth)):
        A, d = _orth_to_line(w.line_orth[k])
        o.line_orth[k] = synth.line_to_orth(R @ A + t, R @ d)
    if w.prior is not None:
        p = w.prior.copy(); n = p.n
        J = p.J0(); x0 = np.ctypeslib.as_array(p.x0)
        for b in range(p.n_blocks):
            kind, c0, xo = p.block_kind[b], p.block_idx[b], p.x0_off[b]
            if kind == abi.BLOCK_POSE:
                x0[xo:xo + 3] = R @ x0[xo:xo + 3] + t
                x0[xo + 3:xo + 7] = synth.R_to_quat(R @ synth.quat_to_R(x0[xo + 3:xo + 7].copy()))
                J[:, c0:c0 + 3] = J[:, c0:c0 + 3] @ R.T
            elif kind == abi.BLOCK_SPEEDBIAS:
                x0[xo:xo + 3] = R @ x0[xo:xo + 3]
                J[:, c0:c0 + 3] = J[:, c0:c0 + 3] @ R.T
        np.ctypeslib.as_array(p.linearized_jacobians)[: n * n] = J.reshape(-1)
        o.prior = p
    return o


def _assert_same_solve(ra, rb, sa, sb_moved_back, cost_rtol, pose_tol):
    assert ra.status == 0 and rb.status == 0 and ra.num_iterations == rb.num_iterations
    assert list(ra.accepted[: ra.num_iterations + 1]) == list(rb.accepted[: rb.num_iterations + 1])
    assert abs(ra.initial_cost - rb.initial_cost) <= cost_rtol * ra.initial_cost and abs(ra.final_cost - rb.final_cost) <= cost_rtol * ra.final_cost
    dp, da = pose_deltas(sa.pose, sb_moved_back.pose)
    assert dp < pose_tol and da < pose_tol, (dp, da)
    assert np.abs(sa.inv_depth - sb_moved_back.inv_depth).max() < 1e-6


@pytest.mark.parametrize("form", FORMS)
def test_translation_of_the_world_frame(gpu_api, form):
    """Points + IMU + prior: positions enter every factor as differences and their Jacobians do not move, so the shifted world takes the SAME LM path to the
    shifted solution.  (Not so with lines: their orthonormal parameters hold the distance from the ORIGIN, so a shift changes the parameterization, the
    diagonal of J^T J and with it the damping -- their residuals are covered below.)"""
    s = gpu_api.Solver(max_batch=2)
    w = synth.make_window(301, n_lines=0, n_tagged=0, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
    t = np.array([13.0, -7.5, 2.25])
    wm = _move_world(w, np.eye(3), t)
    sa, ra = _solve(s, w, form); sb, rb = _solve(s, wm, form)
    s.close()
    back = sb; back.pose = sb.pose.copy(); back.pose[:, :3] -= t
    _assert_same_solve(ra, rb, sa, back, 1e-9, 1e-7)
    assert np.abs(sa.speedbias - sb.speedbias).max() < 1e-7


def test_rotation_about_gravity_leaves_every_residual_alone(gpu_api):
    """Every residual block -- IMU, point, line, vanishing point, prior -- is a function of relative quantities and of gravity's direction: in a world frame
    rotated about gravity and shifted they keep their values (uvs_evaluate, element by element) and the solve starts from the same cost.  (The LM PATH is
    not covariant under the rotation: Ceres' Levenberg-Marquardt damps with diag(J^T J), which is invariant to a rescaling of a coordinate, not to a rotation of
    x / y; so only the translation is checked through whole solves, below.)"""
    s = gpu_api.Solver(max_batch=2)
    w = synth.make_window(302, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
    R, t = _Rz(57.0), np.array([-3.0, 4.0, 0.5])
    wm = _move_world(w, R, t)
    ea, eb = s.evaluate(w, robust=False), s.evaluate(wm, robust=False)
    for name, tol in (("pt_r", 1e-9), ("ln_r", 1e-8), ("vp_r", 1e-8), ("imu_r", 1e-7), ("prior_r", 1e-7)):
        a, b = getattr(ea, name), getattr(eb, name)
        assert a.shape == b.shape and np.abs(a - b).max() <= tol * max(1.0, np.abs(a).max()), (name, np.abs(a - b).max())
    assert abs(ea.cost - eb.cost) <= 1e-10 * ea.cost
    for form in FORMS:
        ra, rb = _solve(s, w, form)[1], _solve(s, wm, form)[1]
        assert abs(ra.initial_cost - rb.initial_cost) <= 1e-10 * ra.initial_cost and rb.final_cost < 1e-6 * rb.initial_cost
    s.close()


def test_gauge_at_configs3_size(gpu_api):
    """BASELINE configs[3] sizes through the fused multi-workgroup loop: 20 000 points / 100 000 observations in a shifted world take the same LM path to the
    shifted solution; the full window (+ 5 000 lines / 35 000 observations) in a world shifted and rotated about gravity starts from the same cost."""
    s = gpu_api.Solver(max_batch=1, max_points=20008, max_point_obs=240000, max_lines=5008, max_line_obs=60000)
    t = np.array([40.0, 25.0, -3.0])
    w = synth.make_window(303, n_points=20000, n_lines=0, n_tagged=0)
    wm = _move_world(w, np.eye(3), t)
    sa, ra, _ = s.large_solve_fused(w); sb, rb, _ = s.large_solve_fused(wm)
    back = sb; back.pose = sb.pose.copy(); back.pose[:, :3] -= t
    _assert_same_solve(ra, rb, sa, back, 1e-8, 1e-6)
    wl = synth.make_window(306, n_points=20000, n_lines=5000, n_tagged=3750)
    ra2 = s.large_solve_fused(wl)[1]; rc2 = s.large_solve_fused(_move_world(wl, _Rz(-121.0), t))[1]
    s.close()
    assert abs(rc2.initial_cost - ra2.initial_cost) <= 1e-10 * ra2.initial_cost and rc2.status == 0


@pytest.mark.parametrize("form", FORMS)
def test_order_of_the_landmarks(gpu_api, form):
    """Renumbering the landmarks (points and lines, with their observation groups) changes the order of every sum over landmarks and nothing else."""
    s = gpu_api.Solver(max_batch=2)
    w = synth.make_window(304, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
    rng = np.random.default_rng(5)
    o = w.copy()
    def regroup(lm, n, arrays):
        perm = rng.permutation(n)                       # new landmark k = old landmark perm[k]
        order = np.concatenate([np.nonzero(lm == perm[k])[0] for k in range(n)])
        new_lm = np.concatenate([np.full(int(np.sum(lm == perm[k])), k, np.int32) for k in range(n)])
        return perm, new_lm, [a[order].copy() for a in arrays]
    pp, o.pt_lm, (o.pt_fi, o.pt_fj, o.pt_pi, o.pt_pj) = regroup(w.pt_lm, len(w.inv_depth), [w.pt_fi, w.pt_fj, w.pt_pi, w.pt_pj])
    o.inv_depth = w.inv_depth[pp].copy()
    lp, o.ln_lm, (o.ln_fj, o.ln_has_vp, o.ln_sp, o.ln_ep, o.ln_vp) = regroup(w.ln_lm, len(w.line_orth), [w.ln_fj, w.ln_has_vp, w.ln_sp, w.ln_ep, w.ln_vp])
    o.line_orth = w.line_orth[lp].copy()
    sa, ra = _solve(s, w, form); sb, rb = _solve(s, o, form)
    s.close()
    assert ra.num_iterations == rb.num_iterations and list(ra.accepted[:11]) == list(rb.accepted[:11])
    assert abs(ra.final_cost - rb.final_cost) <= 1e-10 * ra.final_cost
    dp, da = pose_deltas(sa.pose, sb.pose)
    assert dp < 1e-9 and da < 1e-8
    assert np.abs(sa.inv_depth[pp] - sb.inv_depth).max() < 1e-9 and np.abs(sa.line_orth[lp] - sb.line_orth).max() < 1e-7


@pytest.mark.parametrize("form", FORMS)
def test_the_returned_state_carries_the_reported_cost(gpu_api, form):
    """A solve started from the state another solve returned begins at exactly the cost that solve reported as final (state output and report describe the
    same point), never ends above it, and a run that stopped by the function tolerance is followed by one that stops by a tolerance as well."""
    o = abi.default_options(); o.max_num_iterations = 60; o.function_tolerance = 1e-4
    s = gpu_api.Solver(opts=o, max_batch=2)
    w = synth.make_window(305, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
    s1, r1 = _solve(s, w, form)
    assert r1.termination == 3 and r1.num_iterations < 60, (r1.termination, r1.num_iterations)          # FUNCTION_TOL
    s2, r2 = _solve(s, w.with_state(s1), form)
    s.close()
    assert abs(r2.initial_cost - r1.final_cost) <= 1e-11 * r1.final_cost
    assert r2.final_cost <= r1.final_cost * (1 + 1e-12) and r2.termination in (1, 2, 3)
    costs = np.array(r2.cost[: r2.num_iterations + 1])
    assert np.all(np.diff(costs) <= 1e-12 * costs[0])          # the trace of accepted costs never rises
