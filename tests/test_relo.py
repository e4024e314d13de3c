"""Relocalization residual blocks (estimator.cpp:944-978): ProjectionFactor(pts_i, pts_j) between a landmark's start frame and the free
7-dof block relo_Pose, one per matched feature, CauchyLoss(1.0).  On the HIP path relo_Pose is pseudo frame 12: an ordinary second frame
of a point observation whose six dofs sit in the spare 16th slots of frames 0..5 of the padded reduced system (only with a fixed
extrinsic and no td); the oracle appends a 6-dof block after the frames.  Both are exact solves of the same system."""
import numpy as np
import pytest

from helpers import uvs, abi, synth, pose_deltas, quat_angle


def _rel(pa, pb):
    Ra = synth.quat_to_R(pa[3:]); Rb = synth.quat_to_R(pb[3:])
    return Ra.T @ (pb[:3] - pa[:3]), Ra.T @ Rb


def _relo_window(index, **kw):
    relo = {k: kw.pop(k) for k in ("relo_frame", "fraction", "offset", "pixel_sigma") if k in kw}
    return synth.add_relocalization(synth.make_window(index, **kw), seed=index, **relo)


# ---------------------------------------------------------------- oracle (CPU)
def test_oracle_known_answer(oracle):
    """Noise-free points + IMU + relocalization blocks: relo_Pose returns to the pose that generated the match points (relative to the
    window, the gauge is free)."""
    w = _relo_window(2, n_lines=0, n_tagged=0, noise=False, perturb=True, relo_frame=5)
    assert len(w.relo_lm) > 20
    o = abi.default_options(); o.max_num_iterations = 40
    st, rep = oracle.solve(w, o)
    assert rep.final_cost < 1e-12 * rep.initial_cost and rep.final_cost < 1e-10
    t = w.truth
    pe, Re = _rel(st.pose[5], st.relo_pose); pt, Rt = _rel(t["pose"][5], t["relo_pose"])
    assert np.abs(pe - pt).max() < 1e-7 and np.abs(Re - Rt).max() < 1e-8
    assert np.abs(st.relo_pose - w.relo_pose).max() > 0.05       # it did move (0.25 m / 4 deg away from Pose[5])


def test_oracle_schur_equals_dense_with_relo(oracle):
    w = _relo_window(3)
    s0, r0 = oracle.solve(w, linear_mode=0)
    s1, r1 = oracle.solve(w, linear_mode=1)
    assert r0.num_iterations == r1.num_iterations
    assert np.abs(s0.relo_pose - s1.relo_pose).max() < 1e-9 and pose_deltas(s0.pose, s1.pose)[0] < 1e-9
    assert abs(r0.final_cost - r1.final_cost) <= 1e-9 * r1.final_cost


def test_oracle_without_blocks_leaves_relo_pose(oracle):
    w = synth.make_window(4); w.relo_pose = np.array([1.0, 2.0, 3.0, 0.0, 0.0, 0.0, 1.0])
    st, rep = oracle.solve(w)
    assert np.array_equal(st.relo_pose, w.relo_pose)


def test_window_file_round_trip(tmp_path):
    w = _relo_window(5)
    p = str(tmp_path / "w.bin"); w.save(p)
    v = abi.Window.load(p)
    assert np.array_equal(v.relo_lm, w.relo_lm) and np.array_equal(v.relo_pi, w.relo_pi) and np.array_equal(v.relo_pj, w.relo_pj)
    assert np.array_equal(v.relo_pose, w.relo_pose) and np.array_equal(v.pt_pj, w.pt_pj)


# ---------------------------------------------------------------- HIP vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("index,kw", [(110, {}), (111, dict(relo_frame=8, fraction=1.0)), (112, dict(relo_frame=0, pixel_sigma=0.5)),
                                      (113, dict(with_prior=False, n_points=60, n_lines=10, n_tagged=5))])
def test_relo_solve_matches_oracle(gpu_api, oracle, index, kw):
    w = _relo_window(index, **kw)
    assert len(w.relo_lm) > 0
    s = gpu_api.Solver(max_batch=2)
    sg, rg = s.solve(w)
    s.close()
    so, ro = oracle.solve(w)
    assert rg.status == 0 and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    assert abs(rg.initial_cost - ro.initial_cost) <= 1e-10 * ro.initial_cost
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-4 and da < 1e-4, (dp, da)                     # north-star tolerance; observed far below
    assert np.abs(sg.relo_pose[:3] - so.relo_pose[:3]).max() < 1e-6 and quat_angle(sg.relo_pose[3:], so.relo_pose[3:]) < 1e-6
    assert not np.array_equal(sg.relo_pose, w.relo_pose)
    assert np.abs(sg.inv_depth - so.inv_depth).max() < 1e-6
    assert abs(rg.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost


@pytest.mark.gpu
def test_relo_first_iteration_step(gpu_api, oracle):
    w = _relo_window(114)
    o = abi.default_options(); o.max_num_iterations = 1
    s = gpu_api.Solver(opts=o, max_batch=2)
    sg, rg = s.solve(w)
    s.close()
    so, ro = oracle.solve(w, opts=o)
    assert abs(rg.model_cost_change[1] - ro.model_cost_change[1]) <= 1e-8 * abs(ro.model_cost_change[1])
    assert abs(rg.candidate_cost[1] - ro.candidate_cost[1]) <= 1e-8 * abs(ro.candidate_cost[1])
    assert abs(rg.step_norm[1] - ro.step_norm[1]) <= 1e-8 * ro.step_norm[1]
    assert abs(rg.gradient_max_norm[0] - ro.gradient_max_norm[0]) <= 1e-9 * ro.gradient_max_norm[0]
    assert np.abs(sg.relo_pose - so.relo_pose).max() < 1e-9


@pytest.mark.gpu
def test_relo_known_answer_on_gpu(gpu_api):
    w = _relo_window(2, n_lines=0, n_tagged=0, noise=False, perturb=True, relo_frame=5)
    o = abi.default_options(); o.max_num_iterations = 40
    s = gpu_api.Solver(opts=o, max_batch=2)
    st, rep = s.solve(w)
    s.close()
    assert rep.final_cost < 1e-12 * rep.initial_cost
    t = w.truth
    pe, Re = _rel(st.pose[5], st.relo_pose); pt, Rt = _rel(t["pose"][5], t["relo_pose"])
    assert np.abs(pe - pt).max() < 1e-7 and np.abs(Re - Rt).max() < 1e-8


@pytest.mark.gpu
def test_relo_in_a_batch_and_ignored_by_marginalization(gpu_api, oracle):
    """A batch mixing windows with and without relocalization blocks equals the single solves bit for bit; uvs_evaluate /
    uvs_marginalize keep the caller's observation numbering and leave the blocks out, as the reference's marginalization does."""
    ws = [synth.make_window(120), _relo_window(121), _relo_window(122, relo_frame=2), synth.make_window(123)]
    s = gpu_api.Solver(max_batch=4)
    s.upload(ws); s.solve_resident()
    states, reps = s.download()
    for w, sb, rb in zip(ws, states, reps):
        s1, r1 = s.solve(w)
        assert np.array_equal(sb.pose, s1.pose) and np.array_equal(sb.relo_pose, s1.relo_pose) and rb.final_cost == r1.final_cost
    w = ws[1]
    plain = w.copy(); plain.relo_lm = np.zeros(0, np.int32); plain.relo_pi = np.zeros((0, 3)); plain.relo_pj = np.zeros((0, 3))
    ev, ev0 = s.evaluate(w), s.evaluate(plain)
    assert np.array_equal(ev.pt_r, ev0.pt_r) and np.array_equal(ev.pt_J, ev0.pt_J) and ev.cost == ev0.cost
    pg, p0 = s.marginalize(w, 0), s.marginalize(plain, 0)
    assert np.array_equal(pg.J0(), p0.J0()) and np.array_equal(pg.r0(), p0.r0())
    sg, rg = s.solve(w)
    pr = s.marginalize(w.with_state(sg), 0, resident=True)        # resident blob = the one the solve uploaded (with the blocks merged in)
    pn = s.marginalize(w.with_state(sg), 0)
    assert np.array_equal(pr.J0(), pn.J0())
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["persistent", "fused", "stepwise"])
@pytest.mark.parametrize("index,kw", [(130, {}), (131, dict(relo_frame=9, fraction=1.0)), (132, dict(relo_frame=0, pixel_sigma=0.5, with_prior=False))])
def test_relo_blocks_with_estimate_td(gpu_api, oracle, form, index, kw):
    """ESTIMATE_TD and relocalization blocks in one window (estimator.cpp:784-797 + :944-978 coexist in the reference): the point blocks are
    ProjectionTdFactors, the relocalization blocks stay plain ProjectionFactors (no dependence on td), para_Td has its own slot of the reduced
    system and relo_Pose the six a free extrinsic would take.  Same LM trace and states as the oracle's dense solve, in all three forms."""
    w = synth.add_time_offset(_relo_window(index, **kw))
    assert len(w.relo_lm) > 0
    o = abi.default_options(); o.estimate_td = 1
    s = gpu_api.Solver(opts=o, max_batch=2)
    if form == "persistent": sg, rg = s.solve(w)
    elif form == "fused": sg, rg, _ = s.large_solve_fused(w)
    else: sg, rg = s.large_solve(w)
    s.close()
    so, ro = oracle.solve(w, opts=o)
    assert rg.status == 0 and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    assert abs(rg.initial_cost - ro.initial_cost) <= 1e-10 * ro.initial_cost
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-6 and da < 1e-6, (dp, da)
    assert abs(sg.td - so.td) < 1e-8 and sg.td != w.td
    assert np.abs(sg.relo_pose[:3] - so.relo_pose[:3]).max() < 1e-6 and quat_angle(sg.relo_pose[3:], so.relo_pose[3:]) < 1e-6
    assert not np.array_equal(sg.relo_pose, w.relo_pose)
    assert np.abs(sg.inv_depth - so.inv_depth).max() < 1e-6
    assert abs(rg.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost


@pytest.mark.gpu
@pytest.mark.parametrize("td", [0, 1])
@pytest.mark.parametrize("index,kw", [(140, {}), (141, dict(relo_frame=9, fraction=1.0)), (142, dict(relo_frame=0, pixel_sigma=0.5, with_prior=False)), (143, dict(relo_frame=3, fraction=0.2))])
def test_relo_blocks_with_a_free_extrinsic(gpu_api, oracle, index, kw, td):
    """ESTIMATE_EXTRINSIC (and ESTIMATE_TD) together with relocalization blocks: 6 + 6 (+ 1) free dofs do not fit the 11 spare rows of the reduced
    system, so relo_Pose is eliminated at a second level (rank-6 update of S before the factorization, uvs_solve_kernel.h: relo2_eliminate) -- the
    same exact solve of the damped system; same LM trace and states as the oracle's dense solve.  Persistent kernel, batches, and (round 6) the multi-workgroup forms."""
    w = _relo_window(index, **kw)
    if td: w = synth.add_time_offset(w)
    assert len(w.relo_lm) > 0
    o = abi.default_options(); o.estimate_extrinsic = 1; o.estimate_td = td
    s = gpu_api.Solver(opts=o, max_batch=4)
    sg, rg = s.solve(w)
    s.upload([w, w]); s.solve_resident(); st2, rp2 = s.download()      # the batch entry points take it as well
    so, ro = oracle.solve(w, opts=o)
    assert rg.status == 0 and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    assert abs(rg.initial_cost - ro.initial_cost) <= 1e-10 * ro.initial_cost
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-6 and da < 1e-6, (dp, da)
    assert np.abs(sg.ex_pose[:3] - so.ex_pose[:3]).max() < 1e-7 and quat_angle(sg.ex_pose[3:], so.ex_pose[3:]) < 1e-6
    assert not np.array_equal(sg.ex_pose, w.ex_pose)
    if td: assert abs(sg.td - so.td) < 1e-8
    assert np.abs(sg.relo_pose[:3] - so.relo_pose[:3]).max() < 1e-6 and quat_angle(sg.relo_pose[3:], so.relo_pose[3:]) < 1e-6
    assert not np.array_equal(sg.relo_pose, w.relo_pose)
    assert np.abs(sg.inv_depth - so.inv_depth).max() < 1e-6
    assert abs(rg.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost
    assert rp2[1].final_cost == rg.final_cost and np.array_equal(st2[1].relo_pose, sg.relo_pose)
    # the landmark-sharded forms of one rank (round 6): the 14 gather blocks of block row 13 travel behind the canonical partial, k_large_solve eliminates relo_Pose
    # at the second level like k_solve does (uvs_large_kernel.h: LG_R2) -- the fused device-side loop and the step-wise loop, same trace and states
    sf, rf, _ = s.large_solve_fused(w)
    s1, r1 = s.large_solve(w)
    for name, (st, rep) in {"fused": (sf, rf), "step-wise": (s1, r1)}.items():
        assert rep.status == 0 and rep.num_iterations == ro.num_iterations, name
        assert list(rep.accepted[: rep.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1]), name
        dpf, daf = pose_deltas(st.pose, so.pose)
        assert dpf < 1e-6 and daf < 1e-6, (name, dpf, daf)
        assert np.abs(st.ex_pose[:3] - so.ex_pose[:3]).max() < 1e-7 and quat_angle(st.ex_pose[3:], so.ex_pose[3:]) < 1e-6, name
        assert np.abs(st.relo_pose[:3] - so.relo_pose[:3]).max() < 1e-6 and quat_angle(st.relo_pose[3:], so.relo_pose[3:]) < 1e-6, name
        if td: assert abs(st.td - so.td) < 1e-8, name
        assert np.abs(st.inv_depth - so.inv_depth).max() < 1e-6 and abs(rep.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost, name
    s.close()


@pytest.mark.gpu
def test_relo_rejected_combinations(gpu_api):
    w = _relo_window(124)
    s = gpu_api.Solver(max_batch=2)      # (the large-window path takes the blocks since round 3: test_relo_blocks_in_the_multi_workgroup_forms)
    bad = w.copy(); bad.relo_lm = bad.relo_lm[::-1].copy()
    with pytest.raises(RuntimeError, match="uvs error %d" % abi.UVS_ERR_INVALID_ARG):
        s.solve(bad)
    s.close()


@pytest.mark.gpu
def test_host_estimator_relocalization(gpu_api, tmp_path):
    """Estimator::setReloFrame -> optimization() (relocalization blocks through uvs::Problem) -> double2vector() (estimator.cpp:671-691:
    drift correction + relative pose of the loop pair) in the host mirror, against the same window solved through the C ABI and the
    formulas evaluated in numpy."""
    import ctypes as C, os
    from test_host_mirror import _host
    w = _relo_window(130, relo_frame=6)
    s = gpu_api.Solver(max_batch=2)
    st, rep = s.solve(w)
    s.close()
    pin, pout = str(tmp_path / "in.uvsw"), str(tmp_path / "out.bin")
    w.save(pin)
    assert _host().uvs_host_replay_window(pin.encode(), pout.encode(), 0) == 0
    raw = np.fromfile(pout, dtype=np.float64)
    status, iters, c0, c1 = raw[:4]
    assert status == 0 and iters == rep.num_iterations and abs(c0 - rep.initial_cost) <= 1e-9 * c0 and abs(c1 - rep.final_cost) <= 1e-9 * c1
    tail = raw[-28:]
    relo_pose, dr, dt, rel_t, rel_q, rel_yaw, still_set = tail[:7], tail[7:16].reshape(3, 3), tail[16:19], tail[19:22], tail[22:26], tail[26], tail[27]
    assert still_set == 0.0                                         # relocalization_info is consumed (:689)
    assert np.abs(relo_pose - st.relo_pose).max() < 1e-9
    R = synth.quat_to_R
    yaw = lambda M: np.degrees(np.arctan2(M[1, 0], M[0, 0]))
    Rz = lambda deg: np.array([[np.cos(np.radians(deg)), -np.sin(np.radians(deg)), 0], [np.sin(np.radians(deg)), np.cos(np.radians(deg)), 0], [0, 0, 1.0]])
    rot_diff = Rz(yaw(R(w.pose[0, 3:])) - yaw(R(st.pose[0, 3:])))
    k = w.relo_frame
    relo_r = rot_diff @ R(st.relo_pose[3:]); relo_t = rot_diff @ (st.relo_pose[:3] - st.pose[0, :3]) + w.pose[0, :3]
    Pk = rot_diff @ (st.pose[k, :3] - st.pose[0, :3]) + w.pose[0, :3]; Rk = rot_diff @ R(st.pose[k, 3:])
    old_r, old_t = Rz(30.0), np.array([1.0, -2.0, 0.5])
    dr_e = Rz(yaw(old_r) - yaw(relo_r))
    assert np.abs(dr - dr_e).max() < 1e-8 and np.abs(dt - (old_t - dr_e @ relo_t)).max() < 1e-8
    assert np.abs(rel_t - relo_r.T @ (Pk - relo_t)).max() < 1e-8
    assert np.abs(R(rel_q) - relo_r.T @ Rk).max() < 1e-8
    e = yaw(Rk) - yaw(relo_r); e = e - 360.0 * np.floor((e + 180.0) / 360.0)
    assert abs(rel_yaw - e) < 1e-7
    # gauge-free cross-check straight from the raw solver output
    assert np.abs(rel_t - R(st.relo_pose[3:]).T @ (st.pose[k, :3] - st.relo_pose[:3])).max() < 1e-8


@pytest.mark.gpu
def test_host_estimator_relocalization_with_a_free_extrinsic(gpu_api, tmp_path, monkeypatch):
    """Estimator::optimization() with ESTIMATE_EXTRINSIC = 1 and relocalization blocks: the host mirror hands the window to the persistent kernel
    (the multi-workgroup form does not take that combination) and gets the same solve as the C ABI called directly on the window it assembled.
    (Directly = on the DUMPED window: vector2double() rebuilds every quaternion from its rotation matrix, and the line / VP factors differentiate with
    respect to the raw quaternion -- deviation D1 of the reference -- so q and -q are different inputs to the same problem.)"""
    from test_host_mirror import _host
    w = _relo_window(131, relo_frame=4)
    pin, pout = str(tmp_path / "in.uvsw"), str(tmp_path / "out.bin")
    w.save(pin)
    monkeypatch.setenv("UVS_HOST_ESTIMATE_EXTRINSIC", "1")
    monkeypatch.setenv("UVS_DUMP_WINDOWS", str(tmp_path))
    assert _host().uvs_host_replay_window(pin.encode(), pout.encode(), 0) == 0
    raw = np.fromfile(pout, dtype=np.float64)
    status, iters, c0, c1 = raw[:4]
    wd = abi.Window.load(str(tmp_path / "window_0000.bin"))
    assert len(wd.relo_lm) == len(w.relo_lm) > 0
    o = abi.default_options(); o.estimate_extrinsic = 1
    s = gpu_api.Solver(opts=o, max_batch=2)
    st, rep = s.solve(wd)
    s.close()
    assert status == 0 and iters == rep.num_iterations and abs(c0 - rep.initial_cost) <= 1e-9 * c0 and abs(c1 - rep.final_cost) <= 1e-9 * c1
    assert np.abs(raw[-28:-21] - st.relo_pose).max() < 1e-9 and not np.array_equal(st.ex_pose, wd.ex_pose)
    # the prior built after the solve is linearized at the POST-solve extrinsic (estimator.cpp:1004: vector2double() again before the marginalization)
    x0 = np.fromfile(pout + ".x0", dtype=np.float64).reshape(-1, 12)
    exb = x0[x0[:, 0] == abi.UVS_BLOCK_EX_POSE]
    assert len(exb) == 1 and np.abs(exb[0, 3:6] - st.ex_pose[:3]).max() < 1e-9 and np.abs(np.abs(exb[0, 6:10]) - np.abs(st.ex_pose[3:])).max() < 1e-9
    assert np.abs(exb[0, 3:6] - wd.ex_pose[:3]).max() > 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("index,kw", [(21, dict(relo_frame=5)), (22, dict(relo_frame=0, fraction=0.3)), (23, dict(relo_frame=9, with_prior=True))])
def test_relo_blocks_in_the_multi_workgroup_forms(gpu_api, oracle, index, kw):
    """Relocalization blocks through the landmark-sharded path (round 3: relo_Pose travels in the 192-double frame state of the large kernels): the
    step-wise loop and the fused device-side loop -- what the host mirror's Estimator::optimization() calls during a loop closure -- against the
    persistent kernel and the oracle: identical LM traces, relo_Pose and poses to 1e-8."""
    if kw.get("with_prior"):
        kw = dict(kw); kw["marginalize_fn"] = lambda win, flag: oracle.marginalize(win, flag)
    w = _relo_window(index, **kw)
    assert len(w.relo_lm) > 5
    s = gpu_api.Solver(max_batch=2)
    s0, r0 = s.solve(w)
    s1, r1 = s.large_solve(w)
    s.large_comm_init(None)
    s2, r2, _ = s.large_solve_fused(w)
    s.close()
    so, ro = oracle.solve(w)
    n = ro.num_iterations
    for name, (st, rep) in {"step-wise": (s1, r1), "fused": (s2, r2), "persistent": (s0, r0)}.items():
        assert rep.status == 0 and rep.num_iterations == n and list(rep.accepted[:n + 1]) == list(ro.accepted[:n + 1]), name
        assert abs(rep.final_cost - ro.final_cost) <= 1e-7 * ro.final_cost, name
        assert pose_deltas(st.pose, so.pose)[0] < 1e-7 and np.abs(st.relo_pose - so.relo_pose).max() < 1e-7, name
        assert np.abs(st.inv_depth - so.inv_depth).max() < 1e-6, name
    assert np.abs(s2.relo_pose - w.relo_pose).max() > 1e-3      # relo_Pose did move


@pytest.mark.gpu
@pytest.mark.parametrize("td", [0, 1])
def test_relo_blocks_with_a_free_extrinsic_on_a_large_window(gpu_api, oracle, td):
    """The second-level relo_Pose of the landmark-sharded forms (round 6) where its partial sums cross many chunks and persistent workgroups: 1 000 points with
    ten-frame tracks and 300 lines, EVERY point matched in the loop-closure frame, ESTIMATE_EXTRINSIC (and ESTIMATE_TD) -- the 14 gather blocks of block row 13
    accumulate over dozens of chunks into the tail of the partial rows, are summed by k_large_reduce and eliminated in k_large_solve.  Fused loop, step-wise loop
    and the persistent kernel against the oracle: same LM trace, same states."""
    kw = dict(n_points=1000, n_lines=300, n_tagged=200, pt_track=10, ln_track=10)
    w = synth.add_relocalization(synth.make_window(78, **kw), relo_frame=9, fraction=1.0, pixel_sigma=0.5, seed=2)
    if td: w = synth.add_time_offset(w)
    assert len(w.relo_lm) >= 900
    o = abi.default_options(); o.estimate_extrinsic = 1; o.estimate_td = td
    s = gpu_api.Solver(opts=o, max_batch=2, max_points=1100, max_point_obs=12000, max_lines=320, max_line_obs=3400)
    so, ro = oracle.solve(w, opts=o)
    sf, rf, _ = s.large_solve_fused(w)
    s1, r1 = s.large_solve(w)
    s0, r0 = s.solve(w)
    s.close()
    for name, (st, rep) in {"fused": (sf, rf), "step-wise": (s1, r1), "persistent": (s0, r0)}.items():
        assert rep.status == 0 and rep.num_iterations == ro.num_iterations, name
        assert list(rep.accepted[: rep.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1]), name
        dp, da = pose_deltas(st.pose, so.pose)
        assert dp < 1e-6 and da < 1e-6, (name, dp, da)
        assert np.abs(st.relo_pose[:3] - so.relo_pose[:3]).max() < 1e-6 and quat_angle(st.relo_pose[3:], so.relo_pose[3:]) < 1e-6, name
        assert np.abs(st.ex_pose[:3] - so.ex_pose[:3]).max() < 1e-7 and quat_angle(st.ex_pose[3:], so.ex_pose[3:]) < 1e-6, name
        assert np.abs(st.inv_depth - so.inv_depth).max() < 1e-6 and abs(rep.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost, name
    assert not np.array_equal(sf.relo_pose, w.relo_pose) and not np.array_equal(sf.ex_pose, w.ex_pose)
