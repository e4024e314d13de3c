"""Edge cases of the HIP solve path through the C ABI (MI355X): empty landmark families, ragged tracks, skipped IMU blocks, a prior
built by the product's own marginalization, option corner cases, capacity / argument errors, bitwise reproducibility at the BASELINE
batch size.  The oracle is the checker."""
import numpy as np
import pytest

from helpers import uvs, abi, synth, pose_deltas

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver(gpu_api):
    s = gpu_api.Solver(max_batch=256)
    yield s
    s.close()


FORMS = ("persistent", "fused")      # uvs_solve_window (one workgroup per window) / uvs_large_solve_fused (many workgroups, one window)


def _solve(solver, w, form):
    if form == "fused":
        st, rep, _ = solver.large_solve_fused(w)
        return st, rep
    return solver.solve(w)


def _same_solution(sg, rg, so, ro, tol=1e-6):
    assert rg.status == 0 and rg.num_iterations == ro.num_iterations
    assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1])
    dp, da = pose_deltas(sg.pose, so.pose)
    assert dp < 1e-4 and da < 1e-4, (dp, da)              # BASELINE tolerance: 1e-4 m / 1e-4 rad
    assert abs(rg.final_cost - ro.final_cost) <= tol * max(ro.final_cost, 1e-12)


@pytest.mark.parametrize("kw", [dict(n_lines=0, n_tagged=0), dict(n_points=0), dict(n_points=7, n_lines=3, n_tagged=2),
                                dict(n_tagged=0), dict(pt_track=9, ln_track=10), dict(pt_track=2, ln_track=5)])
@pytest.mark.parametrize("form", FORMS)
def test_landmark_family_corner_cases(solver, oracle, kw, form):
    w = synth.make_window(60, **kw)
    sg, rg = _solve(solver, w, form)
    so, ro = oracle.solve(w)
    _same_solution(sg, rg, so, ro)


@pytest.mark.parametrize("form", FORMS)
def test_ragged_tracks(solver, oracle, form):
    """Tracks of different lengths and start frames (the reference's tracks are whatever the front-end delivers)."""
    w = synth.make_window(61)
    rng = np.random.default_rng(5)
    keep_p = np.ones(len(w.pt_lm), bool)
    for k in range(len(w.inv_depth)):                         # drop a random tail of every third point track (keeps >= 1 observation)
        obs = np.nonzero(w.pt_lm == k)[0]
        if k % 3 == 0 and len(obs) > 2:
            keep_p[obs[int(rng.integers(1, len(obs))):]] = False
    for name in ("pt_lm", "pt_fi", "pt_fj", "pt_pi", "pt_pj"):
        setattr(w, name, getattr(w, name)[keep_p])
    sg, rg = _solve(solver, w, form)
    so, ro = oracle.solve(w)
    _same_solution(sg, rg, so, ro)


@pytest.mark.parametrize("form", FORMS)
def test_skipped_imu_blocks(solver, oracle, form):
    """pre_integrations[j]->sum_dt > 10 s => the IMU factor is not added (estimator.cpp:814-815)."""
    w = synth.make_window(62)
    for b in (2, 7):
        w.imu[b]["skip"] = 1
    sg, rg = _solve(solver, w, form)
    so, ro = oracle.solve(w)
    _same_solution(sg, rg, so, ro)


def test_prior_from_the_products_own_marginalization(solver, oracle):
    w = synth.make_window(63, with_prior=True, marginalize_fn=lambda win, flag: solver.marginalize(win, flag))
    assert w.prior is not None and w.prior.n == 75
    sg, rg = solver.solve(w)
    so, ro = oracle.solve(w)
    _same_solution(sg, rg, so, ro)
    # and the next prior agrees with the oracle's (dense assembly + eigen-decompositions, marginalization_factor.cpp:174-297)
    pg = solver.marginalize(w.with_state(sg), 0)
    po = oracle.marginalize(w.with_state(so), 0)
    assert pg.n == po.n
    Hg, Ho = pg.J0().T @ pg.J0(), po.J0().T @ po.J0()          # J0 is unique up to the eigenvector signs; J0^T J0 is not
    assert np.abs(Hg - Ho).max() <= 1e-6 * np.abs(Ho).max()


@pytest.mark.parametrize("form", FORMS)
def test_prior_that_keeps_the_speed_bias_of_a_later_frame(gpu_api, solver, oracle, form, monkeypatch):
    """The reduced-system Cholesky pairs the rows of blocks (i, j), j < i - 1, because only the pose rows of such a block can be non-zero --
    unless a prior couples the speed / bias of a frame >= 2 to an early frame, which the C ABI allows (the reference never builds one:
    marginalization keeps para_SpeedBias[1] only, estimator.cpp:1010-1083).  Such a window takes the full-row path; same oracle parity.
    And an ordinary window gives the same trace through both paths (UVS_CHOL_FULL_ROWS forces the full-row one)."""
    w = synth.make_window(65, with_prior=True, marginalize_fn=lambda win, flag: solver.marginalize(win, flag)).copy()
    p = w.prior.copy()
    moved = [b for b in range(p.n_blocks) if p.block_kind[b] == abi.BLOCK_SPEEDBIAS]
    assert moved
    for b in moved: p.block_frame[b] = 3
    x0 = np.ctypeslib.as_array(p.x0)
    for b in moved: x0[p.x0_off[b]:p.x0_off[b] + 9] = w.speedbias[3]      # linearization point of the moved block: frame 3's state
    w.prior = p
    sg, rg = _solve(solver, w, form)
    so, ro = oracle.solve(w)
    _same_solution(sg, rg, so, ro)
    w2 = synth.make_window(66, with_prior=True, marginalize_fn=lambda win, flag: solver.marginalize(win, flag))
    s_half, r_half = _solve(solver, w2, form)
    monkeypatch.setenv("UVS_CHOL_FULL_ROWS", "1")
    s2 = gpu_api.Solver(max_batch=1)
    s_full, r_full = _solve(s2, w2, form)
    s2.close()
    assert r_full.num_iterations == r_half.num_iterations and list(r_full.accepted[:11]) == list(r_half.accepted[:11])
    assert abs(r_full.final_cost - r_half.final_cost) <= 1e-10 * r_half.final_cost
    assert np.abs(s_full.pose - s_half.pose).max() < 1e-9


@pytest.mark.parametrize("form", FORMS)
def test_zero_and_one_iterations(gpu_api, oracle, form):
    w = synth.make_window(64)
    for n in (0, 1):
        o = abi.default_options(); o.max_num_iterations = n
        s = gpu_api.Solver(opts=o, max_batch=1)
        sg, rg = _solve(s, w, form)
        s.close()
        so, ro = oracle.solve(w, opts=o)
        assert rg.num_iterations == ro.num_iterations == n
        assert abs(rg.final_cost - ro.final_cost) <= 1e-9 * ro.final_cost
        if n == 0:
            assert np.array_equal(sg.pose, w.pose)


def test_errors_are_reported_not_swallowed(gpu_api):
    w = synth.make_window(65)
    s = gpu_api.Solver(max_batch=1, max_points=100)               # 150 points do not fit
    with pytest.raises(Exception) as e:
        s.solve(w)
    assert "capacity" in str(e.value).lower() or "max_points" in str(e.value).lower() or "CAPACITY" in str(e.value)
    s.close()
    s = gpu_api.Solver(max_batch=1)
    bad = w.copy(); bad.pt_fj = bad.pt_fj.copy(); bad.pt_fj[0] = 11      # frame index out of range
    with pytest.raises(Exception):
        s.solve(bad)
    s.close()


def test_baseline_batch_is_bitwise_reproducible(solver, oracle):
    """BASELINE configs[2] size: 256 windows per launch; every sum has a fixed order (no atomics), so two launches agree bit for bit,
    and a window solved inside the batch equals the same window solved alone."""
    # the benchmarked batch: every window carries the n = 75 prior built by the product's own marginalization (bench.py does the same)
    marg = lambda win, flag: solver.marginalize(win, flag)
    ws = [synth.make_window(i, with_prior=True, marginalize_fn=marg) for i in range(256)]      # window indices 0..255: exactly the batch bench.py times on rank 0
    assert all(w.prior is not None and w.prior.n == 75 for w in ws)
    solver.upload(ws); solver.solve_resident(); s1, r1 = solver.download()
    solver.upload(ws); solver.solve_resident(); s2, r2 = solver.download()
    for a, b, ra, rb in zip(s1, s2, r1, r2):
        assert ra.status == 0 and np.array_equal(a.pose, b.pose) and np.array_equal(a.inv_depth, b.inv_depth) and ra.final_cost == rb.final_cost
    alone, ralone = solver.solve(ws[137])
    assert np.array_equal(alone.pose, s1[137].pose) and ralone.final_cost == r1[137].final_cost
    costs = np.array([r.final_cost for r in r1]); init = np.array([r.initial_cost for r in r1])
    assert np.all(costs < 1e-6 * init)                        # every window converges from the perturbed start
    # and the batch equals the oracle on a sample of its windows (LM trace, final cost, poses)
    from helpers import pose_deltas
    for k in (0, 85, 170, 255):
        so, ro = oracle.solve(ws[k])
        assert r1[k].num_iterations == ro.num_iterations and list(r1[k].accepted[:ro.num_iterations + 1]) == list(ro.accepted[:ro.num_iterations + 1])
        dp, dq = pose_deltas(s1[k].pose, so.pose)
        assert dp < 1e-6 and dq < 1e-6 and abs(r1[k].final_cost - ro.final_cost) <= 1e-6 * ro.final_cost


def test_marginalize_resident_equals_marginalize(gpu_api, oracle):
    """uvs_marginalize_resident (state-only upload after a solve of the same window) gives bit-for-bit the prior of uvs_marginalize,
    for both marginalization kinds, and refuses a window that does not match the resident one."""
    s = gpu_api.Solver(max_batch=2)
    w = synth.make_window(64, with_prior=True, marginalize_fn=lambda win, flag: s.marginalize(win, flag))
    st, rep = s.solve(w)                                   # leaves w's factors resident
    post = w.with_state(st)
    for flag in (0, 1):
        pr = s.marginalize(post, flag, resident=True)
        s.solve(w)
        pf = s.marginalize(post, flag)
        assert pr.n == pf.n and np.array_equal(pr.J0(), pf.J0()) and np.array_equal(pr.r0(), pf.r0())
        s.solve(w)
    other = synth.make_window(65, n_points=40, n_lines=10, n_tagged=8)
    with pytest.raises(RuntimeError):
        s.marginalize(other, 0, resident=True)
    s.close()


def test_threaded_batch_packing_equals_serial(gpu_api):
    """uvs_batch_upload packs the windows of a batch on several host threads (UVS_PACK_THREADS, default min(16, cores)) into per-window
    buffers and concatenates them: the device must see the same bytes as from the serial path, i.e. bitwise equal solves -- on a
    heterogeneous batch (different sizes => different blob lengths and chunk counts)."""
    import os
    rng = np.random.default_rng(11)
    ws = [synth.make_window(900 + i, n_points=int(rng.integers(20, 200)), n_lines=int(rng.integers(0, 50)), n_tagged=0) for i in range(24)]
    def run(threads):
        old = os.environ.get("UVS_PACK_THREADS")
        os.environ["UVS_PACK_THREADS"] = str(threads)
        try:
            s = gpu_api.Solver(max_batch=32)
            s.upload(ws); s.solve_resident(); st, rep = s.download(); s.close()
        finally:
            if old is None: os.environ.pop("UVS_PACK_THREADS")
            else: os.environ["UVS_PACK_THREADS"] = old
        return st, rep
    s1, r1 = run(1); s8, r8 = run(8)
    for a, b, ra, rb in zip(s1, s8, r1, r8):
        assert ra.status == 0 and ra.final_cost == rb.final_cost and np.array_equal(a.pose, b.pose) and np.array_equal(a.inv_depth, b.inv_depth) and np.array_equal(a.line_orth, b.line_orth)


def test_batch_stream_equals_batch_by_batch(gpu_api):
    """uvs_batch_stream: eight heterogeneous batches through the pipeline (packing of batch k + 1 on host threads and its H2D copy while the GPU
    runs k_solve -> gather of batch k; three buffer sets take turns and each is reused at least twice; the results are written into pinned host
    memory by the gather kernel) against upload / solve / download of every batch on a fresh handle: bitwise equal states and reports.  Then once
    more on the same handle (the buffer sets are reused)."""
    rng = np.random.default_rng(12)
    per, nb = 12, 8
    ws = [synth.make_window(1200 + i, n_points=int(rng.integers(20, 200)), n_lines=int(rng.integers(0, 50)), n_tagged=0) for i in range(per * nb)]
    s = gpu_api.Solver(max_batch=16)
    st, rep, ms = s.stream(ws, per)
    st2, rep2, ms2 = s.stream(ws, per)
    with pytest.raises(RuntimeError):      # the stream leaves no resident batch behind (its blobs point into the call's pinned result buffers): solving again needs an upload
        s.solve_resident()
    s.close()
    assert ms > 0.0 and len(st) == per * nb
    ref = gpu_api.Solver(max_batch=16)
    for k in range(nb):
        ref.upload(ws[k * per:(k + 1) * per]); ref.solve_resident(); sr, rr = ref.download()
        for b in range(per):
            i = k * per + b
            assert rep[i].status == 0 and rep[i].final_cost == rr[b].final_cost and rep[i].num_iterations == rr[b].num_iterations
            assert np.array_equal(st[i].pose, sr[b].pose) and np.array_equal(st[i].speedbias, sr[b].speedbias)
            assert np.array_equal(st[i].inv_depth, sr[b].inv_depth) and np.array_equal(st[i].line_orth, sr[b].line_orth)
            assert rep2[i].final_cost == rep[i].final_cost and np.array_equal(st2[i].pose, st[i].pose)
    ref.close()


@pytest.mark.parametrize("env", [{"UVS_STREAM_SETS": "2"}, {"UVS_STREAM_SETS": "4"}, {"UVS_STREAM_CHAIN": "1"}, {"UVS_STREAM_D2H_COPY": "1"}, {"UVS_STREAM_D2H_COPY": "2"}])
def test_batch_stream_switches(gpu_api, env, monkeypatch):
    """The A/B switches of the stream (two buffer sets; kernels of consecutive batches chained by events; results gathered on the device and fetched by a copy, or written to the host by the gather
    kernel, instead of by k_solve itself) give the same bits.
    (Each variant in a child process of its own.)"""
    import subprocess, sys, os
    code = ("import importlib, numpy as np, sys; sys.path.insert(0, %r); u = importlib.import_module('uv-slam_amd'); "
            "ws = [u.synth.make_window(1300 + i, n_points=30 + 7 * i, n_lines=i %% 9, n_tagged=0) for i in range(28)]; s = u.api.Solver(max_batch=8); "
            "st, rep, ms = s.stream(ws, 4); print(' '.join(repr(float(r.final_cost)) for r in rep)); print(repr(float(sum(np.abs(x.pose).sum() for x in st))))") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for e in ({}, env):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **e), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout)
    assert outs[0] == outs[1] and len(outs[0].split()) == 29


@pytest.mark.parametrize("form", ["persistent", "fused", "step-wise"])
def test_max_solver_time_in_seconds(gpu_api, form):
    """options.max_solver_time_in_seconds (estimator.cpp:987-991, SOLVER_TIME): checked at the top of every LM iteration (the GPU's 100 MHz wall clock in
    the kernels, the host's clock in the host-driven loop).  A cap far above a solve changes nothing (bitwise); a cap of 100 ns stops after the first
    iteration with UVS_TERM_MAX_TIME and a VALID state (the first accepted step); 0 = no cap is the default."""
    w = synth.make_window(61)
    def run(cap):
        o = abi.default_options(); o.max_solver_time_in_seconds = cap
        s = gpu_api.Solver(o, max_batch=2)
        if form == "persistent": st, rep = s.solve(w)
        elif form == "fused": s.large_comm_init(None); st, rep, _ = s.large_solve_fused(w)
        else: st, rep = s.large_solve(w)
        s.close()
        return st, rep
    s0, r0 = run(0.0); s1, r1 = run(10.0); s2, r2 = run(1e-7)
    assert r0.termination != 7 and r1.num_iterations == r0.num_iterations and r1.final_cost == r0.final_cost and np.array_equal(s1.pose, s0.pose)
    assert r2.termination == 7 and r2.status == 0 and r2.num_iterations == 1 and abs(r2.final_cost - r0.cost[1]) <= 1e-12 * r0.cost[1] and r2.final_cost < r2.initial_cost      # (the candidate-cost sum vs the next linearization's: another summation order)
    assert np.isfinite(s2.pose).all() and not np.array_equal(s2.pose, w.pose)


def test_structure_cache_of_large_windows(gpu_api):
    """Windows of >= 20 000 observations keep their structure (chunking, work split, gather lists) in the handle: a second window with the SAME index arrays
    rewrites only the value sections of the blob and uploads only that prefix.  Same structure / new values, then a different structure, then the first one
    again -- every result bitwise equal to a fresh handle's (UVS_NO_PACK_CACHE=1), in both forms of the large path."""
    import os
    shape = dict(n_points=3000, n_lines=600, n_tagged=450)
    wa = synth.make_window(71, **shape)
    wb = wa.copy(); rng = np.random.default_rng(5)
    wb.pose = wa.pose.copy(); wb.pose[:, :3] += 1e-3 * rng.standard_normal((11, 3)); wb.inv_depth = wa.inv_depth * (1.0 + 1e-3 * rng.standard_normal(len(wa.inv_depth)))
    wb.pt_pj = wa.pt_pj.copy(); wb.pt_pj[:, :2] += 1e-4 * rng.standard_normal((len(wa.pt_lm), 2))
    wc = synth.make_window(72, n_points=2900, n_lines=640, n_tagged=450)      # other structure
    cap = dict(max_batch=1, max_points=3008, max_point_obs=40000, max_lines=648, max_line_obs=8000)
    def run(ws, fused, cache):
        old = os.environ.pop("UVS_NO_PACK_CACHE", None)
        if not cache: os.environ["UVS_NO_PACK_CACHE"] = "1"
        try:
            s = gpu_api.Solver(**cap); s.large_comm_init(None); out = []
            for w in ws:
                st, rep = (s.large_solve_fused(w)[:2] if fused else s.large_solve(w))
                out.append((st.pose.copy(), st.inv_depth.copy(), rep.final_cost, rep.num_iterations))
            s.close()
        finally:
            os.environ.pop("UVS_NO_PACK_CACHE", None)
            if old is not None: os.environ["UVS_NO_PACK_CACHE"] = old
        return out
    seq = [wa, wb, wa, wc, wa, wb]
    for fused in (True, False):
        a = run(seq, fused, True); b = run(seq, fused, False)
        for (pa, da, ca, na), (pb, db, cb, nb) in zip(a, b):
            assert na == nb and ca == cb and np.array_equal(pa, pb) and np.array_equal(da, db)
        assert a[0][2] != a[1][2] and a[0][2] == a[2][2]      # the values did change between the calls, and came back


def test_malformed_windows_are_errors_at_every_entry_point():
    """tests/gpu_fuzz_bad_windows.py: 21 malformed windows (indices out of range, broken grouping, bad prior tables, a NaN state) through the eight
    entry points that take a window -- an error status each time, never a crash, and the handle keeps working.  In a child process: a crash of the
    library must fail this test, not end the session."""
    import subprocess, sys, os
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_fuzz_bad_windows.py")
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-12:])
    assert r.returncode == 0, tail
    assert "0 wrong outcomes" in r.stdout, tail


def test_prior_residual_near_zero_with_a_large_linearized_residual(solver, oracle):
    """The kernel carries the prior in its quadratic form (cost = c0 + g0 . dx + dx . H0 dx / 2: prior_quad), the reference and the oracle as 0.5 |r0 + J0 dx|^2
    (marginalization_factor.cpp:364).  The three terms cancel when the prior is (nearly) satisfied at a point far from its linearization point: |r0| large, r0 + J0 dx ~ 0.
    Such a prior is built here from the product's own one -- r0 := r0 - (r0 + J0 dx(start)) + eps -- and the solve must still walk the oracle's path to its cost."""
    marg = lambda win, flag: solver.marginalize(win, flag)
    rng = np.random.default_rng(9)
    for index in (2, 6):
        w = synth.make_window(index, with_prior=True, marginalize_fn=marg).copy()
        # move the start away from the prior's linearization point, so that |J0 dx| (and with it the new |r0|) is large
        w.pose[:, :3] += 0.05 * rng.standard_normal((w.pose.shape[0], 3)); w.speedbias[:, :3] += 0.05 * rng.standard_normal((w.speedbias.shape[0], 3))
        ev = oracle.evaluate(w, robust=True)
        n = w.prior.n
        r_at_start = np.asarray(ev.prior_r[:n], dtype=np.float64)
        r0_old = w.prior.r0()
        r0_new = r0_old - r_at_start + 1e-7 * rng.standard_normal(n)      # => residual at the start ~ 1e-7, |r0_new| = O(|J0 dx|)
        for k in range(n): w.prior.linearized_residuals[k] = float(r0_new[k])
        ev2 = oracle.evaluate(w, robust=True)
        assert np.abs(np.asarray(ev2.prior_r[:n])).max() < 1e-5 and np.abs(r0_new).max() > 1e3 * np.abs(np.asarray(ev2.prior_r[:n])).max()
        so, ro = oracle.solve(w)
        sg, rg = solver.solve(w)
        assert rg.status == 0 and rg.num_iterations == ro.num_iterations
        assert list(rg.accepted[: rg.num_iterations + 1]) == list(ro.accepted[: ro.num_iterations + 1]) and rg.termination == ro.termination
        assert abs(rg.initial_cost - ro.initial_cost) <= 1e-9 * ro.initial_cost and abs(rg.final_cost - ro.final_cost) <= 1e-6 * ro.final_cost
        dp, dq = pose_deltas(sg.pose, so.pose)
        assert dp < 1e-6 and dq < 1e-6
