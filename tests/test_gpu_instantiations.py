"""The persistent kernel exists in two instantiations (csrc/uvs_solve512.hip: 512 threads, two waves per SIMD, evaluator / gatherer wave roles -- what
launch_solve uses -- and the 256-thread one that UVS_KSOLVE_NT=256 selects at uvs_create).  Both must walk the same Levenberg-Marquardt path on the same
windows and agree with the oracle; the environment variable is read per handle, so one process can hold both."""
import importlib, os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
uvs = importlib.import_module("uv-slam_amd")
synth = uvs.synth
pytestmark = pytest.mark.gpu


def _solve(windows, nt, opts=None):
    old = os.environ.get("UVS_KSOLVE_NT")
    os.environ["UVS_KSOLVE_NT"] = str(nt)
    try:
        s = uvs.api.Solver(opts=opts, device=0, max_batch=len(windows))
    finally:
        if old is None: os.environ.pop("UVS_KSOLVE_NT", None)
        else: os.environ["UVS_KSOLVE_NT"] = old
    s.upload(windows); s.solve_resident()
    states, reps = s.download()
    s.close()
    return states, reps


def _trace(rep):
    n = rep.num_iterations
    return [int(rep.accepted[i]) for i in range(n + 1)], rep.termination


@pytest.mark.parametrize("kind", ["canonical_with_prior", "mixed_shapes", "rejections"])
def test_both_instantiations_walk_the_same_path(kind):
    from oracle_binding import Oracle          # checker only
    if kind == "canonical_with_prior":
        orc = Oracle()
        windows = [synth.make_window(i, with_prior=True, marginalize_fn=lambda win, flag: orc.marginalize(win, flag)) for i in range(6)]
    elif kind == "mixed_shapes":
        rng = np.random.default_rng(11)
        windows = [synth.make_window(4100 + i, n_points=int(rng.integers(20, 320)), n_lines=int(rng.integers(0, 70)), n_tagged=0, pt_track=int(rng.integers(2, 10)),
                                     ln_track=int(rng.integers(3, 10))) for i in range(10)]
        windows.append(synth.make_window(4200, n_points=150, n_lines=0, n_tagged=0))      # no line chunk at all
        windows.append(synth.make_window(4201, n_points=12, n_lines=3, n_tagged=1))       # chunks far smaller than a wave
    else:
        rng = np.random.default_rng(5)      # perturbed starts: the LM loop rejects steps and re-damps (the 512-thread kernel keeps the gather accumulators of a linearization in the workspace for that)
        windows = []
        for i in range(6):
            w = synth.make_window(7000 + i).copy()
            w.pose[2:, :3] += 0.2 * rng.standard_normal((9, 3)); w.inv_depth *= np.exp(0.2 * rng.standard_normal(len(w.inv_depth)))
            windows.append(w)
    st512, rep512 = _solve(windows, 512)
    st256, rep256 = _solve(windows, 256)
    orc = Oracle()
    for i, w in enumerate(windows):
        assert rep512[i].status == 0 and rep256[i].status == 0
        if kind != "rejections":      # (heavily perturbed windows may part ways by round-off amplification: tests/test_gpu_stress.py measures that; here only the two kernels' agreement is required)
            so, ro = orc.solve(w)
            assert _trace(rep512[i]) == _trace(ro), (kind, i)
        if _trace(rep512[i]) != _trace(rep256[i]):
            assert kind == "rejections", (kind, i)
            continue
        assert abs(rep512[i].final_cost - rep256[i].final_cost) <= 1e-9 * max(1.0, abs(rep256[i].final_cost)), (kind, i)
        a, b = np.asarray(st512[i].pose), np.asarray(st256[i].pose)
        assert np.abs(a - b).max() < 1e-8, (kind, i)


def _fused(w, chunks_nt, solve_nt):
    old = {k: os.environ.get(k) for k in ("UVS_LARGE_CHUNKS_NT", "UVS_LARGE_SOLVE_NT")}
    os.environ["UVS_LARGE_CHUNKS_NT"] = str(chunks_nt); os.environ["UVS_LARGE_SOLVE_NT"] = str(solve_nt)
    try:
        s = uvs.api.Solver(device=0, max_batch=1, max_points=max(1000, len(w.inv_depth)), max_point_obs=max(16000, len(w.pt_lm)), max_lines=max(1000, len(w.line_orth)), max_line_obs=max(16000, len(w.ln_lm)))
    finally:
        for k, v in old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
    st, rep, _ = s.large_solve_fused(w)
    s.close()
    return st, rep


@pytest.mark.parametrize("shape", ["canonical", "many_chunks"])
def test_landmark_sharded_kernels_in_both_instantiations(shape):
    """k_large_chunks and k_large_solve exist with 512 threads (wave roles, csrc/uvs_solve512.hip: the default) and with 256 (UVS_LARGE_CHUNKS_NT / UVS_LARGE_SOLVE_NT = 256):
    the fused loop must take the same LM path with either and end at the same state."""
    w = synth.make_window(31) if shape == "canonical" else synth.make_window(32, n_points=2500, n_lines=400, n_tagged=200)
    a, ra = _fused(w, 512, 512)
    for cn, sn in ((256, 256), (512, 256), (256, 512)):
        b, rb = _fused(w, cn, sn)
        assert ra.status == 0 and rb.status == 0
        assert _trace(ra) == _trace(rb), (shape, cn, sn)
        assert abs(ra.final_cost - rb.final_cost) <= 1e-9 * max(1.0, abs(rb.final_cost)), (shape, cn, sn)
        assert np.abs(np.asarray(a.pose) - np.asarray(b.pose)).max() < 1e-8, (shape, cn, sn)
