"""configs[3] multi-rank path.

CPU (gloo, world 2): the landmark-sharded partial normal equations sum to the full reduced system (the algebra behind the
all-reduce), exchanged with a real torch.distributed all-reduce.
GPU (marked gpu): two ranks SHARE the one GPU of the box (gloo, host-staged all-reduce) and run the real step-wise solver on
their landmark shards; the result must equal the single-process large solve.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import uvs, abi, synth, dense_normal_equations, pose_deltas

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _visual_only(w):
    v = w.copy(); v.imu = []; v.prior = None
    return v


def _reduced_partial(w, ev, n_full_points, n_full_lines):
    """Schur complement of this window's landmarks onto the 165 frame dofs (undamped landmarks regularised by 1e-6)."""
    H, g = dense_normal_equations(w, ev)
    F = 165
    Hll = H[F:, F:] + 1e-6 * np.eye(H.shape[0] - F)
    S = H[:F, :F] - H[:F, F:] @ np.linalg.solve(Hll, H[F:, :F])
    gr = g[:F] - H[:F, F:] @ np.linalg.solve(Hll, g[F:])
    return S, gr


def _cpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_binding import Oracle
    from helpers import synth as sy
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = _visual_only(sy.make_window(51, n_points=60, n_lines=16, n_tagged=12))
    shard, pk, lk = sy.shard_landmarks(w, rank, world)
    ev = Oracle().evaluate(shard, robust=True)
    S, g = _reduced_partial(shard, ev, 60, 16)
    t = torch.from_numpy(np.concatenate([S.ravel(), g, [ev.cost]]))
    dist.all_reduce(t)
    q.put((rank, t.numpy().copy(), len(pk), len(lk)))
    dist.barrier(); dist.destroy_process_group()


def test_sharded_partials_sum_to_the_full_reduced_system(oracle):
    world = 2
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_cpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda r: r[0])
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][2] + res[1][2] == 60 and res[0][3] + res[1][3] == 16
    assert np.array_equal(res[0][1], res[1][1])                           # every rank holds the same reduced system
    w = _visual_only(synth.make_window(51, n_points=60, n_lines=16, n_tagged=12))
    ev = oracle.evaluate(w, robust=True)
    S, g = _reduced_partial(w, ev, 60, 16)
    full = np.concatenate([S.ravel(), g, [ev.cost]])
    assert np.abs(res[0][1] - full).max() <= 1e-9 * np.abs(full).max()


def test_shard_landmarks_partition():
    w = synth.make_window(52)
    parts = [synth.shard_landmarks(w, r, 4) for r in range(4)]
    assert sorted(np.concatenate([p[1] for p in parts])) == list(range(150)) and sorted(np.concatenate([p[2] for p in parts])) == list(range(40))
    assert sum(len(p[0].pt_lm) for p in parts) == 750 and sum(len(p[0].ln_lm) for p in parts) == 280
    s0, pk, lk = parts[1]
    assert np.array_equal(s0.inv_depth, w.inv_depth[pk]) and np.array_equal(s0.pose, w.pose) and len(s0.imu) == 10
    k = 7; obs = np.nonzero(s0.pt_lm == k)[0]; gobs = np.nonzero(w.pt_lm == pk[k])[0]
    assert np.array_equal(s0.pt_pj[obs], w.pt_pj[gobs])


def _gpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    u = importlib.import_module("uv-slam_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    w = u.synth.make_window(53, n_points=400, n_lines=100, n_tagged=75)
    shard, pk, lk = u.synth.shard_landmarks(w, rank, world)
    s = u.api.Solver(device=0, max_batch=2)
    st, rep = s.large_solve(shard, dist=dist, device="cuda:0")
    q.put((rank, st.pose.copy(), st.inv_depth.copy(), pk, rep.final_cost, rep.num_iterations, list(rep.accepted[:11])))
    s.close()
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_single_process(gpu_api):
    world = 2
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda r: r[0])
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    w = synth.make_window(53, n_points=400, n_lines=100, n_tagged=75)
    s = gpu_api.Solver(max_batch=2)
    st, rep = s.large_solve(w)
    s.close()
    for r in res:
        assert r[5] == rep.num_iterations and r[6] == list(rep.accepted[:11])
        # the shards sum the pose-block partials in a different order; over 10 LM iterations from an initial cost of 1e10 the
        # round-off difference grows to ~1e-8 relative in the final cost
        assert abs(r[4] - rep.final_cost) <= 1e-7 * rep.final_cost
        assert pose_deltas(r[1], st.pose)[0] < 1e-7
        assert np.abs(r[2] - st.inv_depth[r[3]]).max() < 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize("comm", [None, "self"])
def test_fused_device_side_loop_equals_the_host_driven_one(gpu_api, comm):
    """uvs_large_solve_fused: trust-region control on the device, every launch of the solve enqueued without a host round trip, the
    exchange vectors all-reduced in place by the handle's own RCCL communicator (comm == "self": a one-rank communicator, which is
    what one GPU can exercise of it; None: no communicator).  Same kernels, same numbers as the step-wise host-driven loop."""
    w = synth.make_window(54, n_points=900, n_lines=200, n_tagged=150)
    s = gpu_api.Solver(max_batch=2, max_points=1000, max_point_obs=12000, max_lines=256, max_line_obs=3000)
    st0, rep0 = s.large_solve(w)
    s.large_comm_init(comm)
    st1, rep1, ms = s.large_solve_fused(w)
    st2, rep2, ms2 = s.large_solve_fused(w)          # the handle is reusable
    s.close()
    n = rep0.num_iterations
    assert rep1.num_iterations == n and list(rep1.accepted[:n + 1]) == list(rep0.accepted[:n + 1]) and rep1.termination == rep0.termination
    # after the rejected steps the fused loop re-damps the stored linearization while the host-driven loop linearizes again: the reduced systems
    # then differ in the last bits, the costs by ~1e-14, and the radius -- a function of the RATIO of two cost differences -- by ~1e-11
    assert np.allclose(np.array(rep1.radius[:n + 1]), np.array(rep0.radius[:n + 1]), rtol=1e-9)
    assert np.allclose(np.array(rep1.cost[:n + 1]), np.array(rep0.cost[:n + 1]), rtol=1e-12)
    assert abs(rep1.final_cost - rep0.final_cost) <= 1e-12 * rep0.final_cost and rep1.initial_cost == rep0.initial_cost
    assert pose_deltas(st1.pose, st0.pose)[0] < 1e-12 and np.abs(st1.inv_depth - st0.inv_depth).max() < 1e-12
    assert np.abs(st1.line_orth - st0.line_orth).max() < 1e-10
    assert np.array_equal(st1.pose, st2.pose) and rep2.final_cost == rep1.final_cost and 0.0 < ms < 100.0


@pytest.mark.gpu
def test_fused_loop_early_termination_and_zero_iterations(gpu_api):
    w = synth.make_window(55, n_points=300, n_lines=60, n_tagged=40)
    for kw in (dict(max_num_iterations=0), dict(max_num_iterations=40, function_tolerance=5e-3), dict(max_num_iterations=3)):
        o = abi.default_options()
        for k, v in kw.items(): setattr(o, k, v)
        s = gpu_api.Solver(o, max_batch=2)
        st0, rep0 = s.large_solve(w)
        s.large_comm_init(None)
        st1, rep1, ms = s.large_solve_fused(w)
        s.close()
        assert rep1.num_iterations == rep0.num_iterations and rep1.termination == rep0.termination, kw
        assert abs(rep1.final_cost - rep0.final_cost) <= 1e-12 * rep0.final_cost and pose_deltas(st1.pose, st0.pose)[0] < 1e-12


# ---------------------------------------------------------------------------------------------------------------------------------
# The fused loop with N = 2.  No multi-GPU node is available to the tests, so two processes share the one GPU and the library's
# dlopen hook (UVS_RCCL_LIB) is pointed at tests/shim/libuvs_fake_nccl.so, which carries the two all-reduces through shared memory.
# Everything on the solver's side -- the per-rank MAX slots inside the SUM payload, k_large_sum_bsums, the 5-scalar exchange,
# k_large_decide on identical numbers on both ranks -- is the code that runs over RCCL / xGMI.
SHIM_DIR = os.path.join(ROOT, "tests", "shim")
SHIM = os.path.join(SHIM_DIR, "libuvs_fake_nccl.so")


def _build_shim():
    src = os.path.join(SHIM_DIR, "fake_nccl.cpp")
    if os.path.exists(SHIM) and os.path.getmtime(SHIM) >= os.path.getmtime(src):
        return
    import subprocess
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O2", "-std=c++17", "-shared", "-fPIC", "-w", src, "-o", SHIM, "-lrt"])


FUSED2 = dict(n_points=900, n_lines=200, n_tagged=150)


def _fused_worker(rank, world, port, q, seed, shape):
    os.environ["UVS_RCCL_LIB"] = SHIM            # before libuvs_solver.so resolves RCCL
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    u = importlib.import_module("uv-slam_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # carries the 128-byte communicator id, nothing else
    w = u.synth.make_window(seed, **shape)
    shard, pk, lk = u.synth.shard_landmarks(w, rank, world)
    s = u.api.Solver(device=0, max_batch=1, max_points=len(pk) + 8, max_point_obs=12 * len(pk) + 64, max_lines=len(lk) + 8, max_line_obs=12 * len(lk) + 64)
    s.large_comm_init(dist)
    st, rep, ms = s.large_solve_fused(shard)
    st2, rep2, _ = s.large_solve_fused(shard)      # the communicator and the handle are reusable; both ranks must stay in step
    n = rep.num_iterations
    q.put((rank, st.pose.copy(), st.speedbias.copy(), st.inv_depth.copy(), st.line_orth.copy(), pk, lk, rep.final_cost, rep.initial_cost, n, list(rep.accepted[:n + 1]),
           list(rep.cost[:n + 1]), list(rep.radius[:n + 1]), rep.termination, bool(np.array_equal(st.pose, st2.pose) and rep2.final_cost == rep.final_cost)))
    s.close()
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,shape", [(56, FUSED2), (57, dict(n_points=3000, n_lines=700, n_tagged=500))])
def test_fused_loop_with_two_ranks_on_one_gpu(gpu_api, oracle, seed, shape):
    """uvs_large_solve_fused with nranks = 2: landmark shards k mod 2, both all-reduces per iteration through the handle's communicator.
    Must equal the one-process fused solve of the whole window (same LM trace, final cost to 1e-7, states) and the CPU oracle."""
    _build_shim()
    world = 2
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_fused_worker, args=(r, world, port, q, seed, shape)) for r in range(world)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=420) for _ in range(world)], key=lambda r: r[0])
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    w = synth.make_window(seed, **shape)
    s = gpu_api.Solver(max_batch=1, max_points=shape["n_points"] + 8, max_point_obs=12 * shape["n_points"], max_lines=shape["n_lines"] + 8, max_line_obs=12 * shape["n_lines"])
    s.large_comm_init(None)
    st, rep, _ = s.large_solve_fused(w)
    s.close()
    so, ro = oracle.solve(w)
    n = rep.num_iterations
    assert n == ro.num_iterations and list(rep.accepted[:n + 1]) == list(ro.accepted[:n + 1])
    # both ranks decided on identical numbers: identical traces, bit for bit, and identical frame states
    assert res[0][9:14] == res[1][9:14] and res[0][7] == res[1][7]
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    inv_depth = np.zeros(shape["n_points"]); line_orth = np.zeros((shape["n_lines"], 4))
    for r in res:
        assert r[14], "second solve on the same communicator differs"
        assert r[9] == n and r[10] == list(rep.accepted[:n + 1]) and r[13] == rep.termination
        assert np.allclose(np.array(r[11]), np.array(rep.cost[:n + 1]), rtol=1e-7) and np.allclose(np.array(r[12]), np.array(rep.radius[:n + 1]), rtol=1e-6)
        assert abs(r[7] - rep.final_cost) <= 1e-7 * rep.final_cost and abs(r[8] - rep.initial_cost) <= 1e-12 * rep.initial_cost
        assert pose_deltas(r[1], st.pose)[0] < 1e-7 and np.abs(r[2] - st.speedbias).max() < 1e-7
        inv_depth[r[5]] = r[3]; line_orth[r[6]] = r[4].reshape(-1, 4)
    assert np.abs(inv_depth - st.inv_depth).max() < 1e-7 and np.abs(line_orth - st.line_orth.reshape(-1, 4)).max() < 1e-6
    dp, da = pose_deltas(res[0][1], so.pose)
    assert dp < 1e-6 and da < 1e-6 and abs(res[0][7] - ro.final_cost) <= 1e-7 * ro.final_cost
