"""N>1 path on CPU: world_size 2 over gloo.  Independent windows are sharded with no data-path collective; the only
collectives are the barrier and the max-over-ranks wall time bench.py reports (exercised here exactly as bench.py does)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import uvs, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, per_gpu, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    u = importlib.import_module("uv-slam_amd")
    from oracle_binding import Oracle            # CPU stand-in for the per-GPU solve (no GPU in this container)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = u.dist.shard_window_indices(rank, world, per_gpu)
    orc = Oracle()
    costs = [orc.solve(u.synth.make_window(i, n_points=30, n_lines=8, n_tagged=6))[1].final_cost for i in idx]
    dist.barrier()
    elapsed = u.dist.max_over_ranks(0.1 * (rank + 1), dist)
    total = u.dist.sum_over_ranks(len(idx), dist)
    q.put((rank, idx, costs, elapsed, total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replicas_cover_the_batch_without_collectives():
    world, per_gpu = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_gpu, q)) for r in range(world)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs: p.join(60)
    assert all(p.exitcode == 0 for p in procs)
    all_idx = [i for r in res for i in r[1]]
    assert sorted(all_idx) == list(range(world * per_gpu))                  # disjoint, complete, weak scaling
    assert all(abs(r[3] - 0.2) < 1e-12 for r in res)                         # MAX over ranks
    assert all(r[4] == world * per_gpu for r in res)
    # each rank's results equal a single-process run of the same window indices
    from oracle_binding import Oracle
    orc = Oracle()
    for r in res:
        for i, c in zip(r[1], r[2]):
            assert orc.solve(synth.make_window(i, n_points=30, n_lines=8, n_tagged=6))[1].final_cost == c


def test_shard_helpers():
    d = uvs.dist
    assert d.shard_window_indices(1, 4, 3) == [3, 4, 5]
    parts = [d.split_windows(10, r, 4) for r in range(4)]
    assert sum(parts, []) == list(range(10)) and max(map(len, parts)) - min(map(len, parts)) <= 1
    assert d.max_over_ranks(1.5) == 1.5
    with pytest.raises(ValueError):
        d.shard_window_indices(4, 4, 1)
