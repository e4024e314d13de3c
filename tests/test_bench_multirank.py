"""bench.py's N > 1 contract exercised on ONE device (the driver runs `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` on an 8-GPU node at round end; no
such node is available to the builder): two ranks share the GPU, UVS_BENCH_BACKEND=gloo carries the barrier / max-reduce on the host, the configs[3] leg -- whose all-reduces
go through RCCL -- reports itself skipped.  What must not fail on plumbing: rendezvous, per-rank window indices, rank 0's single JSON line and its fields."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_with_two_ranks_on_one_device():
    env = dict(os.environ, UVS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32", "--no-cpu-baseline", "--no-replay", "--no-stream"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out.stdout[-2000:]                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["unit"] == "solves/s" and d["value"] > 0 and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["config"]["windows_per_gpu"] == 32 and d["config"]["parallelism"] == "replicas x2"
    assert abs(d["value"] - 2 * 32 * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]      # whole-job rate = windows of ALL ranks / max-over-ranks time
    assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1 and d["cpu_baseline"] is None      # (cpu_baseline: rank 0 at N = 1 only)
    assert "skipped" in d["large_window"]                            # the fused loop all-reduces over RCCL: a gloo dry run says so instead of failing
    assert d["lm_iterations_mean"] > 0 and d["final_cost_mean"] > 0
