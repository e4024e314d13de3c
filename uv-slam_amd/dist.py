"""Multi-GPU plumbing for independent windows (BASELINE configs[2]): one process per GPU, replicas only.

Independent sliding windows share nothing, so the batch is sharded by window index and NO data-path collective is
needed; torch.distributed (RCCL on ROCm, gloo in the CPU tests) is used only for the
barrier and for the max-over-ranks wall time that bench.py reports.
"""
import os


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_window_indices(rank, world, per_gpu):
    """Weak scaling: rank r owns global window indices [r*per_gpu, (r+1)*per_gpu) -> seeds 1000 + index."""
    if not (0 <= rank < world) or per_gpu < 0:
        raise ValueError("bad shard request")
    return list(range(rank * per_gpu, (rank + 1) * per_gpu))


def split_windows(n_total, rank, world):
    """Strong-scaling split of a fixed set of n_total windows: contiguous, sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of a python float (identity when not distributed)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
