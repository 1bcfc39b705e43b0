"""Record-axis sharding of the hot path over the GPUs of one node (one process per GPU).

Every operator is 1-D along one core axis and independent along all others (the property the
reference's `dask="parallelized"` path relies on, xgcm/grid.py:786-789), so a field shards by a
contiguous block split of its outermost non-core axis with NO data-path collective: outputs
stay sharded like a dask array chunked along `time`.  `torch.distributed` (backend "nccl" =
RCCL over xGMI on the GPU box, "gloo" in CPU tests) is used only for barriers and for reducing
scalars: the max-over-ranks time and order-independent checksums.
"""

from __future__ import annotations

from typing import Optional, Tuple


def shard_bounds(n_units: int, world: int, rank: int) -> Tuple[int, int]:
    """[start, stop) of the contiguous block of `n_units` owned by `rank`; the first
    `n_units % world` ranks own one extra unit (90 levels on 8 GPUs -> 12,12,11,11,11,11,11,11)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, extra = divmod(int(n_units), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def reduce_max(value: float, dist=None, device: Optional[str] = None) -> float:
    """max over ranks of a python float (identity without a process group)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value: float, dist=None, device: Optional[str] = None) -> float:
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_throughput(local_units: float, local_seconds: float, dist=None, device: Optional[str] = None) -> Tuple[float, float]:
    """(units of all ranks / max-over-ranks seconds, that max) -- the bench contract's aggregate."""
    total = reduce_sum(local_units, dist, device)
    tmax = reduce_max(local_seconds, dist, device)
    return total / tmax, tmax
