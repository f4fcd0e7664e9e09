"""Multi-GPU sharding of the many-sequence mode: independent stereo+IMU streams are
partitioned across ranks, one process per GPU; there is NO data-path collective.  torch.distributed
is used only for the lock-step barrier and for reducing the timing (MAX) and the unit count (SUM).
Backend "nccl" is RCCL over xGMI on the MI355X node; "gloo" is used by the CPU tests."""
from __future__ import annotations

import os
from typing import List, Tuple


def shard_streams(n_streams: int, world: int, rank: int) -> List[int]:
    """Stream s lives on rank s mod world (8 EuRoC sequences on 8 GPUs -> one each)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_streams, world))


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def barrier(dist, world: int) -> None:
    if world > 1:
        dist.barrier()


def reduce_timing(dist, world: int, elapsed_s: float, units: int, device=None) -> Tuple[float, int]:
    """Whole-job timing: MAX of the per-rank elapsed time, SUM of the units (stereo pairs) done."""
    if world == 1:
        return elapsed_s, units
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())
