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


def _collective(dist, world: int) -> bool:
    """collectives run whenever a process group exists -- also at world size 1 under torchrun, so that the RCCL path
    (init_process_group("nccl"), barrier, CUDA-tensor all_reduce) is the same code on 1 and on 8 GPUs"""
    try:
        return world > 1 or (dist is not None and dist.is_available() and dist.is_initialized())
    except Exception:
        return world > 1


def barrier(dist, world: int) -> None:
    if _collective(dist, world):
        dist.barrier()


def reduce_timing(dist, world: int, elapsed_s: float, units: int, device=None) -> Tuple[float, int]:
    """Whole-job timing: MAX of the per-rank elapsed time, SUM of the units (stereo pairs) done."""
    if not _collective(dist, world):
        return elapsed_s, units
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    n = torch.tensor([units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())
