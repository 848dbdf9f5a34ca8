"""Multi-GPU plumbing: one process per GPU, envs sharded by contiguous global-id ranges, no data-path collective.

Envs never interact (the reference even stacks all cars at one origin, mushr_drift_env_cfg.py:373), so rank r simply
owns global envs [r*n, (r+1)*n): the in-kernel RNG is keyed by the GLOBAL env id, which makes a W-rank run of n envs
each bit-identical to a 1-rank run of W*n envs.  The only thing that crosses GPUs is the episode-metric vector
(WL_M_COUNT floats): one sum all-reduce over RCCL/xGMI at the logging cadence."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None):
    """initialise torch.distributed from torchrun-style env vars; returns (rank, local_rank, world)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_offset(n_envs_per_rank: int) -> int:
    """global id of this rank's env 0"""
    return (dist.get_rank() if dist.is_initialized() else 0) * n_envs_per_rank


def allreduce_metrics(m: torch.Tensor) -> torch.Tensor:
    """sum the episode-metric vector over ranks (in place); no-op for a single process"""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(m, op=dist.ReduceOp.SUM)
    return m


def max_over_ranks(x: float, device="cpu") -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
