"""Multi-GPU plumbing: one process per GPU, envs sharded by contiguous global-id ranges, no data-path collective.

Envs never interact (the reference even stacks all cars at one origin, mushr_drift_env_cfg.py:373), so rank r simply
owns global envs [r*n, (r+1)*n): the in-kernel RNG is keyed by the GLOBAL env id, which makes a W-rank run of n envs
each bit-identical to a 1-rank run of W*n envs.  The only thing that crosses GPUs on the env.step() path is the
episode-metric vector (WL_M_COUNT floats): one sum all-reduce over RCCL/xGMI at the logging cadence.
A data-parallel learner on top (rl/ppo.py) adds one all-reduce per minibatch step: the flat gradient row of the 64-64 MLPs
(10 440 floats, 41 KB -- latency-bound on xGMI, so it is ONE collective, not one per tensor) and, in the torch path, the
scalar KL mean that steers the adaptive learning rate, so that every rank takes the same step without a broadcast."""
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
        if backend is None:   # WL_DIST_BACKEND=gloo: debugging aid (several ranks sharing one GPU)
            backend = os.environ.get("WL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")   # "nccl" IS RCCL on ROCm
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        elif torch.cuda.is_available():
            local_rank %= torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
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


def ranks_agree(t: torch.Tensor) -> bool:
    """True when `t` is bit-identical on every rank (a sync check for replicated state: parameters, learning rate)"""
    w = world_size()
    if w == 1:
        return True
    lo, hi = t.detach().clone(), t.detach().clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def average_(t: torch.Tensor) -> torch.Tensor:
    """mean over ranks, in place (one all-reduce); no-op for a single process"""
    w = world_size()
    if w > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(w)
    return t


def average_gradients_(params) -> None:
    """mean of the .grad of `params` over ranks as ONE flat all-reduce (the nets are ~10 K parameters: per-tensor collectives
    would pay the xGMI latency twelve times)"""
    if world_size() == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads])
    average_(flat)
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()
