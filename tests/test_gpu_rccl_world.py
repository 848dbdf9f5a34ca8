"""SURVEY.md 8(e) on real hardware with more than one GPU: one process per GPU over RCCL (`nccl` backend, `device_id=`), each a
4096-env drift shard placed by `env_offset`.  Skipped on a one-GPU box (the driver's test box has one; the first multi-GPU lease
runs it): until then the N > 1 path is covered by the world-2 gloo tests on CPU (tests/test_multiproc_gloo.py) and by two gloo
ranks sharing one GPU through bench.py (tests/test_gpu_bench_contract.py).

Asserted: (a) every shard's state, episode lengths and observations after 8 steps equal the matching slice of ONE 4096 x W batch
stepped on rank 0's GPU (RNG keyed by the global env id: bit for bit), (b) the all-reduced episode-metric vector equals the big
batch's, (c) a sum all-reduce of ones sees W ranks on the nccl backend."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N_PER_RANK, STEPS, SEED = 4096, 8, 42


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _actions(world):
    g = torch.Generator().manual_seed(77)                      # every rank draws the same global action block and slices it
    return torch.rand(STEPS, N_PER_RANK * world, 2, generator=g) * 2 - 1


def _run(batch, actions):
    batch.reset()
    obs = []
    for k in range(STEPS):
        o, _, _, _ = batch.step(actions[k])
        obs.append(o.clone())
    return torch.stack(obs)


def _worker(rank, world, port, backend):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    from wheeledlab_amd import dist as D
    from wheeledlab_amd.core import DriftBatch
    from wheeledlab_amd.params import drift_params
    r, local, w = D.init_from_env(backend)                          # nccl (= RCCL) + device_id, one GPU per rank
    assert (r, w) == (rank, world) and dist.get_backend() == backend
    if backend == "nccl":
        assert torch.cuda.current_device() == local == rank
    dev = torch.device("cuda", torch.cuda.current_device())
    wire = (lambda t: t.contiguous()) if backend == "nccl" else (lambda t: t.cpu().contiguous())   # gloo gathers host tensors only
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    assert int(ones.item()) == world                                # (c) the collective spans every rank
    p = drift_params()
    p.max_episode_length = 6                                        # time-outs (and their in-step resets) inside the 8 steps
    n = N_PER_RANK
    acts = _actions(world)
    shard = DriftBatch(n, device=dev, params=p, seed=SEED, env_offset=D.shard_offset(n))
    assert shard.env_offset == rank * n
    obs = _run(shard, acts[:, rank * n:(rank + 1) * n].contiguous().to(dev))
    m = shard.read_metrics(zero=False).clone()
    D.allreduce_metrics(m)                                          # the path's one collective
    # everything to rank 0 over RCCL
    mine = [wire(shard.state[:, :n]), wire(shard.episode_len[:n]), wire(obs)]
    got = [[torch.empty_like(t) for _ in range(world)] if rank == 0 else None for t in mine]
    for t, dst in zip(mine, got):
        dist.gather(t, dst, dst=0)
    if rank == 0:
        state, eplen, obss = got
        big = DriftBatch(n * world, device=dev, params=p, seed=SEED)
        big_obs = _run(big, acts.to(dev))
        for k in range(world):                                      # (a) shard k == slice k of the one big batch, bit for bit
            sl = slice(k * n, (k + 1) * n)
            assert torch.equal(state[k].to(dev), big.state[:, sl]), f"state of shard {k}"
            assert torch.equal(eplen[k].to(dev), big.episode_len[sl]), f"episode_len of shard {k}"
            assert torch.equal(obss[k].to(dev), big_obs[:, sl]), f"observations of shard {k}"
        bm = big.read_metrics(zero=False)
        assert bm[8] > 0 and bm[9] > 0                              # resets and time-outs did happen
        torch.testing.assert_close(m, bm, rtol=1e-5, atol=1e-3)     # (b) sums of the same terms in another order
        assert torch.equal(m[8:16], bm[8:16])                       # the counts exactly
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs: RCCL world >= 2 (SURVEY 8(e))")
@pytest.mark.timeout(600)
def test_rccl_shards_equal_the_single_gpu_batch():
    world = min(torch.cuda.device_count(), 8)
    mp.start_processes(_worker, args=(world, _free_port(), "nccl"), nprocs=world, join=True, start_method="spawn")


@pytest.mark.timeout(600)
def test_the_same_worker_over_gloo_with_two_ranks_sharing_one_gpu():
    """the one-GPU box's stand-in: the identical worker (shards by env_offset, metric all-reduce, gather, comparison with the big
    batch) with the gloo backend, so that the test above is known to be right before the first multi-GPU lease runs it"""
    mp.start_processes(_worker, args=(2, _free_port(), "gloo"), nprocs=2, join=True, start_method="spawn")


@pytest.mark.timeout(900)
def test_eight_gloo_ranks_sharing_one_gpu():
    """world = 8 (8 x 4096 envs), the size of the driver's multi-GPU lease: shard k of eight == slice k of ONE
    32 768-env batch bit for bit, the all-reduced metric vector == the big batch's"""
    mp.start_processes(_worker, args=(8, _free_port(), "gloo"), nprocs=8, join=True, start_method="spawn")
