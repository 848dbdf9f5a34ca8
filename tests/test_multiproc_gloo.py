"""world_size-2 test of the multi-GPU design on CPU (gloo): shards keyed by global env id reproduce the single-process
run exactly, and the episode-metric all-reduce returns the single-process totals.  The per-shard step is the oracle
(this is a test of the sharding + collective logic; the kernels themselves are covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_shard(p, rank, world, n, steps, ref):
    from oracle import drift_step as OS
    st = OS.init_state(p, n, seed=3, stride=n, env_offset=rank * n)   # startup draws are keyed by the global env id, like the kernel's
    ep = np.zeros(n, np.int32)
    OS.reset_envs(p, st, ep, ref, np.arange(n), 42, 0, env_offset=rank * n)
    met = np.zeros(16)
    rng = np.random.RandomState(5)
    acts = rng.uniform(-1, 1, (steps, n * world, 2)).astype(np.float32)
    for k in range(steps):
        OS.step(p, st, ep, ref, acts[k, rank * n:(rank + 1) * n], 42, k, met, env_offset=rank * n)
    return st, ep, met


def _worker(rank, world, port, n, steps, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from oracle import drift_reset as R
    from oracle import params as OP
    from wheeledlab_amd import dist as D
    r, _, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world) and D.shard_offset(n) == rank * n
    p = OP.drift_params()
    p.max_episode_length = 12  # force time-outs within the short run
    ref = R.ref_pose_table(R.reference_poses(np.random.RandomState(0).rand(20)))
    st, ep, met = _run_shard(p, rank, world, n, steps, ref)
    m = torch.from_numpy(met.copy())
    D.allreduce_metrics(m)
    t = D.max_over_ranks(float(rank + 1))
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), st=st, ep=ep, met_all=m.numpy(), tmax=t)
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_shards_equal_single_process(tmp_path):
    from oracle import drift_reset as R
    from oracle import params as OP
    n, world, steps = 96, 2, 30
    port = _free_port()
    mp.start_processes(_worker, args=(world, port, n, steps, str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    p = OP.drift_params()
    p.max_episode_length = 12
    ref = R.ref_pose_table(R.reference_poses(np.random.RandomState(0).rand(20)))
    st1, ep1, met1 = _run_shard(p, 0, 1, n * world, steps, ref)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    np.testing.assert_array_equal(np.concatenate([r0["st"], r1["st"]], 1), st1)     # bit-identical to the 1-rank run
    np.testing.assert_array_equal(np.concatenate([r0["ep"], r1["ep"]]), ep1)
    np.testing.assert_allclose(r0["met_all"], met1, rtol=1e-12)                     # all-reduce == global totals
    np.testing.assert_array_equal(r0["met_all"], r1["met_all"])
    assert met1[8] > 0 and met1[9] > 0
    assert float(r0["tmax"]) == 2.0 == float(r1["tmax"])
