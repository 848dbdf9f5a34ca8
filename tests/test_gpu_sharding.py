"""Env shards are exact, on the PRODUCT path (the gloo test in test_multiproc_gloo.py shards the oracle):

* the startup domain randomisation (wl_startup_randomize) equals the numpy restatement keyed by the global env id,
  has the reference's structure (20 friction buckets, mu_d <= mu_s, ranges of mushr_drift_env_cfg.py:98-119,145-154),
  and a shard holds exactly the rows of the big batch;
* two half-size batches built with `env_offset` -- nothing copied between them and the big batch -- stay bit-identical
  to the big batch through resets, pushes, noise and K steps, for all three tasks;
* the N > 1 code path of bench.py / dist.py runs against RCCL (a world of ONE process: the collective calls, the async
  handle and the deferred join are the ones the 8-GPU run makes)."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from wheeledlab_amd import _abi
    return _abi.load()


def test_startup_randomisation_matches_the_keyed_restatement(lib):
    from oracle import startup as OSU
    from wheeledlab_amd import _abi as A
    from wheeledlab_amd.core import DriftBatch
    n = 4096 + 37
    env = DriftBatch(n, device=DEV, seed=11)
    torch.cuda.synchronize()
    st = env.state.cpu().numpy()
    mu_s, mu_d, damp, mass, bucket = OSU.draw(n, 11, 0, **OSU.DRIFT)
    for row, want in ((A.S_MU_S, mu_s), (A.S_MU_D, mu_d), (A.S_DAMP, damp), (A.S_MASS, mass)):
        np.testing.assert_allclose(st[row, :n], want, rtol=3e-7, atol=0)      # fmaf vs mul + add: <= 1 ulp
    assert (st[A.S_QW, :n] == 1).all() and (st[:, n:] == 0).all()              # padding columns stay zero
    # structure of the reference's events
    assert len(np.unique(st[A.S_MU_S, :n])) == 20 and len(np.unique(bucket)) == 20
    assert (st[A.S_MU_D, :n] <= st[A.S_MU_S, :n]).all()
    assert st[A.S_MU_S, :n].min() >= 0.3 and st[A.S_MU_S, :n].max() <= 0.5 and st[A.S_MU_D, :n].min() >= 0.3
    assert 10.0 <= st[A.S_DAMP, :n].min() and st[A.S_DAMP, :n].max() <= 50.0 and abs(st[A.S_DAMP, :n].mean() - 30.0) < 1.0
    assert 3.3 <= st[A.S_MASS, :n].min() and st[A.S_MASS, :n].max() <= 3.5 and abs(st[A.S_MASS, :n].mean() - 3.4) < 0.01
    counts = np.bincount(bucket, minlength=20)
    assert counts.min() > 0.6 * n / 20 and counts.max() < 1.4 * n / 20        # buckets are drawn uniformly
    # a shard is a slice of the big batch; another seed is another draw
    off = 1024
    sh = DriftBatch(512, device=DEV, seed=11, env_offset=off)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(sh.state.cpu().numpy()[23:27, :512], st[23:27, off:off + 512])
    other = DriftBatch(512, device=DEV, seed=12)
    assert not np.array_equal(other.state.cpu().numpy()[A.S_DAMP, :512], st[A.S_DAMP, :512])
    # randomize=False: mid-points
    fixed = DriftBatch(64, device=DEV, seed=11, randomize=False)
    f = fixed.state.cpu().numpy()
    np.testing.assert_allclose(f[23:27, :64], np.array([[0.4], [0.4], [30.0], [3.4]], np.float32) * np.ones((1, 64), np.float32), rtol=1e-6)


def test_wheel_mass_randomisation_matches_the_keyed_restatement(lib):
    """VisualEventsRandomCfg's three startup terms through the registry's flattening -> wl_startup_randomize vs oracle/startup.py"""
    from oracle import startup as OSU
    from wheeledlab_amd import _abi as A
    from wheeledlab_amd.core import VisualBatch
    from wheeledlab_amd.envs.flatten import flatten_visual_cfg
    from wheeledlab_amd.tasks.visual import MushrVisualRLRandomEnvCfg
    flat = flatten_visual_cfg(MushrVisualRLRandomEnvCfg())
    su = flat.startup
    n, off = 2048 + 5, 4096
    env = VisualBatch(n, device=DEV, seed=19, env_offset=off, params=flat.params, startup=su)
    torch.cuda.synchronize()
    st = env.state.cpu().numpy()
    mu_s, mu_d, damp, mass, _ = OSU.draw(n, 19, off, wheel_mu_s=su.wheel_mu_s, wheel_mu_d=su.wheel_mu_d, mu_buckets=su.mu_buckets,
                                         mu_consistent=su.mu_consistent, damping=su.damping, chassis_mass=su.chassis_mass,
                                         mass_add=su.mass_add, wheel_mass=su.wheel_mass)
    for row, want in ((A.S_MU_S, mu_s), (A.S_MU_D, mu_d), (A.S_DAMP, damp), (A.S_MASS, mass)):
        np.testing.assert_allclose(st[row, :n], want, rtol=6e-7, atol=0)
    assert 1.04 <= st[A.S_MASS, :n].min() and st[A.S_MASS, :n].max() <= 4.2 and abs(st[A.S_MASS, :n].mean() - 2.62) < 0.05


def _shard_check(make, n, K, act_scale=1.0, extra=()):
    """big batch vs two halves built independently (same seed, env_offset): bitwise equal after K steps"""
    big = make(n, 0)
    halves = [make(n // 2, r * (n // 2)) for r in range(2)]
    g = torch.Generator(device=DEV).manual_seed(3)
    acts = (torch.rand(K, n, 2, device=DEV, generator=g) * 2 - 1) * act_scale
    for e in [big] + halves:
        e.reset()
    resets = 0.0
    for k in range(K):
        ob, rb, tb, ub = [t.clone() for t in big.step(acts[k])]
        for r, h in enumerate(halves):
            sl = slice(r * (n // 2), (r + 1) * (n // 2))
            oh, rh, th, uh = h.step(acts[k, sl].contiguous())
            assert torch.equal(oh, ob[sl]), (k, r, "obs")
            assert torch.equal(rh, rb[sl]) and torch.equal(th, tb[sl]) and torch.equal(uh, ub[sl]), (k, r)
        resets += float((tb | ub).sum())
    torch.cuda.synchronize()
    for r, h in enumerate(halves):
        sl = slice(r * (n // 2), (r + 1) * (n // 2))
        assert torch.equal(h.state[:, : n // 2], big.state[:, sl]), r
        assert torch.equal(h.episode_len[: n // 2], big.episode_len[sl])
        for name in extra:
            assert torch.equal(getattr(h, name), getattr(big, name)), name
    # the shards' episode metrics add up to the big batch's (what the all-reduce computes)
    m = halves[0].metrics + halves[1].metrics
    assert abs(float(m[8]) - float(big.metrics[8])) < 0.5 and float(big.metrics[8]) == resets
    torch.testing.assert_close(m, big.metrics, rtol=1e-4, atol=1e-3)
    return resets


def test_drift_half_shards_equal_the_big_batch(lib):
    from wheeledlab_amd.core import DriftBatch
    from wheeledlab_amd.params import drift_params

    def make(n, off):
        p = drift_params()
        p.max_episode_length = 9          # time-outs (and their resets) inside the short run
        return DriftBatch(n, device=DEV, seed=21, env_offset=off, params=p)
    assert _shard_check(make, 1024, 24) > 1024


def test_elevation_half_shards_equal_the_big_batch(lib):
    from wheeledlab_amd.core import ElevBatch
    from wheeledlab_amd.params import elev_params

    def make(n, off):
        p = elev_params()
        p.max_episode_length = 5
        return ElevBatch(n, device=DEV, seed=22, env_offset=off, params=p)
    assert _shard_check(make, 512, 12) > 512


def test_visual_half_shards_equal_the_big_batch(lib):
    from wheeledlab_amd.core import VisualBatch
    from wheeledlab_amd.params import visual_params

    def make(n, off):
        p = visual_params()
        p.max_episode_length = 4
        return VisualBatch(n, device=DEV, seed=23, env_offset=off, params=p)
    assert _shard_check(make, 256, 9, extra=("trav_map",)) > 256


def test_rccl_world_of_one_runs_the_multi_gpu_code_path(lib):
    """torch.distributed "nccl" IS RCCL on ROCm: the process-group creation with device_id, dist.allreduce_metrics,
    max_over_ranks / ranks_agree and bench.py's async all-reduce + deferred join, on one GPU"""
    import torch.distributed as td

    from wheeledlab_amd import dist as D
    from wheeledlab_amd.core import DriftBatch
    if td.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device(DEV)
    torch.cuda.set_device(dev)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert td.get_backend() == "nccl" and D.world_size() == 1 and D.shard_offset(4096) == 0
        env = DriftBatch(4096, device=DEV, seed=42, env_offset=D.shard_offset(4096))
        env.reset()
        acts = torch.rand(128, 4096, 2, device=DEV) * 2 - 1
        total = torch.zeros_like(env.metrics)
        pending = None
        for it in range(3):   # bench.py's run(): rollout, read the metrics, async all-reduce, join one interval later
            env.rollout(acts)
            m = env.read_metrics(zero=True)
            if pending is not None:
                pending[0].wait()
                total.add_(pending[1])
            pending = (td.all_reduce(m, async_op=True), m)
        pending[0].wait()
        total.add_(pending[1])
        td.barrier()
        torch.cuda.synchronize()
        assert float(total[8]) > 0                           # episodes ended and were counted through the collective
        # the synchronous forms go through RCCL too when called on the raw API (dist.py short-circuits world 1)
        t = torch.full((16,), 2.0, device=DEV)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        assert torch.equal(t, torch.full((16,), 2.0, device=DEV))
        x = torch.tensor([3.0], device=DEV, dtype=torch.float64)
        td.all_reduce(x, op=td.ReduceOp.MAX)
        assert float(x) == 3.0
        assert D.ranks_agree(t) and D.max_over_ranks(1.5, device=DEV) == 1.5
        assert torch.equal(D.allreduce_metrics(total.clone()), total)
    finally:
        td.destroy_process_group()
