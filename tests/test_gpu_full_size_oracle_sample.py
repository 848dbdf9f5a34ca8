"""The BASELINE-size batches (4096 envs: configs 3 and 5 of BASELINE.json, and the visual task) against the oracle DIRECTLY: every step a
contiguous 256-env window of the 4096-env batch -- a different window each step, RNG keyed by the global env id through the oracle's
`env_offset` -- is stepped by the oracle from the device's pre-step rows and held to the single-step tests' bars (state at 5e-4 with
the discontinuity predicate of tests/parity_predicates.py, termination decisions exact up to boundary ties, rewards).  Until round 5 the
4096-env batches were held by properties and shard equalities only; the step-vs-oracle tests ran at 512 / 256 / 128 envs (the same kernel
instantiations: the launchers pick the form by env count, and 4096 <= 8192 takes the same fused / quad kernels)."""
import numpy as np
import pytest
import torch

from oracle import elev_step as OE
from oracle import heightfield as OH
from oracle import visual_step as OV
from tests import parity_predicates as PRED

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N, W, STEPS = 4096, 256, 10


def _window(k):
    return (k * 1361 + 517) % (N - W)          # a different, unaligned window each step (also across block boundaries)


def _actions(rng):
    a = rng.uniform(-1.2, 1.2, (N, 2)).astype(np.float32)
    a[:, 0] = np.abs(a[:, 0]) * 0.7 + 0.2
    return a


def _compare(k, got, st, probe, term, o_term, trunc, o_trunc, rew, o_rew, rew_atol):
    np.testing.assert_array_equal(trunc, o_trunc)
    bad = term != o_term
    assert bad.sum() <= 1, k                                  # a termination decision on an fp32 boundary tie
    ok, n_ex = PRED.check_state(got, st, probe, W, ~bad, where=f"step {k}")
    assert PRED.state_error(got, st, W)[:, ok].max() <= 1.0
    return ok, n_ex


def test_elevation_4096_window_matches_the_oracle():
    from wheeledlab_amd.core import ElevBatch
    hf = OH.make_terrain()
    env = ElevBatch(N, device=DEV, seed=7, heightfield=hf)
    env.reset()
    env.episode_len[:N] = torch.randint(150, 199, (N,), device=DEV, dtype=torch.int32, generator=torch.Generator(device=DEV).manual_seed(1))
    p = OE.elev_params()
    rng = np.random.RandomState(0)
    excused = resets = 0
    for k in range(STEPS):
        off = _window(k)
        st = env.state[:, off:off + W].cpu().numpy().copy()
        ep = env.episode_len[off:off + W].cpu().numpy().copy()
        a = _actions(rng)
        obs, rew, term, trunc = env.step(torch.from_numpy(a).to(DEV))
        torch.cuda.synchronize()
        probe = {}
        o_obs, o_rew, o_term, o_trunc, _ = OE.step(p, st, ep, hf, a[off:off + W], 7, k, env_offset=off, probe=probe)
        got = env.state[:, off:off + W].cpu().numpy()
        sl = slice(off, off + W)
        ok, n_ex = _compare(k, got, st, probe, term[sl].cpu().numpy(), o_term, trunc[sl].cpu().numpy(), o_trunc, rew, o_rew, 5e-2)
        excused += n_ex
        resets += int((o_term | o_trunc).sum())
        np.testing.assert_allclose(rew[sl].cpu().numpy()[ok], o_rew[ok], rtol=2e-3, atol=5e-2)
        d = np.abs(obs[sl].cpu().numpy() - o_obs)[ok]
        d[:, 2:5] = np.minimum(d[:, 2:5], np.abs(2 * np.pi - d[:, 2:5]))
        assert d[:, :13].max() < 3e-3 and (d[:, 13:] > 2e-3).sum() <= 4, k
        np.testing.assert_array_equal(env.episode_len[sl].cpu().numpy()[ok], ep[ok])
    assert excused <= 3 and resets > 0, (excused, resets)          # time-outs (and their in-step resets) inside the windows


def test_visual_4096_window_matches_the_oracle(golden):
    from wheeledlab_amd.core import VisualBatch
    g = golden("visual_trav")
    trav = np.unpackbits(g["full_map_packed"])[: 500 * 500].reshape(500, 500).astype(bool)
    env = VisualBatch(N, device=DEV, seed=7, trav_map=trav)
    env.reset()
    env.episode_len[:N] = torch.randint(30, 49, (N,), device=DEV, dtype=torch.int32, generator=torch.Generator(device=DEV).manual_seed(1))
    p = OV.visual_params()
    cells = OV.spawn_cells(trav)
    rng = np.random.RandomState(0)
    excused = resets = 0
    for k in range(STEPS):
        off = _window(k)
        st = env.state[:, off:off + W].cpu().numpy().copy()
        ep = env.episode_len[off:off + W].cpu().numpy().copy()
        a = _actions(rng)
        obs, rew, term, trunc = env.step(torch.from_numpy(a).to(DEV))
        torch.cuda.synchronize()
        probe = {}
        o_obs, o_rew, o_term, o_trunc, _ = OV.step(p, st, ep, trav, cells, a[off:off + W], 7, k, env_offset=off, probe=probe)
        got = env.state[:, off:off + W].cpu().numpy()
        sl = slice(off, off + W)
        ok, n_ex = _compare(k, got, st, probe, term[sl].cpu().numpy(), o_term, trunc[sl].cpu().numpy(), o_trunc, rew, o_rew, 2e-3)
        excused += n_ex
        resets += int((o_term | o_trunc).sum())
        cell_flip = np.abs(rew[sl].cpu().numpy() - o_rew) > 0.5          # +-1 traversability flips exactly on a cell edge
        assert (cell_flip & ok).sum() <= 1
        sel = ok & ~cell_flip
        np.testing.assert_allclose(rew[sl].cpu().numpy()[sel], o_rew[sel], rtol=2e-3, atol=2e-3)
        d = np.abs(obs[sl].cpu().numpy() - o_obs)[sel]
        assert d[:, 3200:].max() < 3e-3 and (d[:, :3200] > 2e-3).mean() < 5e-3, k
    assert excused <= 3 and resets > 0, (excused, resets)


def test_visual_depth_task_4096_window_matches_the_oracle():
    from wheeledlab_amd.core import VisualDepthBatch
    hf = OH.make_terrain()
    env = VisualDepthBatch(N, device=DEV, seed=7, heightfield=hf, max_depth=20.0)
    env.reset()
    env.episode_len[:N] = torch.randint(30, 49, (N,), device=DEV, dtype=torch.int32, generator=torch.Generator(device=DEV).manual_seed(1))
    trav = env.trav_map.cpu().numpy().astype(bool)
    cells = OV.spawn_cells(trav)
    p = OV.visual_params()
    p.map_rows, p.map_cols = int(env._map.rows), int(env._map.cols)
    rng = np.random.RandomState(0)
    excused = resets = 0
    for k in range(STEPS):
        off = _window(k)
        st = env.state[:, off:off + W].cpu().numpy().copy()
        ep = env.episode_len[off:off + W].cpu().numpy().copy()
        a = _actions(rng)
        obs, rew, term, trunc = env.step(torch.from_numpy(a).to(DEV))
        torch.cuda.synchronize()
        probe = {}
        o_obs, o_rew, o_term, o_trunc, _ = OV.step(p, st, ep, trav, cells, a[off:off + W], 7, k, env_offset=off, hf=hf, max_depth=20.0,
                                                   probe=probe)
        got = env.state[:, off:off + W].cpu().numpy()
        sl = slice(off, off + W)
        ok, n_ex = _compare(k, got, st, probe, term[sl].cpu().numpy(), o_term, trunc[sl].cpu().numpy(), o_trunc, rew, o_rew, 3e-3)
        excused += n_ex
        resets += int((o_term | o_trunc).sum())
        cell_flip = np.abs(rew[sl].cpu().numpy() - o_rew) > 0.5
        assert (cell_flip & ok).sum() <= 1
        sel = ok & ~cell_flip
        np.testing.assert_allclose(rew[sl].cpu().numpy()[sel], o_rew[sel], rtol=2e-3, atol=3e-3)
        assert np.abs(obs[sl].cpu().numpy()[sel, 4800:] - o_obs[sel, 4800:]).max() < 3e-3, k      # proprio; the image: tests/test_gpu_visual_depth_task.py
    assert excused <= 3 and resets > 0, (excused, resets)
