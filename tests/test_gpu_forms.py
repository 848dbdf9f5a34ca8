"""Every kernel INSTANTIATION the launchers pick from the batch size, run where a test can check it (`pytest -m gpu`).

The step / scan / camera launchers switch forms by size: the streaming drift step beyond 1.22 M envs (`sc1 nt` row stores,
csrc/wl_drift.hip launch_step), non-temporal observation rows of the height scan beyond 97 k envs and of the camera beyond
20.9 k (wl_elev.hip launch_elev_scan, wl_visual.hip launch_visual_obs); the height scan through LDS patches is a flag (round 6: with
the row-pair table the gather form is the faster one at every size and the default).
bench.py's `large_n_sweep` / `other_tasks_large_n` rows -- the 4 M-env row carries the north-star's HBM fraction -- are
measured on exactly those forms.  Here:

* `WlEnvBuffers.flags` forces each form at a size the oracle finishes in seconds: against the ORACLE (drift streaming form, MuSHR
  and F1Tenth), and bit for bit against the form the other tests already cover (all tasks);
* the sweep's TRUE sizes: 4 194 304 drift envs == four 1 048 576-env shards built with `env_offset` (bit-identical state,
  observations, rewards, flags after 3 steps; the shards run the cache-allocating form, the big batch the streaming one), with the
  batch's LAST 1024 envs (the highest row offsets: the 31-bit buffer range) against the oracle; 262 144 elevation envs == two
  shards; 65 536 visual envs == two shards.
"""
import numpy as np
import pytest
import torch

from oracle import drift_step as OS
from oracle import params as OP

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def A():
    from wheeledlab_amd import _abi
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    _abi.load()
    return _abi


def _drift(n, seed, flags=0, lanes=0, off=0, params=None):
    from wheeledlab_amd.core import DriftBatch
    env = DriftBatch(n, device=DEV, seed=seed, env_offset=off, params=params)
    env.set_lanes(lanes)
    env.set_flags(flags)
    env.reset()
    return env


# ---- drift: the streaming instantiation ------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["philox", "noise_tensor", "no_corruption"])
def test_streaming_drift_form_matches_oracle_single_steps(A, mode):
    from test_gpu_drift_parity import _single_step_parity
    n = 1024
    env = _drift(n, 5, flags=A.FLAG_STREAM)
    torch.cuda.synchronize()
    _single_step_parity(env, OP.drift_params(), n, 5, mode)


@pytest.mark.parametrize("mode", ["philox", "no_corruption"])
def test_streaming_f1tenth_form_matches_oracle_single_steps(A, mode):
    from test_gpu_drift_parity import _f1tenth_batch, _single_step_parity
    n = 1024
    env, flat = _f1tenth_batch(n, seed=5)
    env.set_flags(A.FLAG_STREAM)
    _single_step_parity(env, flat.params, n, 5, mode)


def test_flag_combinations_are_validated(A):
    env = _drift(256, 1)
    a = torch.zeros(256, 2, device=DEV)
    for bad in (A.FLAG_STREAM | A.FLAG_NO_STREAM, A.FLAG_SCAN_LDS | A.FLAG_SCAN_GATHER, 16, -1):
        env.set_flags(bad)
        with pytest.raises(A.WlError):
            env.step(a)
    env.set_flags(A.FLAG_STREAM)
    env.set_lanes(4)                    # the streaming form is a lane form
    with pytest.raises(A.WlError):
        env.step(a)
    env.set_lanes(0)
    env.step(a)
    torch.cuda.synchronize()


def _same_rollout(envs, n, K, seed=9, act_scale=1.0, extra_check=None):
    """the same K steps on every env of `envs` (same seed / actions): outputs and final state must be bit-identical"""
    g = torch.Generator(device=DEV).manual_seed(seed)
    acts = (torch.rand(K, n, 2, device=DEV, generator=g) * 2 - 1) * act_scale
    resets = 0
    for k in range(K):
        ref = None
        for i, e in enumerate(envs):
            out = [t.clone() for t in e.step(acts[k])]
            if ref is None:
                ref = out
                resets += int((out[2] | out[3]).sum())
            else:
                for name, x, y in zip(("obs", "reward", "terminated", "truncated"), out, ref):
                    assert torch.equal(x, y), (k, i, name, float((x.float() - y.float()).abs().max()))
    torch.cuda.synchronize()
    for e in envs[1:]:
        assert torch.equal(e.state, envs[0].state) and torch.equal(e.episode_len, envs[0].episode_len)
        torch.testing.assert_close(e.metrics, envs[0].metrics, rtol=1e-5, atol=1e-3)
    return resets


def test_streaming_drift_form_is_bit_identical_to_the_lane_forms(A):
    """lanes = 2 (scalar wheel loop, cache-allocating stores) is the same arithmetic as the streaming form: bit-identical; a ragged
    env count (tail wavefront, tail block) and enough short episodes that resets / pushes / time-outs all occur"""
    from wheeledlab_amd.params import drift_params
    n = 5000 + 37

    def make(flags, lanes):
        p = drift_params()
        p.max_episode_length = 11
        return _drift(n, 13, flags, lanes, params=p)
    envs = [make(0, 2), make(A.FLAG_STREAM, 0), make(A.FLAG_STREAM, 2), make(A.FLAG_NO_STREAM, 2)]
    assert _same_rollout(envs, n, 30) > n


# ---- elevation: height scan forms --------------------------------------------------------------------------------------------
def _elev(n, seed, flags=0, lanes=0, off=0, params=None, hf=None):
    from wheeledlab_amd.core import ElevBatch
    env = ElevBatch(n, device=DEV, seed=seed, env_offset=off, params=params, heightfield=hf)
    env.set_lanes(lanes)
    env.set_flags(flags)
    env.reset()
    return env


@pytest.mark.parametrize("z_scale", [None, 0.005])
def test_height_scan_forms_are_bit_identical(A, z_scale):
    """gather / LDS-patch x cache-allocating / non-temporal: the four scan instantiations write the same 676 values per env --
    cars anywhere on the terrain at any yaw, incl. on and beyond its border (rays that miss: +inf clipped to 10; a patch origin
    clamped to the grid) and tilted.  z_scale 0.005: the terrain's codes at IsaacLab's default vertical_scale (not a power of two:
    the decode's product is rounded; every form must round it the same way) -- and the scan against the oracle on those codes."""
    n = 3000 + 11
    hf = None
    if z_scale is not None:
        from oracle import elev_step as OE
        from oracle import heightfield as OH
        from tests.depth_cases import on_lattice
        hf = on_lattice(OH.make_terrain(), z_scale)
    env = _elev(n, 31, flags=A.FLAG_SCAN_GATHER | A.FLAG_NO_STREAM, hf=None if hf is None else hf + (z_scale,))
    if hf is not None:
        assert env.hf.z_scale == z_scale and torch.equal(env.height.cpu(), torch.from_numpy(hf[0]))        # decoded = the oracle's grid
    g = torch.Generator(device=DEV).manual_seed(2)
    st = env.state
    half = 20.0
    st[0, :n] = (torch.rand(n, device=DEV, generator=g) * 2 - 1) * (half + 1.5)     # x: some cars past the border
    st[1, :n] = (torch.rand(n, device=DEV, generator=g) * 2 - 1) * (half + 1.5)
    # a band of cars exactly on the border lines and corners
    st[0, :64] = half - 0.01 * torch.arange(64, device=DEV)
    st[1, 64:128] = -half + 0.013 * torch.arange(64, device=DEV)
    st[0, 128:160], st[1, 128:160] = half - 0.02, half - 0.03
    q = torch.randn(4, n, device=DEV, generator=g)
    q[1:3] *= 0.15                                                                  # mostly yaw, some roll / pitch
    st[3:7, :n] = q / q.norm(dim=0, keepdim=True)
    st[3:7, 160:192] = torch.tensor([[0.92388, 0.0, 0.0, 0.38268]], device=DEV).T    # yaw 45 deg: the widest bounding box
    ref = env.observe().clone()
    torch.cuda.synchronize()
    assert torch.isfinite(ref).all() and (ref[:, 13:].abs() <= 10.0).all()
    assert (ref[:, 13:] == 10.0).any() and (ref[:, 13:].abs() < 5.0).any()           # misses and hits both present
    if hf is not None:
        want = OE.height_map(OE.elev_params(), env.state[:, :n].cpu().numpy(), hf)
        d = np.abs(ref[:, 13:].cpu().numpy() - want)
        # a ray within 1e-4 cell of a cell line or the border may fall on the other side in the other arithmetic (counted)
        assert (d > 2e-5).mean() < 2e-4 and np.median(d) < 1e-6, (float((d > 2e-5).mean()), float(d.max()))
    for flags in (A.FLAG_SCAN_GATHER | A.FLAG_STREAM, A.FLAG_SCAN_LDS | A.FLAG_NO_STREAM, A.FLAG_SCAN_LDS | A.FLAG_STREAM,
                  A.FLAG_SCAN_LDS, A.FLAG_STREAM):
        env.set_flags(flags)
        env.obs.fill_(-77.0)
        got = env.observe()
        torch.cuda.synchronize()
        bad = (got != ref)
        assert not bad.any(), (flags, int(bad.sum()), bad.nonzero()[:5].tolist(), float((got - ref).abs().max()))


def test_a_field_the_lds_patch_cannot_take_falls_back_to_the_gather_scan(A):
    """the LDS form stages 16-byte words from 4-byte aligned addresses: a field with an ODD row pitch (or narrower than a patch row)
    is scanned by the gather form whatever the flags ask for -- same observation rows, and right against the oracle"""
    from oracle import elev_step as OE
    from oracle import heightfield as OH
    full = OH.make_terrain()
    for ny, nx in ((613, 349), (200, 90)):
        hf = (np.ascontiguousarray(full[0][:ny, :nx]), full[1], full[2], full[3])
        n = 777
        env = _elev(n, 23, flags=A.FLAG_SCAN_GATHER | A.FLAG_NO_STREAM, hf=hf)
        g = torch.Generator(device=DEV).manual_seed(3)
        st = env.state
        st[0, :n] = float(hf[1]) + torch.rand(n, device=DEV, generator=g) * (nx - 1) * 0.05
        st[1, :n] = float(hf[2]) + torch.rand(n, device=DEV, generator=g) * (ny - 1) * 0.05
        q = torch.randn(4, n, device=DEV, generator=g)
        q[1:3] *= 0.1
        st[3:7, :n] = q / q.norm(dim=0, keepdim=True)
        ref = env.observe().clone()
        for flags in (A.FLAG_SCAN_LDS, A.FLAG_SCAN_LDS | A.FLAG_STREAM):
            env.set_flags(flags)
            env.obs.fill_(-77.0)
            assert torch.equal(env.observe(), ref), (nx, flags)
        want = OE.height_map(OE.elev_params(), env.state[:, :n].cpu().numpy(), hf)
        d = np.abs(ref[:, 13:].cpu().numpy() - want)
        assert (d > 2e-5).mean() < 2e-4 and (np.abs(want) < 5).any() and (want == 10.0).any(), (nx, float(d.max()))


def test_elevation_lane_form_steps_are_bit_identical_across_scan_forms(A):
    """the two-launch lane form (what runs beyond 32 768 envs) with the scan through LDS / with streaming rows == the same form
    with gathers, through resets and command resamples"""
    from wheeledlab_amd.params import elev_params
    n = 2048 + 19

    def make(flags):
        p = elev_params()
        p.max_episode_length = 6
        return _elev(n, 17, flags, 1, params=p)
    envs = [make(A.FLAG_SCAN_GATHER | A.FLAG_NO_STREAM), make(A.FLAG_SCAN_LDS | A.FLAG_STREAM), make(A.FLAG_SCAN_LDS), make(A.FLAG_STREAM)]
    assert _same_rollout(envs, n, 14) > n


# ---- visual: camera rows ------------------------------------------------------------------------------------------------------
def _visual(n, seed, flags=0, lanes=0, off=0, params=None):
    from wheeledlab_amd.core import VisualBatch
    env = VisualBatch(n, device=DEV, seed=seed, env_offset=off, params=params)
    env.set_lanes(lanes)
    env.set_flags(flags)
    env.sample_augmentation(torch.Generator().manual_seed(4))
    env.reset()
    return env


@pytest.mark.parametrize("lanes", [1, 4])
def test_streaming_camera_rows_are_bit_identical(A, lanes):
    from wheeledlab_amd.params import visual_params
    n = 700 + 3

    def make(flags):
        p = visual_params()
        p.max_episode_length = 5
        return _visual(n, 19, flags, lanes, params=p)
    envs = [make(A.FLAG_NO_STREAM), make(A.FLAG_STREAM)]
    assert _same_rollout(envs, n, 11) > n


# ---- the sweep's true sizes ---------------------------------------------------------------------------------------------------
def _shards_equal_big(make, n, n_shards, K, seed=5):
    """big batch (default form selection: whatever bench.py's sweep runs at this size) vs `n_shards` shards built independently
    with env_offset: bit-identical outputs every step and state at the end"""
    m = n // n_shards
    big = make(n, 0)
    shards = [make(m, r * m) for r in range(n_shards)]
    g = torch.Generator(device=DEV).manual_seed(seed)
    done = 0
    for k in range(K):
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        ob, rb, tb, ub = big.step(a)
        done += int((tb | ub).sum())
        for r, s in enumerate(shards):
            sl = slice(r * m, (r + 1) * m)
            o, rw, t, u = s.step(a[sl].contiguous())
            assert torch.equal(o, ob[sl]), (k, r, "obs")
            assert torch.equal(rw, rb[sl]) and torch.equal(t, tb[sl]) and torch.equal(u, ub[sl]), (k, r)
    torch.cuda.synchronize()
    for r, s in enumerate(shards):
        sl = slice(r * m, (r + 1) * m)
        assert torch.equal(s.state[:, :m], big.state[:, sl]) and torch.equal(s.episode_len[:m], big.episode_len[sl]), r
    tot = sum(s.metrics for s in shards)
    assert float(tot[8]) == float(big.metrics[8]) == done
    return big, done


def test_drift_4m_envs_equal_four_1m_shards_and_the_oracle(A):
    """large_n_sweep's 4 194 304-env row (streaming form by size) and its 1 048 576-env row (cache-allocating lane form)"""
    from wheeledlab_amd.params import drift_params
    n = 4194304

    def make(m, off):
        p = drift_params()
        p.max_episode_length = 3           # time-outs + resets inside the three steps
        return _drift(m, 42, 0, 0, off, params=p)
    big, done = _shards_equal_big(make, n, 4, 3)
    assert done >= n
    # the batch's last 1024 envs (row offsets near the end of the 688 MB matrix) against the oracle, one step from the device state
    k0 = 1024
    cols = slice(n - k0, n)
    p = OP.drift_params()
    p.max_episode_length = 3
    st = big.state[:, cols].cpu().numpy().copy()
    ep = big.episode_len[cols].cpu().numpy().copy()
    a = (torch.rand(n, 2, device=DEV) * 2 - 1)
    obs, rew, term, trunc = big.step(a)
    torch.cuda.synchronize()
    o_obs, o_rew, o_term, o_trunc, _ = OS.step(p, st, ep, big.ref_table.cpu().numpy(), a[cols].cpu().numpy(), 42, 3, env_offset=n - k0)
    got = big.state[:, cols].cpu().numpy()
    ok = term[cols].cpu().numpy().astype(bool) == o_term
    assert (~ok).sum() <= 1
    np.testing.assert_array_equal(trunc[cols].cpu().numpy().astype(bool), o_trunc)
    np.testing.assert_allclose(got[:23][:, ok], st[:23][:, ok], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rew[cols].cpu().numpy()[ok], o_rew[ok], rtol=2e-3, atol=2e-3)
    d = np.abs(obs[cols].cpu().numpy() - o_obs)[ok]
    d[:, 3:6] = np.minimum(d[:, 3:6], np.abs(2 * np.pi - d[:, 3:6]))
    assert d.max() < 1e-3


def test_drift_65536_envs_equal_two_shards(A):
    """large_n_sweep's first row: the lane form with packed axles (the 32 768-env shards are told to run the same form: by size they
    would take the quad form, which orders its arithmetic differently -- equal to the oracle within tolerance, not to the lane form
    bit for bit)"""
    from wheeledlab_amd.params import drift_params

    def make(m, off):
        p = drift_params()
        p.max_episode_length = 4
        return _drift(m, 42, 0, 0 if m == 65536 else 1, off, params=p)
    _, done = _shards_equal_big(make, 65536, 2, 6)
    assert done >= 65536


def test_elevation_262144_envs_equal_two_shards(A):
    """other_tasks_large_n: lane-form step + the large-batch height scan (LDS patches, non-temporal rows)"""
    from wheeledlab_amd.params import elev_params

    def make(m, off):
        p = elev_params()
        p.max_episode_length = 2
        return _elev(m, 42, 0, 0, off, params=p)
    _, done = _shards_equal_big(make, 262144, 2, 3)
    assert done >= 262144


def test_visual_65536_envs_equal_two_shards(A):
    """other_tasks_large_n: lane-form step + the streaming camera (shards forced to the lane form, see the drift test)"""
    from wheeledlab_amd.params import visual_params

    def make(m, off):
        p = visual_params()
        p.max_episode_length = 2
        return _visual(m, 42, 0, 0 if m == 65536 else 1, off, params=p)
    _, done = _shards_equal_big(make, 65536, 2, 3)
    assert done >= 65536
