"""Elevation task on the GPU: mdp terms vs the reference's golden outputs, fused step + 689-dim observation vs the
oracle, full-size invariants.  Everything through the C ABI."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests import parity_predicates as PRED
from oracle import elev_step as OS
from oracle import heightfield as OH

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def soa(a, stride):
    a = np.atleast_2d(a.T).T if a.ndim == 1 else a
    n, k = a.shape
    t = torch.zeros(k, stride, dtype=torch.float32)
    t[:, :n] = torch.from_numpy(np.ascontiguousarray(a.T))
    return t.to(DEV)


def test_elev_mdp_kernel_matches_reference_golden(golden):
    from wheeledlab_amd import _abi as A
    from wheeledlab_amd import params as PP
    lib = A.load()
    g = golden("elevation_mdp")
    n = g["pos"].shape[0]
    stride = ((n + 63) // 64) * 64
    p = PP.elev_params()
    K = g["ray_hits_z"].shape[1]
    ins = [soa(g[k], stride) for k in ("pos", "quat", "lin_vel_b", "lin_vel_w")]
    wheel = soa(g["joint_vel"][:, 2:6], stride)
    cmd = soa(g["command"][:, :2], stride)
    sens = torch.from_numpy(g["sensor_pos_w"][:, 2].copy()).to(DEV)
    hits = soa(g["ray_hits_z"], stride)
    terms = torch.zeros(4, stride, device=DEV)
    flags = torch.zeros(4, stride, dtype=torch.uint8, device=DEV)
    goal = torch.zeros(2, stride, device=DEV)
    hmap = torch.zeros(K, stride, device=DEV)
    rc = lib.wl_elev_mdp(C.byref(p), n, stride, *[t.data_ptr() for t in ins], wheel.data_ptr(), cmd.data_ptr(), None, K,
                         sens.data_ptr(), hits.data_ptr(), terms.data_ptr(), flags.data_ptr(), goal.data_ptr(),
                         hmap.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    T, Fg = terms.cpu().numpy()[:, :n], flags.cpu().numpy()[:, :n].astype(bool)
    np.testing.assert_allclose(T[0], g["goal_progress_rate"], rtol=1e-5, atol=1e-5, equal_nan=True)
    np.testing.assert_allclose(T[1], g["higher_elevation"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(T[2] > 0.5, g["is_falling_penalty"])
    np.testing.assert_array_equal(Fg[1], g["stuck"])
    np.testing.assert_array_equal(Fg[3], g["close_to_goal"])
    np.testing.assert_array_equal(Fg[0], g["pos"][:, 2] < 0.15)
    away = np.abs(g["upright_penalty"]) > 1e-2          # R33 < cos(60 deg) vs acos(R33) > 60 deg: equal away from the tie
    sel = away | (g["upright_penalty"] == 0)
    np.testing.assert_array_equal(Fg[2][sel], g["upright_bool"][sel])
    np.testing.assert_allclose(goal.cpu().numpy()[:, :n].T, g["goal_relative_xyz"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(hmap.cpu().numpy()[:, :n].T, g["world_height_map"], rtol=1e-6, atol=4e-6)


def _fresh(n, seed=3, z_scale=None):
    """z_scale: the bench terrain re-quantised to another vertical scale (not a power of two: the decode is a ROUNDED product, the
    same on the device and in the oracle's decoded grid), passed as IsaacLab passes its terrains: codes + scale"""
    from wheeledlab_amd.core import ElevBatch
    hf = OH.make_terrain()
    if z_scale is not None:
        codes = OH.quantize(hf[0], z_scale)
        hf = (OH.decode(codes, z_scale),) + hf[1:]
        env = ElevBatch(n, device=DEV, seed=seed, heightfield=(codes,) + hf[1:] + (z_scale,))
        assert env.hf.z_scale == z_scale and torch.equal(env.height.cpu(), torch.from_numpy(hf[0]))
    else:
        env = ElevBatch(n, device=DEV, seed=seed, heightfield=hf)
    env.reset()
    torch.cuda.synchronize()
    return env, hf


def test_elev_reset_and_observation_match_oracle():
    env, hf = _fresh(200, seed=11)
    st = env.state.cpu().numpy()
    p = OS.elev_params()
    o = np.zeros_like(st)
    o[3] = 1
    o[23:27] = st[23:27]
    ep = np.ones(st.shape[1], np.int32)
    OS.reset_envs(p, o, ep, hf, np.arange(200), 11, 0)
    OS.update_command(o)
    np.testing.assert_allclose(st[:, :200], o[:, :200], rtol=2e-6, atol=2e-5)
    obs = env.observe().cpu().numpy()
    want = OS.observe(p, st[:, :200].copy(), hf)
    d = np.abs(obs - want)
    d[:, 2:5] = np.minimum(d[:, 2:5], np.abs(2 * np.pi - d[:, 2:5]))
    assert d[:, :13].max() < 2e-5
    assert d[:, 13:].max() < 2e-5                      # 26 x 26 height map: bilinear gathers agree to fp32 rounding
    assert obs.shape == (200, 689)


@pytest.mark.parametrize("lanes,z_scale", [(4, None), (1, None), (4, 1e-4), (1, 1e-4)])
def test_elev_fused_step_matches_oracle_single_steps(lanes, z_scale):
    n = 512
    env, hf = _fresh(n, seed=5, z_scale=z_scale)
    env.set_lanes(lanes)
    p = OS.elev_params()
    rng = np.random.RandomState(0)
    flips = excused = reach = 0
    for k in range(24):
        st = env.state.cpu().numpy().copy()
        ep = env.episode_len.cpu().numpy().copy()
        if k == 12:
            ep[: n // 4] = 199
            st[40, n // 4: n // 2] = 0.05                 # command timers about to expire -> resample path
            env.episode_len.copy_(torch.from_numpy(ep))
            env.state.copy_(torch.from_numpy(st))
        a = rng.uniform(-1.2, 1.2, (n, 2)).astype(np.float32)
        a[:, 0] = np.abs(a[:, 0]) * 0.6 + 0.2
        met0 = env.metrics.cpu().numpy().astype(np.float64)
        obs, rew, term, trunc = env.step(torch.from_numpy(a).to(DEV))
        torch.cuda.synchronize()
        met = np.zeros(16)
        probe = {}
        o_obs, o_rew, o_term, o_trunc, info = OS.step(p, st, ep, hf, a, 5, k, met, probe=probe)
        got = env.state.cpu().numpy()
        np.testing.assert_array_equal(trunc.cpu().numpy(), o_trunc)
        bad = term.cpu().numpy() != o_term
        flips += int(bad.sum())
        ok = ~bad
        # 10 sub-steps over a bilinear heightfield: 5e-4 abs/rel on the dynamic state.  An env may miss it only if the ORACLE's own step
        # shows the cause -- a wheel within reach of making / breaking contact or of a cell line (tests/parity_predicates.py) -- is then
        # held to a loose bound, and must re-converge (each step restarts from the device state)
        ok, n_ex = PRED.check_state(got, st, probe, n, ok, where=f"step {k}")
        excused += n_ex
        reach += int(PRED.explainable(probe, n).sum())
        assert PRED.state_error(got, st, n)[:, ok].max() <= 1.0, k
        np.testing.assert_allclose(got[35:41, :n][:, ok], st[35:41, :n][:, ok], rtol=5e-4, atol=2e-3)
        np.testing.assert_allclose(rew.cpu().numpy()[ok], o_rew[ok], rtol=2e-3, atol=5e-2)   # weights 5000*0.1 amplify z
        d = np.abs(obs.cpu().numpy() - o_obs)[ok]
        d[:, 2:5] = np.minimum(d[:, 2:5], np.abs(2 * np.pi - d[:, 2:5]))
        assert d[:, :13].max() < 3e-3, (k, d[:, :13].max())
        # height scan: a ray that lands within rounding of the terrain's edge hits in one build and misses in the other
        # (the clipped reading jumps by ~10): a handful of rays per step at most, every other reading agrees
        scan_bad = d[:, 13:] > 2e-3
        assert scan_bad.sum() <= 4 and scan_bad.any(1).sum() <= 2, (k, int(scan_bad.sum()), float(d[:, 13:].max()))
        if not bad.any():
            dm = env.metrics.cpu().numpy().astype(np.float64) - met0
            np.testing.assert_allclose(dm[8:16], met[8:16], atol=1e-3)
    assert flips <= 3
    # measured (round 6, all four variants): NO env needs the excuse -- 0 of 12 288 env-steps, spawn drops included (until round 5, with
    # 20 explicit sub-steps, up to 1 % of the envs per step did, by a count).  A handful stays allowed for other seeds / compilers, each
    # one explained by the predicate.
    assert excused <= 3, (excused, reach)


@pytest.mark.parametrize("n", [4096])
def test_elev_full_size_properties(n):
    env, hf = _fresh(n, seed=1)
    g = torch.Generator(device=DEV).manual_seed(0)
    resets = 0
    for k in range(220):
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        obs, rew, term, trunc = env.step(a)
        resets += int((term | trunc).sum())
    torch.cuda.synchronize()
    st = env.state[:, :n]
    assert torch.isfinite(st).all() and torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert ((st[3:7] ** 2).sum(0).sqrt() - 1).abs().max() < 1e-5
    assert (obs[:, 13:].abs() <= 10).all() and (obs[:, 5:11].abs() <= 10).all() and (obs[:, 11:13].abs() <= 1).all()
    assert (env.episode_len[:n] < 200).all()
    m = env.metrics.cpu().numpy()
    assert m[8] == resets and m[14] == 0
    # height map == terrain under the car: the centre rays bracket the terrain height at the root position
    z_t, _, inside = OH.sample(*hf, st[0].cpu().numpy(), st[1].cpu().numpy())
    centre = obs[:, 13 + 12 * 26 + 12: 13 + 12 * 26 + 14].mean(1).cpu().numpy() + 0.106
    sel = inside & (np.abs(st[0].cpu().numpy()) < 18) & (np.abs(st[1].cpu().numpy()) < 18)
    assert np.abs(centre[sel] - z_t[sel]).max() < 0.08
    # cars ride on the terrain (root within a few cm of the surface unless airborne right after a reset)
    riding = (env.episode_len[:n] > 5).cpu().numpy() & sel
    assert np.abs(st[2].cpu().numpy()[riding] - z_t[riding]).max() < 0.12


@pytest.mark.parametrize("n,K,slots", [(4096, 24, 1), (1000, 7, 1), (256, 5, 4)])
def test_persistent_rollout_equals_stepping(n, K, slots):
    """wl_elev_rollout_persistent (K steps in one launch: state in registers, the scan of step k overlapped with the
    integration of step k + 1) against wl_elev_rollout (a launch per step): every output row and the final state bit for bit,
    the episode metrics to summation order -- incl. in-rollout resets, a partly filled last block (n = 1000) and the metric ring"""
    from wheeledlab_amd.core import ElevBatch
    ea, eb = ElevBatch(n, device=DEV, seed=21, metrics_slots=slots), ElevBatch(n, device=DEV, seed=21, metrics_slots=1)
    g = torch.Generator(device=DEV).manual_seed(4)
    for e in (ea, eb):
        e.reset()
        e.episode_len[:n] = torch.randint(0, 248, (n,), device=DEV, dtype=torch.int32, generator=torch.Generator(device=DEV).manual_seed(1))
    a = torch.rand(K, n, 2, device=DEV, generator=g) * 2 - 1
    outs = []
    for e, persistent in ((ea, True), (eb, False)):
        obs = torch.zeros(K, n, e.OBS_DIM, device=DEV)
        rew = torch.zeros(K, n, device=DEV)
        term = torch.zeros(K, n, dtype=torch.bool, device=DEV)
        trunc = torch.zeros(K, n, dtype=torch.bool, device=DEV)
        dones = torch.zeros(K, n, dtype=torch.long, device=DEV)
        e.rollout(a, obs, rew, term, trunc, dones_out=dones, persistent=persistent)
        outs.append((obs, rew, term, trunc, dones))
    torch.cuda.synchronize()
    for x, y, name in zip(outs[0], outs[1], ("obs", "reward", "terminated", "truncated", "dones")):
        assert torch.equal(x, y), name
    assert torch.equal(ea.state, eb.state) and torch.equal(ea.episode_len, eb.episode_len) and ea.step_count == eb.step_count == K
    assert int(outs[0][4].sum()) > 0
    # episode metrics: the same events, summed in a different order (K steps folded in LDS before they reach the shards) --
    # counts exact, float sums to rounding; with a ring the one launch books all K steps into the first step's slot
    ma, mb = ea.metrics_raw.sum((0, 1)), eb.metrics_raw.sum((0, 1))
    if slots > 1:
        assert float(ea.metrics_raw[1:].abs().sum()) == 0.0
    torch.testing.assert_close(ma, mb, rtol=1e-5, atol=1e-3)
    assert torch.equal(ma[8:12], mb[8:12]) and float(ma[8]) == float(outs[0][4].sum())      # resets, time-outs, first two terminations


@pytest.mark.parametrize("terrain", ["plane", "tilted", "bench"])
@pytest.mark.parametrize("lanes", [4, 1])
def test_settled_cars_need_no_contact_excuse(lanes, terrain):
    """Companion of test_elev_fused_step_matches_oracle_single_steps: that test excuses up to 1 % of envs per step as contact
    make / break discontinuities (the spawn drop; on the rough synthetic terrain also a wheel unloading over a crest -- the
    suspension's static deflection is 2.8 mm).  Here nothing makes or breaks contact: the terrain is a tilted plane with a faint
    long swell (all four wheels stay loaded; normals still vary), every termination is switched off on BOTH sides, the cars
    settle for 18 steps and then crawl -- the excused set must be EMPTY for 24 steps, in both forms of the kernel.
    terrain "bench" (round 4): the same on the synthetic 800 x 800 terrain bench.py and the step test run on (hills, ramps up to
    plateaus, the 4 cm undulation): settled, gently driven cars keep all four wheels loaded there too.
    Round 5 (16-bit height codes): a CURVED surface on the code lattice carries +-0.06 mm of rounding per grid point -- slope noise of
    0.24 % between neighbouring cells, a kink in the ground normal at every cell line -- so "tilted" (plane + swell) now behaves like
    "bench": now and then ONE car's wheel sits on a cell line in one arithmetic and beside it in the other (measured: 1 env of 512 in
    1 - 2 of the 24 steps, 2 - 4 x the bound).  The EMPTY excuse set is held on terrain "plane": a tilted plane that IS on the lattice
    (41 and 20 codes of rise per cell: 10 % and 4.9 % grade), all four wheels loaded, the body rolled and pitched against gravity."""
    from wheeledlab_amd.core import ElevBatch
    n = 512
    xs = (np.arange(800) * 0.05 - 20.0).astype(np.float64)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    from tests.depth_cases import on_lattice
    # (heights within +-3.3 m: codes of 2^-13 m.  Lifted by 2.5 m -- as until round 4 -- the 5.7 m range takes codes of 2^-12 m, and
    # the coarser lattice's slope noise, 0.5 % per cell, triples the host-vs-oracle step error: 1.6e-4 against 5.9e-5 on CPU)
    hf = on_lattice(((0.19 + 0.10 * X + 0.05 * Y + 0.02 * np.sin(0.5 * X) * np.cos(0.4 * Y)).astype(np.float32), np.float32(-20.0),
                     np.float32(-20.0), np.float32(0.05)))
    if terrain == "bench":
        hf = OH.make_terrain()
    if terrain == "plane":
        ii, jj = np.meshgrid(np.arange(800) - 400, np.arange(800) - 400, indexing="xy")
        codes = (1556 + 41 * ii + 20 * jj).astype(np.int16)
        hf = (OH.decode(codes), np.float32(-20.0), np.float32(-20.0), np.float32(0.05))
        assert np.abs(np.diff(hf[0], 2, axis=1)).max() == 0 and np.abs(np.diff(hf[0], 2, axis=0)).max() == 0     # exactly flat: no lattice noise
    env = ElevBatch(n, device=DEV, seed=8, heightfield=hf)
    env.set_lanes(lanes)
    p = OS.elev_params()
    for q in (env.p, p):      # no resets: no time-out, no below-minimum-height, no stuck, no rollover, no at-goal
        q.max_episode_length = 10 ** 9
        q.min_height, q.stuck_min_vel, q.upright_cos, q.goal_dist = -1e9, -1e9, -2.0, -1.0
    env.p.reset_xy = 12.0     # spawn away from the border (no drive off the edge of the field within the test)
    env.reset()
    rng = np.random.RandomState(1)
    gentle = lambda: np.stack([rng.uniform(0.1, 0.3, n), rng.uniform(-0.3, 0.3, n)], -1).astype(np.float32)
    SETTLE = 18      # the spawn drop is up to 0.7 m on these planes (reset_z = 0.25 over terrain down to -0.5): 1.2 s to ring out
    for _ in range(SETTLE):
        env.step(torch.from_numpy(gentle()).to(DEV))
    torch.cuda.synchronize()
    assert int(env.metrics[8]) == 0
    excused = 0
    for k in range(24):
        st = env.state.cpu().numpy().copy()
        pre = st.copy()
        ep = env.episode_len.cpu().numpy().copy()
        a = gentle()
        obs, rew, term, trunc = env.step(torch.from_numpy(a).to(DEV))
        torch.cuda.synchronize()
        probe = {}
        o_obs, o_rew, o_term, o_trunc, info = OS.step(p, st, ep, hf, a, 8, SETTLE + k, probe=probe)
        got = env.state.cpu().numpy()
        assert not term.any() and not trunc.any() and not o_term.any() and not o_trunc.any()
        err = PRED.state_error(got, st, n)
        # the cars this test is about: upright ones.  (All terminations are off: on the curved terrains a few of the 512 came to rest on
        # their side or roof after the spawn drop -- up to 0.7 m here -- and keep touching down with one wheel; the single-step test above
        # covers such states with the same predicate.)
        q0 = pre[3:7, :n]
        upright = (1.0 - 2.0 * (q0[1] ** 2 + q0[2] ** 2)) > 0.8
        assert upright.mean() > 0.9, (k, float(upright.mean()))
        # (on the curved terrains a crawling car keeps 3 - 5 % of the env-steps busy with a wheel touching down or lifting off: the chassis is
        # rigid and the suspension's static deflection 2.8 mm, so over uneven ground it rocks on a diagonal pair of wheels.)
        # every upright car to the tight bound; on the curved terrains ("tilted", "bench") one may miss it only with a wheel on a cell line
        # of the 16-bit lattice (the predicate of tests/parity_predicates.py) -- on the exact plane not at all
        ok, n_ex = PRED.check_state(got, st, probe, n, upright, loose=60.0, where=f"{terrain} step {k}")
        if terrain == "plane":
            assert n_ex == 0, (k, n_ex, float(err.max()))
        excused += n_ex
        np.testing.assert_allclose(rew.cpu().numpy()[ok], o_rew[ok], rtol=2e-3, atol=5e-2)
        d = np.abs(obs.cpu().numpy() - o_obs)[ok]
        d[:, 2:5] = np.minimum(d[:, 2:5], np.abs(2 * np.pi - d[:, 2:5]))
        assert d[:, :13].max() < 3e-3 and (d[:, 13:] > 2e-3).sum() <= 4
    assert excused <= 6, excused                                              # 0.05 % of the 24 x 512 env-steps; each one explained (round 5: up to 12, by a count)
    assert float(np.abs(env.state[7:9, :n].cpu().numpy()).mean()) > 0.05     # they do drive


def test_row_pair_table_is_what_the_header_says():
    """WlHeightField.pair (ABI 23): wl_heightfield_pairs on the device == the header's definition restated in torch (core.pair_table);
    the elevation entry points refuse a field without it (the height scan gathers a cell's four corners from it in one request)."""
    from wheeledlab_amd import _abi as A
    from wheeledlab_amd.core import DeviceHeightField, ElevBatch, pair_table
    g = torch.Generator().manual_seed(3)
    codes = torch.randint(-32767, 32768, (37, 52), generator=g, dtype=torch.int32).to(torch.int16)
    hf = DeviceHeightField((codes, -1.0, -1.0, 0.05, 2.0 ** -13), "cuda:0")
    want = pair_table(codes)
    assert torch.equal(hf.pairs.cpu(), want)
    lo, hi = (want & 0xffff).to(torch.int16), (want >> 16).to(torch.int16)
    assert torch.equal(lo, codes) and torch.equal(hi[:-1], codes[1:]) and torch.equal(hi[-1], codes[-1])
    env = ElevBatch(64, device="cuda:0", seed=1)
    env.reset()
    bare = A.WlHeightField(env._hf.height, env._hf.nx, env._hf.ny, env._hf.x0, env._hf.y0, env._hf.cell, env._hf.outside_z, env._hf.z_scale)
    out = torch.empty(64, 689, device="cuda:0")
    rc = env.lib.wl_elev_observe(C.byref(env.p), C.byref(env._bufs), C.byref(bare), out.data_ptr(), None)
    assert rc == -1      # WL_EINVAL
