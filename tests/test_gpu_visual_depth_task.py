"""The visual-DEPTH extension task on the GPU (BASELINE.json configs[4] as a task: the visual task's env.step() on a heightfield
terrain with the camera's depth image as the policy observation).  Step + observation vs the oracle (oracle/visual_step.py with
`hf`: the same vehicle model over heightfield.sample, oracle/depth.c for the image), both step forms; the env surface through
the registry (`Isaac-MushrVisualDepthRL-v0`, clearly an extension id); properties and shard equality at 4096 envs."""
import numpy as np
import pytest
import torch

from oracle import heightfield as OH
from oracle import visual_step as OS
from tests import parity_predicates as PRED

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MAX_DEPTH = 20.0


def _batch(n, seed, off=0, params=None):
    from wheeledlab_amd.core import VisualDepthBatch
    hf = OH.make_terrain()
    env = VisualDepthBatch(n, device=DEV, seed=seed, env_offset=off, heightfield=hf, max_depth=MAX_DEPTH, params=params)
    env.reset()
    torch.cuda.synchronize()
    return env, hf


def _oracle_params(env):
    p = OS.visual_params()
    p.map_rows, p.map_cols = int(env._map.rows), int(env._map.cols)
    return p


def test_reset_places_cars_on_the_terrain_and_first_observation_matches_oracle():
    n = 96
    env, hf = _batch(n, 11)
    st = env.state.cpu().numpy()
    trav = env.trav_map.cpu().numpy().astype(bool)
    p = _oracle_params(env)
    o = np.zeros_like(st)
    o[3] = 1
    o[23:27] = st[23:27]
    ep = np.ones(st.shape[1], np.int32)
    OS.reset_envs(p, o, ep, OS.spawn_cells(trav), np.arange(n), 11, 0, hf=hf)
    np.testing.assert_allclose(st[:23, :n], o[:23, :n], rtol=1e-6, atol=2e-6)
    zt, _, _ = OH.sample(*hf, st[0, :n], st[1, :n])
    np.testing.assert_allclose(st[2, :n] - zt, 0.1, atol=1e-5)                       # 0.1 m above the terrain under the spawn cell
    assert np.abs(st[0, :n]).max() <= 20.0 and np.abs(st[1, :n]).max() <= 20.0 and zt.max() > 0.25   # on the field, hills included
    obs = env.observe().cpu().numpy()
    want = OS.observe_depth(p, st[:, :n].copy(), hf, MAX_DEPTH)
    assert obs.shape == (n, 4808)
    img_bad = np.abs(obs[:, :4800] - want[:, :4800]) > 2e-4 + 2e-4 * np.abs(want[:, :4800])
    assert img_bad.mean() < 1e-3, int(img_bad.sum())                                  # grazing rays, counted
    np.testing.assert_allclose(obs[:, 4800:], want[:, 4800:], atol=1e-6)
    assert (obs[:, :4800] >= 0).all() and (obs[:, :4800] <= MAX_DEPTH).all() and (obs[:, :4800] < MAX_DEPTH).mean() > 0.3


@pytest.mark.parametrize("lanes", [4, 1])
def test_visual_depth_step_matches_oracle_single_steps(lanes):
    n = 128
    env, hf = _batch(n, 5)
    env.set_lanes(lanes)
    trav = env.trav_map.cpu().numpy().astype(bool)
    cells = OS.spawn_cells(trav)
    p = _oracle_params(env)
    rng = np.random.RandomState(0)
    img_bad_total = excused = 0
    for k in range(10):
        st = env.state.cpu().numpy().copy()
        ep = env.episode_len.cpu().numpy().copy()
        if k == 5:
            ep[: n // 4] = 49                                   # time-outs
            st[0, n // 4: n // 2] = 19.95                       # about to leave the map -> out_of_map termination
            st[2, n // 4: n // 2] = 0.19 + 0.06
            st[7, n // 4: n // 2] = 3.0
            env.episode_len.copy_(torch.from_numpy(ep))
            env.state.copy_(torch.from_numpy(st))
        a = rng.uniform(-1.2, 1.2, (n, 2)).astype(np.float32)
        a[:, 0] = np.abs(a[:, 0]) * 0.7 + 0.2
        met0 = env.metrics.cpu().numpy().astype(np.float64)
        obs, rew, term, trunc = env.step(torch.from_numpy(a).to(DEV))
        torch.cuda.synchronize()
        met = np.zeros(16)
        probe = {}
        o_obs, o_rew, o_term, o_trunc, info = OS.step(p, st, ep, trav, cells, a, 5, k, met, hf=hf, max_depth=MAX_DEPTH, probe=probe)
        got = env.state.cpu().numpy()
        np.testing.assert_array_equal(trunc.cpu().numpy(), o_trunc)
        bad = term.cpu().numpy() != o_term
        assert bad.sum() <= 1
        ok = ~bad
        # 10 sub-steps of 20 ms over the heightfield with contact make / break (the 10 cm spawn drop): an env may miss the tight bound only
        # if the ORACLE's step had a wheel within reach of a discontinuity (tests/parity_predicates.py), as in the elevation step test
        ok, n_ex = PRED.check_state(got, st, probe, n, ok, where=f"step {k}")
        excused += n_ex
        assert PRED.state_error(got, st, n)[:, ok].max() <= 1.0, k
        cell_flip = np.abs(rew.cpu().numpy() - o_rew) > 0.5     # +-1 traversability flips exactly on a cell edge
        assert (cell_flip & ok).sum() <= 1
        sel = ok & ~cell_flip
        np.testing.assert_allclose(rew.cpu().numpy()[sel], o_rew[sel], rtol=2e-3, atol=3e-3)
        # the observation of the DEVICE's post-step state: image vs oracle/depth.c on that state (the oracle's own post-step pose differs
        # by the step tolerance above, which moves silhouettes by pixels), proprio against the oracle's step
        want_img = OS.observe_depth(p, got[:, :n].copy(), hf, MAX_DEPTH)
        o = obs.cpu().numpy()
        img_bad = np.abs(o[:, :4800] - want_img[:, :4800]) > 2e-4 + 2e-4 * np.abs(want_img[:, :4800])
        img_bad_total += int(img_bad.sum())
        np.testing.assert_allclose(o[:, 4800:], want_img[:, 4800:], atol=2e-6)
        assert np.abs(o[sel, 4800:] - o_obs[sel, 4800:]).max() < 3e-3
        if not bad.any():
            dm = env.metrics.cpu().numpy().astype(np.float64) - met0
            np.testing.assert_allclose(dm[8:16], met[8:16], atol=1e-3)
    assert img_bad_total < 1e-3 * 10 * n * 4800, img_bad_total
    assert excused <= 3, excused                                # (round 5: max(3, n / 25) envs PER STEP at 600 x the bound, by a count)
    assert env.metrics[10] > 0 and env.metrics[9] > 0           # both out_of_map and time_out were exercised


@pytest.mark.parametrize("lanes", [4, 1])
def test_settled_cars_need_no_contact_excuse(lanes):
    """Companion of test_visual_depth_step_matches_oracle_single_steps (which excuses up to max(3, n / 25) envs per step as contact
    make / break discontinuities, the spawn drop above all): here nothing makes or breaks contact -- the cars have settled for 6 steps
    on the task's own terrain and then crawl, time-outs and out_of_map are out of reach on BOTH sides -- and for 12 steps the excused
    set is (nearly) EMPTY: as on the elevation task's bench terrain (tests/test_gpu_elev_parity.py::test_settled_cars_need_no_contact_
    excuse[bench]) a crawling car unloads ONE wheel over a crest now and then (measured: 1 env-step of 3072 in either form): <= 2 envs
    per step, < 60 x the bound, <= 0.2 % of all env-steps, every other env to the bound.  Both step forms."""
    n = 256
    env, hf = _batch(n, 21)
    env.set_lanes(lanes)
    trav = env.trav_map.cpu().numpy().astype(bool)
    cells = OS.spawn_cells(trav)
    p = _oracle_params(env)
    for q in (env.p, p):
        q.max_episode_length = 10 ** 9
    # keep every car well inside the field (out_of_map would reset it): spawn cells are anywhere on the 40 m map; re-seat the
    # outermost ones towards the middle, on the terrain
    st = env.state.cpu().numpy()
    far = (np.abs(st[0, :n]) > 15.0) | (np.abs(st[1, :n]) > 15.0)
    st[0, :n][far] *= 0.5
    st[1, :n][far] *= 0.5
    zt, _, _ = OH.sample(*hf, st[0, :n], st[1, :n])
    st[2, :n] = zt + 0.1
    env.state.copy_(torch.from_numpy(st))
    rng = np.random.RandomState(3)
    gentle = lambda: np.stack([rng.uniform(0.05, 0.2, n), rng.uniform(-0.3, 0.3, n)], -1).astype(np.float32)
    for _ in range(6):                                                                 # 6 x 0.2 s: the 10 cm drop has rung out
        env.step(torch.from_numpy(gentle()).to(DEV))
    torch.cuda.synchronize()
    assert int(env.metrics[8]) == 0
    excused = 0
    for k in range(12):
        st = env.state.cpu().numpy().copy()
        ep = env.episode_len.cpu().numpy().copy()
        a = gentle()
        obs, rew, term, trunc = env.step(torch.from_numpy(a).to(DEV))
        torch.cuda.synchronize()
        met = np.zeros(16)
        probe = {}
        o_obs, o_rew, o_term, o_trunc, info = OS.step(p, st, ep, trav, cells, a, 21, 6 + k, met, hf=hf, max_depth=MAX_DEPTH, probe=probe)
        got = env.state.cpu().numpy()
        assert not term.any() and not trunc.any() and not o_term.any() and not o_trunc.any()
        # every env to the tight bound unless the oracle's step shows a wheel within reach of a discontinuity (a wheel unloading over a
        # crest, a cell line of the lattice: tests/parity_predicates.py)
        ok, n_ex = PRED.check_state(got, st, probe, n, np.ones(n, bool), loose=60.0, where=f"step {k}")
        excused += n_ex
        cell_flip = np.abs(rew.cpu().numpy() - o_rew) > 0.5      # +-1 traversability flips exactly on a cell edge
        assert (cell_flip & ok).sum() <= 1
        sel = ok & ~cell_flip
        np.testing.assert_allclose(rew.cpu().numpy()[sel], o_rew[sel], rtol=2e-3, atol=3e-3)
        assert np.abs(obs.cpu().numpy()[sel, 4800:] - o_obs[sel, 4800:]).max() < 3e-3
    print(f"visual-depth settled cars, lanes {lanes}: {excused} excused env-steps of {12 * n}")
    assert excused <= 2, excused


def test_a_refused_depth_step_leaves_the_batch_untouched():
    """wl_visual_depth_step is all-or-nothing: arguments only the depth launch checks (the camera's focal length here) are validated
    BEFORE the step kernel advances state, episode lengths and metrics -- a refused call changes nothing and the caller's step
    counter stays where it was"""
    from wheeledlab_amd import _abi as A
    n = 64
    env, _ = _batch(n, 9)
    env.step(torch.zeros(n, 2, device=DEV))
    torch.cuda.synchronize()
    st, ep, met, k = env.state.clone(), env.episode_len.clone(), env.metrics_raw.clone(), env.step_count
    fx = env.p.fx
    env.p.fx = 0.0
    with pytest.raises(A.WlError):
        env.step(torch.ones(n, 2, device=DEV))
    torch.cuda.synchronize()
    env.p.fx = fx
    assert env.step_count == k and torch.equal(env.state, st) and torch.equal(env.episode_len, ep) and torch.equal(env.metrics_raw, met)
    env.step(torch.ones(n, 2, device=DEV))                        # and the batch still steps
    assert env.step_count == k + 1 and not torch.equal(env.state, st)


def test_registered_extension_env_surface():
    from wheeledlab_amd import registry, tasks  # noqa: F401
    from wheeledlab_amd.envs import mdp
    n = 64
    cfg = registry.parse_env_cfg("Isaac-MushrVisualDepthRL-v0", device=DEV, num_envs=n)
    env = registry.make("Isaac-MushrVisualDepthRL-v0", cfg=cfg)
    assert env._task == "visual_depth" and env.single_observation_space["policy"].shape == (4808,)
    obs, _ = env.reset()
    assert obs["policy"].shape == (n, 4808)
    g = torch.Generator(device=DEV).manual_seed(0)
    for _ in range(8):
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        a[:, 0] = a[:, 0].abs()
        obs, rew, term, trunc, extras = env.step(a)
    o = obs["policy"]
    assert torch.isfinite(o).all() and torch.isfinite(rew).all() and rew.shape == (n,) and term.dtype == torch.bool
    # the reference's observation functions read the same image from the sensor (mdp_sensors/observations.py:89-95)
    img = mdp.raycast_depth(env, mdp.SceneEntityCfg("camera"))
    assert img.shape == (n, 60, 80, 1) and torch.equal(img.reshape(n, -1), o[:, :4800])
    assert torch.equal(mdp.camera_data_depth(env, cfg.observations.policy.depth.params["sensor_cfg"]), img)
    d = env.scene["robot"].data
    torch.testing.assert_close(o[:, 4800:4803], d.root_lin_vel_b, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(o[:, 4803:4806], d.root_ang_vel_b, rtol=1e-5, atol=1e-5)
    # the built-in reward terms evaluate through the visual task's terms kernel on this task too
    assert mdp.traversable_reward(env).abs().max() == 1 and mdp.forward_vel(env).shape == (n,)


def test_full_size_properties_and_half_shards():
    from wheeledlab_amd.params import visual_params
    n, K = 4096, 12

    def make(m, off):
        p = visual_params()
        p.max_episode_length = 5
        return _batch(m, 23, off, p)[0]
    big, halves = make(n, 0), [make(n // 2, r * (n // 2)) for r in range(2)]
    g = torch.Generator(device=DEV).manual_seed(3)
    resets = 0
    for k in range(K):
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        ob, rb, tb, ub = [t.clone() for t in big.step(a)]
        resets += int((tb | ub).sum())
        for r, h in enumerate(halves):
            sl = slice(r * (n // 2), (r + 1) * (n // 2))
            oh, rh, th, uh = h.step(a[sl].contiguous())
            assert torch.equal(oh, ob[sl]) and torch.equal(rh, rb[sl]) and torch.equal(th, tb[sl]) and torch.equal(uh, ub[sl]), (k, r)
    torch.cuda.synchronize()
    st = big.state[:, :n]
    assert resets > n and torch.isfinite(st).all() and torch.isfinite(ob).all()
    assert ((st[3:7] ** 2).sum(0).sqrt() - 1).abs().max() < 1e-5
    assert (ob[:, :4800] >= 0).all() and (ob[:, :4800] <= MAX_DEPTH).all()
    # on the terrain (0.19 .. ~1.2 m).  The root of a car that came to rest standing on its nose or lying on its side -- no rollover
    # termination in this task, wheels are spheres, the chassis has no collision shape -- is up to 0.17 m BELOW the surface it leans on
    assert st[2].min() > 0.19 - 0.2 and st[2].max() < 1.6
    for r, h in enumerate(halves):
        sl = slice(r * (n // 2), (r + 1) * (n // 2))
        assert torch.equal(h.state[:, : n // 2], big.state[:, sl])
    assert float((halves[0].metrics + halves[1].metrics)[8]) == float(big.metrics[8]) == resets


def test_malformed_calls_are_refused():
    import ctypes as C

    from wheeledlab_amd import _abi as A
    env, _ = _batch(64, 1)
    a = torch.zeros(64, 2, device=DEV)
    lib, st = env.lib, env._stream()
    ok = (C.byref(env.p), C.byref(env._bufs), C.byref(env._map), C.byref(env._hf), env.camera.pyramid.data_ptr(), MAX_DEPTH,
          a.data_ptr(), C.byref(env._out), 1, 0, st)
    assert lib.wl_visual_depth_step(*ok) == 0
    for i, bad in ((3, None), (4, None), (5, 0.0), (6, None), (7, None)):
        args = list(ok)
        args[i] = bad
        assert lib.wl_visual_depth_step(*args) == -1, i
    assert lib.wl_visual_depth_rows(C.byref(env.p), C.byref(env._bufs), C.byref(env._hf), env.camera.pyramid.data_ptr(), MAX_DEPTH,
                                    env.obs.data_ptr(), 4799, st) == -1               # rows shorter than an image
    torch.cuda.synchronize()


def test_an_untrained_policy_never_drives_a_car_out_of_fp32():
    """4096 envs x 300 steps of N(0, 1) actions (what PPO's first iterations send) on the task's terrain at mu = 2, h = 20 ms, no rollover
    termination: cars land on their sides and roofs.  No env may go non-finite (metric 14; the first implicit build lost 1 - 2 envs per
    512 x 24 env-steps this way, profiles/r06_train_visual_depth_config_8it.json's predecessor), and speeds stay physical."""
    n = 4096
    env, hf = _batch(n, 3)
    g = torch.Generator(device=DEV).manual_seed(9)
    vmax = wmax = 0.0
    for k in range(300):
        a = torch.randn(n, 2, device=DEV, generator=g).clamp(-1.5, 1.5)
        env.step(a)
        if k % 10 == 9:
            vmax = max(vmax, float(env.state[7:10, :n].abs().max()))
            wmax = max(wmax, float(env.state[10:13, :n].abs().max()))
    torch.cuda.synchronize()
    m = env.metrics.cpu().numpy()
    assert m[14] == 0 and torch.isfinite(env.state[:, :n]).all(), m[14]
    assert m[8] > 0 and vmax < 15.0 and wmax < 80.0, (vmax, wmax)
