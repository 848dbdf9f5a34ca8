"""The term plugin API on the GPU: reward / termination / observation terms WITHOUT a HIP implementation
-- any `f(env, **params) -> Tensor[N]` as the reference's configs allow (mushr_drift_env_cfg.py:219,343-362,
elevation/mushr_elevation_env_cfg.py:44-48,155-231,349-376, common/observations.py:24-54) -- registered through the
config run as torch on the state views behind the fused kernel: custom rewards are added (and logged), a custom
termination ends the episode through a masked reset launch, custom observation terms are concatenated behind the fused
block, and the elevation task's height scanner is readable through `env.scene.sensors`."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cfg(task, n):
    from wheeledlab_amd import registry, tasks  # noqa: F401
    return registry, registry.parse_env_cfg(task, device=DEV, num_envs=n)


def test_drift_lambda_termination_observation_and_reward_run_behind_the_kernel():
    from wheeledlab_amd.core import DriftBatch
    from wheeledlab_amd.envs import mdp
    from wheeledlab_amd.envs.managers_cfg import ObservationTermCfg as ObsTerm
    from wheeledlab_amd.envs.managers_cfg import RewardTermCfg as RewTerm
    from wheeledlab_amd.envs.managers_cfg import TerminationTermCfg as DoneTerm
    n, K, X_MAX = 512, 40, 0.95
    registry, cfg = _cfg("Isaac-MushrDriftRL-v0", n)
    # plain callables, nothing kernel-backed about them
    cfg.terminations.too_far_right = DoneTerm(func=lambda env, x_max: mdp.root_pos_w(env)[:, 0] > x_max, params=dict(x_max=X_MAX))
    cfg.rewards.go_fast = RewTerm(func=lambda env: mdp.base_lin_vel(env)[:, 0].clamp(min=0.0), weight=2.5)
    cfg.observations.policy.speed = ObsTerm(func=lambda env: mdp.base_lin_vel(env)[:, :2], clip=(-2.0, 2.0), scale=0.5)
    cfg.observations.policy.enable_corruption = False
    env = registry.make("Isaac-MushrDriftRL-v0", cfg=cfg)
    assert env.observation_manager.group_obs_dim["policy"] == (16,) and env._has_custom_rewards
    assert "too_far_right" in env.termination_manager.active_terms
    # the twin: the same kernel without the custom terms, re-seated on the env's state before every step
    twin = DriftBatch(n, device=DEV, seed=env._batch.seed, params=env._batch.p, startup=env._flat.startup)
    obs, _ = env.reset()
    assert obs["policy"].shape == (n, 16)
    g = torch.Generator(device=DEV).manual_seed(5)
    fired = resets = 0
    dt = cfg.sim.dt * cfg.decimation
    for k in range(K):
        twin.state.copy_(env._batch.state)
        twin.episode_len.copy_(env._batch.episode_len)
        twin.step_count = env._batch.step_count
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        a[:, 0] = a[:, 0].abs()
        o, rew, term, trunc, extras = env.step(a)
        o2, r2, t2, u2 = twin.step(a)
        o, rew, term, trunc = o["policy"].clone(), rew.clone(), term.clone(), trunc.clone()
        st = twin.state                                           # post-step state incl. the kernel's own resets
        live = ~(t2 | u2)
        q = st[3:7, :n]
        vw = st[7:10, :n]
        # body-frame velocity of the twin's state: R(q)^T v
        w, x, y, z = q
        r00, r10, r20 = 1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)
        r01, r11, r21 = 2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)
        vbx = r00 * vw[0] + r10 * vw[1] + r20 * vw[2]
        vby = r01 * vw[0] + r11 * vw[1] + r21 * vw[2]
        want_flag = (st[0, :n] > X_MAX) & live
        assert torch.equal(term, t2 | want_flag) and torch.equal(trunc, u2), k
        assert torch.equal(env.termination_manager.get_term("too_far_right"), want_flag)
        want_rew = r2 + torch.where(live, vbx.clamp(min=0.0) * (2.5 * dt), torch.zeros_like(vbx))
        torch.testing.assert_close(rew, want_rew, rtol=1e-5, atol=1e-5)
        # envs ended by the custom term were reset by the masked launch: fresh episode, at rest, on the centre line
        es = env._batch.state
        if bool(want_flag.any()):
            assert (env._batch.episode_len[:n][want_flag] == 0).all() and (es[7:13, :n][:, want_flag] == 0).all()
            assert (es[0, :n][want_flag].abs() <= 0.8 + 0.5 + 1e-5).all()
        keep = ~want_flag
        assert torch.equal(es[:, :n][:, keep], st[:, :n][:, keep])         # everyone else: exactly the kernel's state
        assert torch.equal(o[keep][:, :14], o2[keep])
        # the custom observation columns: post-reset state, clip then scale
        q = es[3:7, :n]
        w, x, y, z = q
        vw = es[7:10, :n]
        vb = torch.stack([(1 - 2 * (y * y + z * z)) * vw[0] + 2 * (x * y + w * z) * vw[1] + 2 * (x * z - w * y) * vw[2],
                          2 * (x * y - w * z) * vw[0] + (1 - 2 * (x * x + z * z)) * vw[1] + 2 * (y * z + w * x) * vw[2]], -1)
        torch.testing.assert_close(o[:, 14:], vb.clamp(-2.0, 2.0) * 0.5, rtol=1e-5, atol=1e-5)
        fired += int(want_flag.sum())
        resets += int((term | trunc).sum())
        log = extras["log"]
        assert "Episode_Reward/go_fast" in log and "Episode_Termination/too_far_right" in log
        assert int(log["Episode_Termination/too_far_right"]) == int(want_flag.sum())
    assert fired > 0 and resets > fired
    assert float(env.episode_metrics()[8]) == resets                # every episode end was booked, whoever ended it
    env.close()


def test_elevation_unwired_reward_custom_termination_and_the_height_scanner():
    from oracle import elev_step as OE
    from oracle import heightfield as OH
    from wheeledlab_amd.envs import mdp
    from wheeledlab_amd.envs.managers_cfg import ObservationTermCfg as ObsTerm
    from wheeledlab_amd.envs.managers_cfg import RewardTermCfg as RewTerm
    from wheeledlab_amd.envs.managers_cfg import SceneEntityCfg
    from wheeledlab_amd.envs.managers_cfg import TerminationTermCfg as DoneTerm
    n = 256
    registry, cfg = _cfg("Isaac-MushrElevationRL-v0", n)
    cfg.rewards.upright = RewTerm(func=mdp.upright_penalty, weight=-3.0, params=dict(thresh_deg=2.0))      # an E12 function
    cfg.rewards.low_vel = RewTerm(func=mdp.low_vel_penalty, weight=-1.0, params=dict(min_vel=0.1))
    cfg.terminations.tilted = DoneTerm(func=lambda env, deg: mdp.upright_penalty(env, deg) > 0, params=dict(deg=12.0))
    cfg.observations.policy.scan_min = ObsTerm(
        func=lambda env, sensor_cfg: mdp.height_scan(env, sensor_cfg, offset=0.0).amin(1, keepdim=True),
        params=dict(sensor_cfg=SceneEntityCfg("height_scanner")))
    env = registry.make("Isaac-MushrElevationRL-v0", cfg=cfg)
    b = env._batch
    assert env.observation_manager.group_obs_dim["policy"] == (690,) and "height_scanner" in env.scene.sensors
    obs, _ = env.reset()
    # the sensor view against the numpy restatement of the ray caster
    p = OE.elev_params()
    hf = OH.make_terrain()
    st = b.state.cpu().numpy()
    want_map = OE.height_map(p, st[:, :n], hf)
    sensor = env.scene.sensors["height_scanner"]
    hits = sensor.data.ray_hits_w
    assert hits.shape == (n, 676, 3) and sensor.data.pos_w.shape == (n, 3)
    inside = want_map < p.obs_clip
    got_z = hits[..., 2].cpu().numpy()
    np.testing.assert_allclose(got_z[inside], (want_map - np.float32(p.scan_offset) + np.float32(p.elev_z0))[inside], atol=2e-5)
    assert np.isinf(got_z[~inside]).all()
    torch.testing.assert_close(sensor.data.pos_w[:, 2], b.state[2, :n] + 20.0)
    hs = mdp.height_scan(env, SceneEntityCfg("height_scanner"), offset=0.084)            # IsaacLab's definition
    np.testing.assert_allclose(hs.cpu().numpy()[inside], (st[2, :n, None] + 20.0 - got_z - 0.084)[inside], rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(obs["policy"][:, 689:], mdp.height_scan(env, SceneEntityCfg("height_scanner"), 0.0).amin(1, keepdim=True))
    # steps: rewards of the two E12 terms + the custom termination, against the same formulas in numpy on the device state
    g = torch.Generator(device=DEV).manual_seed(2)
    ended = 0
    for k in range(30):
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        base_w = [(nm, env.reward_manager.get_term_cfg(nm).weight) for nm in ("upright", "low_vel")]
        o, rew, term, trunc, extras = env.step(a)
        flag = env.termination_manager.get_term("tilted").clone()
        s = b.state.cpu().numpy()[:, :n]
        # envs that are still running (neither the kernel nor the custom term ended them) show their post-step state
        running = ~(term | trunc).cpu().numpy()
        q = s[3:7]
        up = 1 - 2 * (q[1] ** 2 + q[2] ** 2)
        tilt = np.degrees(np.arccos(np.clip(up, -1, 1)))
        assert (tilt[running] <= 12.0 + 1e-3).all()                       # whoever tilted further was ended
        ended += int(flag.sum())
        assert (b.episode_len[:n][flag] == 0).all()
        assert "Episode_Reward/upright" in extras["log"] and "Episode_Termination/tilted" in extras["log"]
        assert base_w == [(nm, env.reward_manager.get_term_cfg(nm).weight) for nm in ("upright", "low_vel")]
        assert o["policy"].shape == (n, 690) and torch.isfinite(o["policy"][:, :13]).all()
    assert ended > 0
    # one more step, checked term by term: reward = kernel reward + the two E12 terms on the running envs
    from wheeledlab_amd.core import ElevBatch
    twin = ElevBatch(n, device=DEV, seed=b.seed, params=b.p, startup=env._flat.startup)
    twin.state.copy_(b.state)
    twin.episode_len.copy_(b.episode_len)
    twin.step_count = b.step_count
    a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
    o, rew, term, trunc, _ = env.step(a)
    o2, r2, t2, u2 = twin.step(a)
    s = twin.state[:, :n]
    live = ~(t2 | u2)
    q = s[3:7]
    up = 1 - 2 * (q[1] ** 2 + q[2] ** 2)
    tilt = torch.rad2deg(torch.arccos(up))
    pen = torch.where(tilt > 2.0, tilt - 2.0, torch.zeros_like(tilt))
    w, x, y, z = q
    vbx = (1 - 2 * (y * y + z * z)) * s[7] + 2 * (x * y + w * z) * s[8] + 2 * (x * z - w * y) * s[9]
    dt = cfg.sim.dt * cfg.decimation
    extra = torch.where(live, (-3.0 * pen - 1.0 * (vbx < 0.1).float()) * dt, torch.zeros_like(pen))
    torch.testing.assert_close(rew, r2 + extra, rtol=1e-4, atol=1e-4)
    assert torch.equal(term, t2 | ((pen > 10.0) & live))
    env.close()


def test_visual_depth_observation_term_through_the_scene_camera():
    """the reference's depth observation functions (visual/mdp_sensors/observations.py:89-95: `camera_data_depth` /
    `raycast_depth` return `env.scene.sensors[cfg.name].data.output["distance_to_image_plane"]`) against this env: the scene's
    camera renders it with the depth ray-cast kernel; registered as an ObsTerm it is concatenated behind the fused 3208 values.
    On the visual task's flat ground every pixel below the horizon is the analytic ray / plane distance."""
    from oracle.mathlib import matrix_from_quat
    from wheeledlab_amd.envs import mdp
    from wheeledlab_amd.envs.managers_cfg import ObservationTermCfg as ObsTerm
    from wheeledlab_amd.envs.managers_cfg import SceneEntityCfg
    n = 64
    registry, cfg = _cfg("Isaac-MushrVisualRL-v0", n)
    # written as the reference writes it: a plain function of (env, sensor_cfg) over the scene's sensor data
    ref_style = lambda env, sensor_cfg: env.scene.sensors[sensor_cfg.name].data.output["distance_to_image_plane"]
    cfg.observations.policy.depth = ObsTerm(func=ref_style, params=dict(sensor_cfg=SceneEntityCfg("camera")), clip=(0.0, 20.0))
    env = registry.make("Isaac-MushrVisualRL-v0", cfg=cfg)
    assert env.observation_manager.group_obs_dim["policy"] == (3208 + 4800,)
    # the constructor's shape probe ran the term on the un-reset state (all cars at the origin): nothing of it may survive into the
    # first observation -- the scene camera caches one render per (step, pose epoch), and reset() starts a new epoch
    cam = env.scene.sensors["camera"].data
    assert cam._cached == (None, None)
    probe = cam._camera().render(env._batch, cam.far).clone()              # what the shape probe saw: the un-reset state
    obs, _ = env.reset()
    fresh = cam._camera().render(env._batch, cam.far)
    assert torch.equal(obs["policy"][:, 3208:], fresh.reshape(n, -1).clamp(0.0, 20.0))
    assert not torch.equal(fresh, probe)                                   # (on flat ground the image depends on height and tilt only)
    # a second reset without a step in between moves the cars again: again a fresh render
    obs, _ = env.reset()
    again = cam._camera().render(env._batch, cam.far)
    assert torch.equal(obs["policy"][:, 3208:], again.reshape(n, -1).clamp(0.0, 20.0))
    # a plugin's pose write (what a user reset event does) invalidates the cached image as well
    before = mdp.camera_data_depth(env).clone()
    robot = env.scene["robot"]
    pose0 = torch.cat([robot.data.root_pos_w, robot.data.root_quat_w], 1)
    pose = pose0.clone()
    pose[:, 2] += 0.5
    robot.write_root_pose_to_sim(pose)
    lifted = mdp.camera_data_depth(env)
    assert not torch.equal(lifted, before)
    assert torch.equal(lifted[..., 0], cam._camera().render(env._batch, cam.far))
    robot.write_root_pose_to_sim(pose0)
    assert torch.equal(mdp.camera_data_depth(env), before)
    for _ in range(5):
        obs, *_ = env.step(torch.rand(n, 2, device=DEV) * 2 - 1)
    d = mdp.raycast_depth(env)
    assert d.shape == (n, 60, 80, 1) and torch.equal(d, mdp.camera_data_depth(env))
    assert torch.equal(obs["policy"][:, 3208:], d.reshape(n, -1).clamp(0.0, 20.0))
    st = env._batch.state[:, :n].cpu().numpy()
    p = env._batch.p
    R = matrix_from_quat(st[3:7].T)
    o = st[0:3].T + R @ np.array(list(p.cam_pos), np.float32)
    rows, cols = np.arange(60), np.arange(80)
    db = np.stack(np.broadcast_arrays(np.ones((60, 80)), -((cols[None, :] + 0.5 - p.cx) / p.fx), -((rows[:, None] + 0.5 - p.cy) / p.fy)), -1)
    dw = np.einsum("nij,rcj->nrci", R, db)
    want = np.where(dw[..., 2] < -1e-9, -o[:, None, None, 2] / np.minimum(dw[..., 2], -1e-9), 100.0).clip(0, 100.0)
    got = d[..., 0].cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)
    assert (got[:, 45:] < 5.0).all() and (got[:, :20] == 100.0).mean() > 0.9      # ground close below, sky (far plane) above
    # pixels beyond the far plane: "max" (default here) / "zero" / "none" (+inf, IsaacLab's default)
    cam = env.scene.sensors["camera"].data
    for mode, val in (("zero", 0.0), ("none", float("inf"))):
        cam.beyond = val
        dm = mdp.camera_data_depth(env)[..., 0]
        assert torch.equal(dm == val, d[..., 0] >= 100.0) and torch.equal(dm[d[..., 0] < 100.0], d[..., 0][d[..., 0] < 100.0]), mode
    cam.beyond = None


def test_visual_unwired_terms_wired_through_a_config_override():
    """the term functions the reference's visual cfg module defines without registering them (mushr_visual_env_cfg.py:314-368,
    400-403; pinned to the reference's outputs on CPU: tests/test_plugin_terms_cpu.py) wired in as the reference's users would --
    `is_traversable_wheels` / `vel_rew_trav` as rewards, `binary_is_traversable_wheels` as a termination, `roll_over` and
    `low_speed_penalty` read directly -- run as torch terms behind the fused kernel.  Checked against a twin batch stepped with the
    same kernel and the formulas written out on the twin's post-step state (wheel centres from the oracle's rotation matrix, the map
    lookup from the reference's index arithmetic)."""
    from oracle.mathlib import matrix_from_quat
    from wheeledlab_amd.core import VisualBatch
    from wheeledlab_amd.envs import mdp
    from wheeledlab_amd.envs.managers_cfg import RewardTermCfg as RewTerm
    from wheeledlab_amd.envs.managers_cfg import TerminationTermCfg as DoneTerm
    n = 256
    registry, cfg = _cfg("Isaac-MushrVisualRL-v0", n)
    cfg.rewards.wheels = RewTerm(func=mdp.is_traversable_wheels, weight=0.5)
    cfg.rewards.speed = RewTerm(func=mdp.vel_rew_trav, weight=2.0, params=dict(speed_target_on_trav=1.0, speed_target_on_non_trav=2.0))
    cfg.terminations.all_wheels_off = DoneTerm(func=mdp.binary_is_traversable_wheels)
    env = registry.make("Isaac-MushrVisualRL-v0", cfg=cfg)
    b = env._batch
    assert env._has_custom_rewards and "all_wheels_off" in env.termination_manager.active_terms
    tmap, (rs, cs) = env.traversability
    tm = tmap.cpu().numpy().astype(bool)
    rows, cols = tm.shape
    v = b.p.vehicle
    local = np.array([[-v.half_wheelbase_r, v.half_track, v.wheel_z], [-v.half_wheelbase_r, -v.half_track, v.wheel_z],
                      [v.half_wheelbase_f, v.half_track, v.wheel_z], [v.half_wheelbase_f, -v.half_track, v.wheel_z]], np.float32)

    def lookup(xy):
        xi = np.clip(((xy[..., 0] + np.float32(rows * rs / 2.0) + np.float32(rs / 2.0)) / np.float32(rs)).astype(np.int64), 0, rows - 1)
        yi = np.clip(((xy[..., 1] + np.float32(cols * cs / 2) + np.float32(cs / 2)) / np.float32(cs)).astype(np.int64), 0, cols - 1)
        return tm[yi, xi]

    env.reset()
    # the scene view of the wheel links against the geometry written out
    st = b.state[:, :n].cpu().numpy()
    want_body = st[0:3].T[:, None, :] + np.einsum("nij,bj->nbi", matrix_from_quat(st[3:7].T), local)
    cfgw = mdp.SceneEntityCfg("robot", body_names=".*wheel_link").resolve(env.scene)
    got_body = env.scene["robot"].data.body_pos_w[:, cfgw.body_ids].cpu().numpy()
    np.testing.assert_allclose(got_body, want_body, rtol=1e-5, atol=1e-5)
    assert mdp.roll_over(env).dtype == torch.bool and mdp.low_speed_penalty(env).shape == (n,)
    assert not mdp.bool_is_not_traversable(env).any()                   # fewer than 1000 episodes so far: the delayed term is off
    twin = VisualBatch(n, device=DEV, seed=b.seed, params=b.p, startup=env._flat.startup, trav_map=tm,
                       spacing=(rs, cs))
    g = torch.Generator(device=DEV).manual_seed(4)
    dt = cfg.sim.dt * cfg.decimation
    ended = 0
    for k in range(12):
        twin.state.copy_(b.state)
        twin.episode_len.copy_(b.episode_len)
        twin.step_count = b.step_count
        a = torch.rand(n, 2, device=DEV, generator=g) * 2 - 1
        a[:, 0] = a[:, 0].abs()
        o, rew, term, trunc, extras = env.step(a)
        rew, term = rew.clone(), term.clone()
        _, r2, t2, u2 = twin.step(a)
        live = ~(t2 | u2).cpu().numpy()
        s = twin.state[:, :n].cpu().numpy()
        R = matrix_from_quat(s[3:7].T)
        wheels = s[0:3].T[:, None, :] + np.einsum("nij,bj->nbi", R, local)
        on = lookup(wheels[..., :2])                                              # [n, 4]
        r_wheels = np.where(on, 1.0, -5.0).sum(-1)
        vb = np.einsum("nji,nj->ni", R, s[7:10].T)
        target = np.where(lookup(s[0:2].T), 1.0, 2.0)
        sd = -((np.linalg.norm(vb, axis=-1) - target) ** 2) + target ** 2
        r_speed = np.where(sd > 0, sd, 0.0)
        extra = np.where(live, (0.5 * r_wheels + 2.0 * r_speed) * dt, 0.0)
        # a wheel within a hair of a cell line may round to the other cell: those envs are excused from the reward comparison
        frac = (wheels[..., :2] + 125.25) / 0.5
        near = (np.abs(frac - np.round(frac)) < 1e-3).any(axis=(1, 2))
        d = np.abs(rew.cpu().numpy() - (r2.cpu().numpy() + extra))
        assert (d[~near] < 2e-4 + 2e-4 * np.abs(extra[~near])).all(), (k, d[~near].max())
        off = (~on).all(-1) & live
        assert np.array_equal(term.cpu().numpy()[~near], (t2.cpu().numpy() | off)[~near])
        ended += int(off.sum())
        assert "Episode_Reward/wheels" in extras["log"] and "Episode_Termination/all_wheels_off" in extras["log"]
    assert ended > 0                                                             # cars do leave the paths within 12 steps
    env.close()


def test_camera_data_rgb_is_what_the_fused_grey_image_is_made_of():
    """`camera_data_rgb` (visual/mdp_sensors/observations.py:60-62): the un-flattened [N, 60, 80, 3] uint8 image of the scene's camera.
    Pushed through the reference's own chain for the flattened term (:64-73: drop the top third, grey = 0.2989 R + 0.587 G + 0.114 B
    of the image / 255, normalise with mean 0.5 / std 0.5) it is the fused kernel's un-augmented observation -- up to the uint8 rounding
    of the sky's 0.5 and the pixels whose ray lands within rounding of a cell edge of the map"""
    from wheeledlab_amd.envs import mdp
    n = 64
    registry, cfg = _cfg("Isaac-MushrVisualRL-v0", n)
    env = registry.make("Isaac-MushrVisualRL-v0", cfg=cfg)
    env.reset()
    for _ in range(3):
        env.step(torch.rand(n, 2, device=env.device) * 2 - 1)
    rgb = mdp.camera_data_rgb(env)
    assert rgb.shape == (n, 60, 80, 3) and rgb.dtype == torch.uint8
    assert torch.equal(rgb[..., 0], rgb[..., 1]) and torch.equal(rgb[..., 1], rgb[..., 2])
    assert set(torch.unique(rgb).tolist()) <= {0, 127, 255} and (rgb == 255).any() and (rgb == 0).any()
    img = rgb[:, 20:].permute(0, 3, 1, 2).float() / 255.0
    grey = 0.2989 * img[:, 0] + 0.587 * img[:, 1] + 0.114 * img[:, 2]
    want = ((grey - 0.5) / 0.5).reshape(n, -1)
    got = mdp.camera_data_rgb_flattened(env)
    assert got.shape == want.shape == (n, 3200)
    off = (got - want).abs() > 0.01
    assert off.float().mean() < 5e-3, float(off.float().mean())
    env.close()
