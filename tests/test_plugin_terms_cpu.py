"""the reward functions the reference's elevation cfg module defines without registering them
(mushr_elevation_env_cfg.py:159-164,175-231,256-266) exist in the build as torch terms on the env's state views
(wheeledlab_amd/envs/mdp.py) so that a config override can wire them in; here they are held against the outputs of the
reference's own functions (tests/golden*/elevation_unwired.npz, made by tests/golden/gen_golden.py)."""
import types

import numpy as np
import pytest
import torch

from wheeledlab_amd.envs import mdp
from wheeledlab_amd.envs.scene import MUSHR_BODY_NAMES, MUSHR_JOINT_NAMES, ArticulationView


class _Data:
    def __init__(self, g, pos_key="pos"):
        t = lambda k: torch.from_numpy(g[k].copy())   # noqa: E731
        self.root_pos_w, self.root_quat_w = t(pos_key), t("quat")
        self.root_lin_vel_b, self.root_ang_vel_b, self.root_lin_vel_w = t("lin_vel_b"), t("ang_vel_b"), t("lin_vel_w")
        self.joint_vel = t("joint_vel")


class _Scene:
    def __init__(self, robot, n):
        self._robot, self.env_origins = robot, torch.zeros(n, 3)

    def __getitem__(self, key):
        return self._robot


def _make_env(g, pos_key="pos"):
    robot = types.SimpleNamespace(data=_Data(g, pos_key), joint_names=list(MUSHR_JOINT_NAMES))
    robot.find_joints = types.MethodType(ArticulationView.find_joints, robot)
    return types.SimpleNamespace(scene=_Scene(robot, g["pos"].shape[0]))


def test_unwired_elevation_reward_functions_match_the_reference(golden):
    g = golden("elevation_unwired")
    env = _make_env(g)
    cases = {
        "forward_wheel_spin": mdp.forward_wheel_spin(env),
        "change_in_elevation": mdp.change_in_elevation(env),
        "steep_penalty": mdp.steep_penalty(env, 0.2),
        "yaw_change_onElev": mdp.yaw_change_onElev(env, 0.5, 0.1),
        "roll_on_elev": mdp.roll_on_elev(env, 0.1, 0.1),
        "ascending": mdp.ascending(env),
        "low_vel_penalty": mdp.low_vel_penalty(env, 0.1),
        "upright_penalty_30": mdp.upright_penalty(env, 30.0),
    }
    for name, got in cases.items():
        want = g[name]
        assert got.shape == want.shape, name
        np.testing.assert_allclose(got.numpy(), want.astype(np.float32), rtol=2e-5, atol=2e-5, err_msg=name)
        assert np.abs(want).max() > 0, name            # the case exercises the term
    # elevation_continuity: state across calls lives on the env (the reference: one function attribute per process)
    first = mdp.elevation_continuity(env, 0.1)
    np.testing.assert_array_equal(first.numpy(), g["elevation_continuity_first"])
    env2 = _make_env(g, "pos_second")
    env2._elev_continuity_prev = env._elev_continuity_prev
    second = mdp.elevation_continuity(env2, 0.1)
    np.testing.assert_allclose(second.numpy(), g["elevation_continuity_second"], rtol=1e-4, atol=1e-5)
    assert np.abs(g["elevation_continuity_second"]).max() > 0


# ---- the visual cfg module's unwired terms (mushr_visual_env_cfg.py:314-368,400-403) against the reference's own outputs --------

def _make_visual_env(g, pos_key="pos"):
    """a fake env over the golden inputs: state views, wheel-link positions as the reference's test input gave them, and the
    reference singleton's map as `env.traversability`"""
    n = g["pos"].shape[0]
    t = lambda k: torch.from_numpy(g[k].copy())   # noqa: E731
    data = types.SimpleNamespace(root_pos_w=t(pos_key), root_quat_w=t("quat"), root_lin_vel_b=t("lin_vel_b"), body_pos_w=t("body_pos_w"))
    robot = types.SimpleNamespace(data=data, body_names=list(MUSHR_BODY_NAMES))
    robot.find_bodies = types.MethodType(ArticulationView.find_bodies, robot)
    rows, cols = (int(v) for v in g["map_shape"])
    tmap = torch.from_numpy(np.unpackbits(g["map_packed"])[: rows * cols].reshape(rows, cols).astype(np.uint8))
    return types.SimpleNamespace(scene=_Scene(robot, n), num_envs=n, device="cpu", traversability=(tmap, tuple(float(v) for v in g["spacing"])),
                                 common_step_counter=0, max_episode_length=50)


def test_unwired_visual_terms_match_the_reference(golden):
    g = golden("visual_unwired")
    env = _make_visual_env(g)
    env.common_step_counter, env.max_episode_length = int(g["counters"][0]), int(g["counters"][2])
    early = mdp.bool_is_not_traversable(env)
    env.common_step_counter = int(g["counters"][1])
    cases = {
        "bool_is_not_traversable_early": early,
        "bool_is_not_traversable_late": mdp.bool_is_not_traversable(env),
        "is_traversable_speed_scaled": mdp.is_traversable_speed_scaled(env),
        "is_traversable_wheels": mdp.is_traversable_wheels(env),
        "binary_is_traversable_wheels": mdp.binary_is_traversable_wheels(env),
        "vel_rew_trav": mdp.vel_rew_trav(env),
        "vel_rew_trav_2_3": mdp.vel_rew_trav(env, 2.0, 3.0),
        "low_speed_penalty": mdp.low_speed_penalty(env),
        "low_speed_penalty_2": mdp.low_speed_penalty(env, 2.0),
        "roll_over": mdp.roll_over(env),
    }
    for name, got in cases.items():
        want = g[name]
        assert tuple(got.shape) == want.shape, name
        if want.dtype == bool:
            assert got.dtype == torch.bool, name
            np.testing.assert_array_equal(got.numpy(), want, err_msg=name)
        else:
            np.testing.assert_allclose(got.numpy(), want, rtol=1e-6, atol=1e-6, err_msg=name)
    assert not g["bool_is_not_traversable_early"].any() and g["bool_is_not_traversable_late"].any()      # the 1000-episode delay
    assert len(np.unique(g["is_traversable_wheels"])) == 5                  # 0 .. 4 wheels on the path: every mix occurs
    assert 0.2 < g["roll_over"].mean() < 0.8
    # off_track: the drift cfg's function, defined again in the visual module -- here the torch form (non-drift envs)
    env2 = _make_visual_env(g, "pos_track")
    for name, r_out in (("off_track", 2.0), ("off_track_1", 1.0)):
        got = mdp.off_track(env2, 0.8, r_out)
        assert got.dtype == torch.int64
        np.testing.assert_array_equal(got.numpy(), g[name], err_msg=name)
        assert 0 < g[name].mean() < 1


def test_wheel_link_positions_follow_the_vehicle_geometry():
    """ArticulationData.body_pos_w: root pose (+) the wheel-centre offsets the step kernels use (wl_vehicle.h::wheel_contact),
    held against the oracle's rotation matrix; `.*wheel_link` selects bodies 1..4"""
    from oracle.mathlib import matrix_from_quat
    from wheeledlab_amd.envs.managers_cfg import SceneEntityCfg
    from wheeledlab_amd.envs.scene import ArticulationData
    from wheeledlab_amd.params import visual_params
    rng = np.random.RandomState(0)
    n = 64
    q = rng.normal(size=(n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    pos = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    state = torch.zeros(41, n)
    state[0:3], state[3:7] = torch.from_numpy(pos.T), torch.from_numpy(q.T)
    p = visual_params()
    batch = types.SimpleNamespace(state=state, n=n, device="cpu", p=p)
    body = ArticulationData(batch).body_pos_w
    assert body.shape == (n, 5, 3)
    v = p.vehicle
    local = np.array([[0, 0, 0], [-v.half_wheelbase_r, v.half_track, v.wheel_z], [-v.half_wheelbase_r, -v.half_track, v.wheel_z],
                      [v.half_wheelbase_f, v.half_track, v.wheel_z], [v.half_wheelbase_f, -v.half_track, v.wheel_z]], np.float32)
    want = pos[:, None, :] + np.einsum("nij,bj->nbi", matrix_from_quat(q), local)
    np.testing.assert_allclose(body.numpy(), want, rtol=1e-5, atol=1e-5)
    robot = types.SimpleNamespace(body_names=list(MUSHR_BODY_NAMES))
    robot.find_bodies = types.MethodType(ArticulationView.find_bodies, robot)
    cfg = SceneEntityCfg("robot", body_names=".*wheel_link").resolve({"robot": robot})
    assert cfg.body_ids == [1, 2, 3, 4]


def test_cfg_class_lookups_match_the_reference(golden):
    """VisualTerrainImporterCfg.get_map_id / get_traversability (visual/mushr_visual_env_cfg.py:188-208): the cfg class's OWN lookup --
    floor((x + width / 2 - spacing / 2) / spacing), clamped, map[x_idx, y_idx] -- which is NOT the rule of the reward terms' singleton
    (truncation of (x + width / 2 + spacing / 2) / spacing, map[y_idx, x_idx]); against the reference class's outputs on points over
    the whole map, beyond its edge and exactly on the rule's cell lines"""
    from wheeledlab_amd.tasks.visual.mushr_visual_env_cfg import VisualTerrainImporterCfg
    g = golden("visual_unwired")
    m = np.unpackbits(g["map_packed"])[: int(np.prod(g["map_shape"]))].reshape(tuple(g["map_shape"])).astype(bool)
    cfg = VisualTerrainImporterCfg()
    with pytest.raises(ValueError):
        cfg.get_traversability(torch.zeros(1, 2))                     # no map yet: generated when the env is built
    cfg.traversability_hashmap = m.tolist()                           # the reference holds a nested list
    pts = torch.from_numpy(g["cfg_points"].copy())
    xi, yi = cfg.get_map_id(pts[:, 0], pts[:, 1])
    assert xi.dtype == torch.int64
    np.testing.assert_array_equal(xi.numpy(), g["cfg_map_id_x"])
    np.testing.assert_array_equal(yi.numpy(), g["cfg_map_id_y"])
    np.testing.assert_array_equal(np.asarray(cfg.get_traversability(pts)), g["cfg_traversability"])
    assert 0 < g["cfg_traversability"].mean() < 1 and g["cfg_map_id_x"].min() == 0 and g["cfg_map_id_x"].max() == 499
    # and it is a different rule from the singleton's: the two disagree on a good share of the points
    from oracle import visual_mdp as VM
    other = VM.get_traversability(m, g["cfg_points"])
    assert (other != g["cfg_traversability"]).mean() > 0.05


def test_lidar_terms_read_any_sensor_with_linear_depth():
    """lidar_ranges / lidar_ranges_normalized (mdp_sensors/observations.py:25-58): the sensor's `linear_depth` as is; with N(0, 0.1)
    noise, clipped to [min_range, max_range] and mapped to [0, 1]"""
    from wheeledlab_amd.envs.managers_cfg import SceneEntityCfg
    r = torch.rand(256, 360) * 12.0
    sensor = types.SimpleNamespace(data=types.SimpleNamespace(output={"linear_depth": r}), cfg=types.SimpleNamespace(min_range=0.4, max_range=10.0))
    env = types.SimpleNamespace(scene=types.SimpleNamespace(sensors={"lidar": sensor}))
    cfg = SceneEntityCfg("lidar")
    assert mdp.lidar_ranges(env, cfg) is r
    torch.manual_seed(0)
    got = mdp.lidar_ranges_normalized(env, cfg)
    torch.manual_seed(0)
    want = (torch.clip(r + torch.normal(mean=0.0, std=0.1, size=r.shape), min=0.4, max=10.0) - 0.4) / 9.6
    assert torch.equal(got, want) and got.min() == 0.0 and got.max() == 1.0
    inner = (r > 1.0) & (r < 9.0)
    assert abs(float(((got * 9.6 + 0.4) - r)[inner].std()) - 0.1) < 0.005          # the noise is there, at its sigma
