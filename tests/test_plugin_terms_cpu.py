"""the reward functions the reference's elevation cfg module defines without registering them
(mushr_elevation_env_cfg.py:159-164,175-231,256-266) exist in the build as torch terms on the env's state views
(wheeledlab_amd/envs/mdp.py) so that a config override can wire them in; here they are held against the outputs of the
reference's own functions (tests/golden*/elevation_unwired.npz, made by tests/golden/gen_golden.py)."""
import types

import numpy as np
import torch

from wheeledlab_amd.envs import mdp
from wheeledlab_amd.envs.scene import MUSHR_JOINT_NAMES, ArticulationView


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
