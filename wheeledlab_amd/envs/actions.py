"""Action term configs + a thin action-term object.  Mirrors the public surface of the reference's
`wheeledlab.envs.mdp.actions` (actions_cfg.py:15-67, ackermann_actions.py:103-145): `action_dim`, `raw_actions`,
`processed_actions`, `process_actions`, `apply_actions`.  The arithmetic runs in csrc (fused step kernel, or
`wl_action_map` when this object is used directly)."""
from __future__ import annotations

import ctypes as C

import torch

from .. import _abi as A
from .configclass import MISSING, configclass
from .managers_cfg import ActionTermCfg


class AckermannAction:
    # the base class' true-Ackermann map (ackermann_actions.py:150-201; no registered task selects it): steer joints take
    # atan(L / (R -+ W / 2)), wheels the 4WD speeds
    wl_map = 2

    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env
        self._asset = env.scene[cfg.asset_name]
        self._wheel_ids, self._wheel_names = self._asset.find_joints(cfg.wheel_joint_names)
        self._steering_ids, self._steering_names = self._asset.find_joints(cfg.steering_joint_names)
        self._raw_actions = torch.zeros(env.num_envs, 2, device=env.device)
        self._processed_actions = torch.zeros(env.num_envs, 2, device=env.device)
        self._steer_target = torch.zeros(env.num_envs, 2, device=env.device)
        self._wheel_target = torch.zeros(env.num_envs, 4, device=env.device)

    @property
    def action_dim(self) -> int:
        return 2

    @property
    def raw_actions(self):
        return self._raw_actions

    @property
    def processed_actions(self):
        return self._processed_actions

    def params(self, clip_wrapper: bool = False) -> A.WlActionParams:
        c = self.cfg
        if type(self).wl_map is None:
            raise NotImplementedError(f"{type(self).__name__}: no HIP joint-target map (use RCCarRWDAction / RCCar4WDAction)")
        a = A.WlActionParams()
        a.scale[0], a.scale[1] = c.scale
        a.offset[0], a.offset[1] = c.offset
        a.bounding = {"clip": 1, "tanh": 2, None: 0}[c.bounding_strategy]
        a.no_reverse, a.clip_wrapper, a.map = int(c.no_reverse), int(clip_wrapper), type(self).wl_map
        a.base_length, a.base_width, a.wheel_radius = c.base_length, c.base_width, c.wheel_radius
        return a

    def process_actions(self, actions):
        self._raw_actions[:] = actions
        self._run(actions)

    def apply_actions(self):
        """joint targets for the current processed actions (inspection only: the fused kernel applies them itself)"""
        return self._steer_target, self._wheel_target

    def _run(self, actions):
        env = self._env
        ap = self.params(False)
        a = actions.to(torch.float32).contiguous()
        A.check(env._batch.lib.wl_action_map(C.byref(ap), env.num_envs, a.data_ptr(), self._processed_actions.data_ptr(),
                                             self._steer_target.data_ptr(), self._wheel_target.data_ptr(),
                                             env._batch._stream()), "wl_action_map")


class RCCarRWDAction(AckermannAction):
    """rear-wheel drive, tan steering (rc_car_actions.py:6-29)"""
    wl_map = 0


class RCCar4WDAction(AckermannAction):
    """4WD with Ackermann-adjusted wheel speeds, tan steering (rc_car_actions.py:33-64)"""
    wl_map = 1


@configclass
class AckermannActionCfg(ActionTermCfg):
    class_type = AckermannAction
    wheel_joint_names: list = MISSING
    steering_joint_names: list = MISSING
    scale: tuple = (1.0, 1.0)
    offset: tuple = (0.0, 0.0)
    bounding_strategy = "tanh"
    base_length: float = 1.0
    base_width: float = 1.0
    wheel_radius: float = 1.0
    no_reverse: bool = False


@configclass
class RCCarRWDActionCfg(AckermannActionCfg):
    class_type = RCCarRWDAction


@configclass
class RCCar4WDActionCfg(AckermannActionCfg):
    class_type = RCCar4WDAction
