"""Adapters the reference's training script puts around the env (scripts/train_rl.py:73-93)."""
from __future__ import annotations

import torch


class ClipAction:
    """wheeledlab_rl/utils/clip_action.py:5-27 work-alike.  The clip itself is folded into the fused kernel's action
    stage (`env.set_clip_actions`), so the wrapper adds no launch; bounds other than +-1 fall back to torch.clip."""

    def __init__(self, env):
        self.env = env
        lo, hi = env.action_space.low, env.action_space.high
        self._fused = (float(lo), float(hi)) == (-1.0, 1.0)
        if self._fused:
            env.unwrapped.set_clip_actions(True)

    def __getattr__(self, name):
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def action(self, action):
        if self._fused:
            return action
        return torch.clip(action, min=self.env.action_space.low, max=self.env.action_space.high)

    def step(self, action):
        return self.env.step(self.action(action))

    def reset(self, **kw):
        return self.env.reset(**kw)


class RslRlVecEnvWrapper:
    """isaaclab_rl.rsl_rl.RslRlVecEnvWrapper work-alike: the members the reference's runner touches
    (utils/modified_rsl_rl_runner.py:43-109): get_observations, step -> (obs, rew, dones, infos), num_envs,
    episode_length_buf (assignable), max_episode_length, cfg, device, unwrapped."""

    def __init__(self, env):
        from ..envs import ManagerBasedRLEnv
        if not isinstance(env.unwrapped, ManagerBasedRLEnv):
            raise ValueError("RslRlVecEnvWrapper expects a ManagerBasedRLEnv")
        self.env = env
        self.num_envs = self.unwrapped.num_envs
        self.device = self.unwrapped.device
        self.max_episode_length = self.unwrapped.max_episode_length
        self.num_actions = self.unwrapped.action_manager.total_action_dim
        self.num_obs = self.unwrapped.observation_manager.group_obs_dim["policy"][0]
        self.num_privileged_obs = 0
        self.env.reset()

    @property
    def cfg(self):
        return self.unwrapped.cfg

    @property
    def unwrapped(self):
        return self.env.unwrapped

    @property
    def episode_length_buf(self):
        return self.unwrapped.episode_length_buf

    @episode_length_buf.setter
    def episode_length_buf(self, value):
        self.unwrapped.episode_length_buf = value

    def seed(self, seed: int = -1) -> int:
        return self.unwrapped.seed(seed)

    def reset(self):
        obs, _ = self.env.reset()
        return obs["policy"], {"observations": obs}

    def get_observations(self):
        obs = self.unwrapped.observation_manager.compute()
        return obs["policy"], {"observations": obs}

    def step(self, actions):
        obs, rew, terminated, truncated, extras = self.env.step(actions)
        dones = self.unwrapped._batch.dones   # int64 terminated | truncated, written by the step kernel
        extras["observations"] = obs
        if not self.unwrapped.cfg.is_finite_horizon:
            extras["time_outs"] = truncated
        return obs["policy"], rew, dones, extras

    def close(self):
        return self.env.close()
