"""ManagerBasedRLEnv -- the drop-in host surface of the fused env.step().

Same constructor and attribute names as `isaaclab.envs.ManagerBasedRLEnv`, the class the reference registers its
tasks with (wheeledlab_tasks/__init__.py:14-63) and the callers rely on (scripts/train_rl.py:70-116,
utils/modified_rsl_rl_runner.py:47-109): `step/reset/seed/close`, `num_envs`, `device`, `max_episode_length`,
`common_step_counter`, `episode_length_buf`, `reward_manager.get_term_cfg/set_term_cfg`, `scene[...]`,
`action_space`, `unwrapped.single_action_space`, `cfg`, `extras["log"]`.

step() is ONE launch of the fused HIP kernel; the managers below are thin views/config holders, not executors."""
from __future__ import annotations

import os

import ctypes as C
import math

import torch

from .. import _abi as A
from ..core import DriftBatch, ElevBatch, VisualBatch, VisualDepthBatch
from .configclass import fields_of
from .flatten import flatten_cfg
from .scene import SceneView


class Box:
    """minimal gymnasium.spaces.Box stand-in (gymnasium is not a dependency): low/high are assignable, as
    scripts/train_rl.py:73-74 does"""

    def __init__(self, low, high, shape, dtype=torch.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    def sample(self, generator=None, device="cpu"):
        lo = -1.0 if not math.isfinite(float(self.low)) else float(self.low)
        hi = 1.0 if not math.isfinite(float(self.high)) else float(self.high)
        return torch.rand(self.shape, generator=generator, device=device) * (hi - lo) + lo

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape})"


class RewardManager:
    def __init__(self, env, flat):
        self._env = env
        self._cfgs = dict((k, v) for k, v in fields_of(env.cfg.rewards) if hasattr(v, "func")) if env.cfg.rewards else {}
        self._slots = dict(flat.reward_names)
        self._custom = [n for n, _ in flat.custom_rewards]

    @property
    def active_terms(self):
        return list(self._cfgs)

    def get_term_cfg(self, name):
        return self._cfgs[name]

    def set_term_cfg(self, name, cfg):
        self._cfgs[name] = cfg
        if name in self._slots:  # push the new weight into the kernel's parameter block
            self._env._batch.p.weight[self._slots[name]] = float(cfg.weight)

    @property
    def episode_sums(self):
        b = self._env._batch
        return {n: b.state[A.S_EPSUM0 + s, : b.n] for n, s in self._slots.items()}

    _episode_sums = episode_sums


class TerminationManager:
    def __init__(self, env, flat):
        self._env, self._names = env, flat.termination_names

    @property
    def active_terms(self):
        return list(self._names.values()) + [n for n, _ in self._env._custom_term]

    @property
    def terminated(self):
        return self._env._batch.terminated

    @property
    def time_outs(self):
        return self._env._batch.truncated

    @property
    def dones(self):
        return self.terminated | self.time_outs

    def get_term(self, name):
        if name in self._env._custom_flags:          # a torch-fallback term: its value at the last step
            return self._env._custom_flags[name]
        if name == self._names.get("time_out"):
            return self.time_outs
        slots = [k for k, v in self._names.items() if v == name and k != "time_out"]
        if not slots:
            raise KeyError(name)
        if self._env._task == "elevation":      # several `terminated` terms: re-evaluate the one asked for
            return self._env._eval_elev_terms()["flags"][slots[0]]
        return self.terminated


class ActionManager:
    def __init__(self, env):
        self._env = env
        name, acfg = [(k, v) for k, v in fields_of(env.cfg.actions) if hasattr(v, "class_type")][0]
        self._terms = {name: acfg.class_type(acfg, env)}
        self.prev_action = torch.zeros(env.num_envs, 2, device=env.device)

    @property
    def total_action_dim(self):
        return 2

    @property
    def action(self):
        b = self._env._batch
        return b.state[A.S_ACT0:A.S_ACT0 + 2, : b.n].T

    def get_term(self, name):
        return self._terms[name]

    @property
    def active_terms(self):
        return list(self._terms)


class CommandManager:
    """elevation task: UniformPose2dCommand "goal_pose" -> [N, 4] (x_b, y_b, z_b, heading_b) (isaaclab, unpinned)"""

    def __init__(self, env):
        self._env = env
        self.active_terms = ["goal_pose"] if env._task == "elevation" else []

    def get_command(self, name):
        if name != "goal_pose" or self._env._task != "elevation":
            raise KeyError(name)
        b = self._env._batch
        s = b.state
        q = s[A.S_QW:A.S_QW + 4, : b.n]
        yaw = torch.atan2(2.0 * (q[0] * q[3] + q[1] * q[2]), 1.0 - 2.0 * (q[2] * q[2] + q[3] * q[3]))
        hb = torch.remainder(s[A.S_TGT_H, : b.n] - yaw + math.pi, 2 * math.pi) - math.pi
        return torch.stack([s[A.S_CMD_BX, : b.n], s[A.S_CMD_BY, : b.n], b.p.reset_z - s[A.S_PZ, : b.n], hb], dim=-1)


class ObservationManager:
    def __init__(self, env):
        self._env = env
        self.group_obs_dim = {"policy": (env._batch.OBS_DIM,)}      # grown by the env once its custom terms are sized
        self.active_terms = {"policy": [k for k, v in fields_of(env.cfg.observations.policy) if hasattr(v, "func")]}

    def compute(self):
        return {"policy": self._env._with_custom_obs(self._env._batch.observe())}


def episode_log_keys(reward_slots, term_names):
    """key -> (kind, metric index) table of an env's extras["log"] (built once per env)"""
    keys = {f"Episode_Reward/{n}": ("r", s) for n, s in reward_slots.items()}
    if "time_out" in term_names:
        keys[f"Episode_Termination/{term_names['time_out']}"] = ("c", A.M_TIMEOUTS)
    for k in range(4):
        if k in term_names:
            keys[f"Episode_Termination/{term_names[k]}"] = ("c", A.M_TERM0 + k)
    keys["Metrics/resets"] = ("c", A.M_RESETS)
    keys["Metrics/nonfinite_envs"] = ("c", A.M_NONFINITE)
    return keys


class EpisodeLog(dict):
    """extras["log"]: IsaacLab's per-reset statistics (`Episode_Reward/<term>` = mean over the envs that were reset of
    episode_sum / episode_length_s; `Episode_Termination/<term>` = count), evaluated LAZILY from one slot of the
    device-side metric ring so that stepping never synchronises with the host (and costs no torch launch unless a
    value is read).  Values are 0-dim device tensors; with no reset in the step the means are NaN (IsaacLab would
    omit the keys: use nanmean, or cfg.sync_episode_log)."""
    __slots__ = ("_metrics", "_idx", "_len_s", "_keys")

    def __init__(self, metrics, idx, keys, episode_length_s, extra=None):
        super().__init__()
        self._metrics, self._idx, self._keys, self._len_s = metrics, idx, keys, episode_length_s
        if extra:       # torch-fallback terms: ready-made device scalars
            self._keys = dict(keys)
            for k, v in extra.items():
                self._keys[k] = ("x", v)

    def __missing__(self, key):
        kind, i = self._keys[key]
        if kind == "x":
            self[key] = i
            return i
        m = self._metrics if self._idx is None else self._metrics[self._idx]
        if m.dim() == 2:                       # raw accumulator [WL_M_SHARDS][WL_M_COUNT]: fold the shards once
            m = m.sum(0)
            self._metrics, self._idx = m, None
        v = m[i] if kind == "c" else m[A.M_EPSUM0 + i] / m[A.M_RESETS] / self._len_s
        self[key] = v
        return v

    def __contains__(self, key):
        return key in self._keys

    def __iter__(self):
        return iter(self._keys)

    def keys(self):
        return self._keys.keys()

    def items(self):
        return [(k, self[k]) for k in self._keys]

    def __len__(self):
        return len(self._keys)


class ManagerBasedRLEnv:
    metadata = {"render_modes": [None]}
    is_vector_env = True

    def __init__(self, cfg, render_mode=None, **kwargs):
        self.cfg = cfg
        self.render_mode = render_mode
        self.num_envs = int(cfg.scene.num_envs)
        self.device = torch.device(cfg.sim.device)
        flat = flatten_cfg(cfg)
        self._flat = flat
        self._task = flat.task
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()
        seed = 42 if cfg.seed is None else int(cfg.seed)
        common = dict(device=self.device, params=flat.params, seed=seed, env_offset=rank * self.num_envs,
                      metrics_slots=int(cfg.metrics_slots), startup=flat.startup)
        if flat.task == "elevation":
            self._batch = ElevBatch(self.num_envs, heightfield=flat.extra.get("heightfield"), **common)
        elif flat.task in ("visual", "visual_depth"):
            x = flat.extra
            kw = dict(trav_map=x["map"], spacing=x["spacing"], map_kwargs=dict(map_size=x["map_size"], env_size=x["env_size"],
                                                                              sub_group_size=x["group"], num_walkers=x["walkers"]))
            if flat.task == "visual_depth":
                self._batch = VisualDepthBatch(self.num_envs, heightfield=x.get("heightfield"), max_depth=x["max_depth"], **kw, **common)
            else:
                self._batch = VisualBatch(self.num_envs, **kw, **common)
        else:
            self._batch = DriftBatch(self.num_envs, **common)
        self.scene = SceneView(self._batch, cfg.scene, task=flat.task)
        # plugin terms without a HIP implementation: any f(env, **params) evaluates with torch on the state views
        self._custom_rew, self._custom_term, self._custom_obs = list(flat.custom_rewards), list(flat.custom_terminations), list(flat.custom_obs)
        self._custom_flags = {n: torch.zeros(self.num_envs, dtype=torch.bool, device=self.device) for n, _ in self._custom_term}
        self._custom_epsum = {n: torch.zeros(self.num_envs, device=self.device) for n, _ in self._custom_rew}
        self._custom_log = {}
        self.common_step_counter = 0
        self.step_dt = cfg.sim.dt * cfg.decimation
        self.physics_dt = cfg.sim.dt
        self.max_episode_length_s = cfg.episode_length_s
        self.max_episode_length = math.ceil(cfg.episode_length_s / self.step_dt)
        self.action_manager = ActionManager(self)
        self.command_manager = CommandManager(self)
        self.observation_manager = ObservationManager(self)
        self.reward_manager = RewardManager(self, flat)
        self.termination_manager = TerminationManager(self, flat)
        self._event_terms = {}
        for name, term in ([(k, v) for k, v in fields_of(cfg.events) if hasattr(v, "func")] if cfg.events else []):
            if isinstance(term.func, type):  # class-type terms are instantiated with (cfg, env), as IsaacLab does
                self._event_terms[name] = term.func(term, self)
        self.single_action_space = Box(-math.inf, math.inf, (2,))
        self.action_space = Box(-math.inf, math.inf, (self.num_envs, 2))
        self.single_observation_space = {"policy": Box(-math.inf, math.inf, (self._batch.OBS_DIM,))}
        self.observation_space = {"policy": Box(-math.inf, math.inf, (self.num_envs, self._batch.OBS_DIM))}
        self.extras = {}
        self.obs_buf = {}
        self._log_keys = episode_log_keys(self.reward_manager._slots, flat.termination_names)
        # "torch terms run between the kernel launches": the fused collectors (one launch per rollout / writing straight
        # into the runner's storage) are off whenever any kind of custom term is registered
        self._has_custom_rewards = bool(self._custom_rew or self._custom_term or self._custom_obs)
        self._obs_dim = self._batch.OBS_DIM
        if self._custom_obs:
            # shape probe on the un-reset state: nothing it renders may be kept (the scene camera caches per step)
            cams = [s.data for s in self.scene.sensors.values() if hasattr(s.data, "caching")]
            for c in cams:
                c.caching = False
            self._obs_dim += sum(self._eval_obs_term(t).shape[1] for _, t in self._custom_obs)
            for c in cams:
                c.caching = True
            self.observation_manager.group_obs_dim["policy"] = (self._obs_dim,)
            self.single_observation_space = {"policy": Box(-math.inf, math.inf, (self._obs_dim,))}
            self.observation_space = {"policy": Box(-math.inf, math.inf, (self.num_envs, self._obs_dim))}
        self._has_curriculum = bool(flat.curriculum)
        self._clip_actions = False
        self._sim_step_counter = 0

    # ---- gym.Env surface --------------------------------------------------------------------------------------
    @property
    def unwrapped(self):
        return self

    @property
    def episode_length_buf(self):
        return self._batch.episode_len[: self.num_envs]

    @episode_length_buf.setter
    def episode_length_buf(self, value):  # the RSL-RL runner assigns it for init_at_random_ep_len
        self._batch.episode_len[: self.num_envs] = value.to(torch.int32)

    @property
    def traversability(self):
        """(map [rows, cols] uint8 on the device, (row_spacing, col_spacing)) of the visual tasks -- what the reference keeps in
        its TraversabilityHashmapUtil singleton (visual/utils/traversability_utils.py:57-63); read by the torch terms of mdp.py"""
        b = self._batch
        if not hasattr(b, "trav_map"):
            raise AttributeError(f"the {self._task} task has no traversability map")
        return b.trav_map, (float(b._map.row_spacing), float(b._map.col_spacing))

    @property
    def reward_buf(self):
        return self._batch.reward

    @property
    def reset_buf(self):
        return self._batch.terminated | self._batch.truncated

    def seed(self, seed: int = -1) -> int:
        if seed is not None and seed >= 0:
            self._batch.seed = int(seed)
            torch.manual_seed(seed)
        return self._batch.seed

    def set_clip_actions(self, on: bool = True):
        """fold the ClipAction wrapper (wheeledlab_rl/utils/clip_action.py:27) into the kernel's action stage"""
        self._clip_actions = bool(on)
        self._batch.p.action.clip_wrapper = int(on)

    def reset(self, seed=None, options=None):
        if seed is not None:
            self.seed(seed)
        self._batch.reset()
        self.extras = {}
        for t in self._custom_epsum.values():
            t.zero_()
        self.obs_buf = {"policy": self._with_custom_obs(self._batch.observe())}
        return self.obs_buf, self.extras

    def step(self, action: torch.Tensor):
        b = self._batch
        self.action_manager.prev_action = action
        slot = b.step_count % b.metrics_slots if b.metrics_slots > 1 else None   # this step's slot of the metric ring
        if self._task == "visual" and self._flat.extra.get("augment"):
            b.sample_augmentation()       # one ColorJitter / GaussianBlur draw per call, as torchvision does for a batch
        obs, rew, terminated, truncated = b.step(action)
        self.common_step_counter += 1
        self._sim_step_counter += self.cfg.decimation
        if self._has_custom_rewards:
            obs, rew, terminated, truncated = self._apply_custom_terms(obs, rew, terminated, truncated, slot)
        # curriculum: evaluated inside _reset_idx in IsaacLab, i.e. on steps where >= 1 env resets; every built-in
        # term is a no-op off episode boundaries, so the (synchronising) any() runs once per max_episode_length steps
        if self._has_curriculum and self.common_step_counter % self.max_episode_length == 0:
            if bool((terminated | truncated).any()):
                for name, term in self._flat.curriculum:
                    term.func(self, None, **term.params)
        if self.cfg.sync_episode_log:
            if bool((terminated | truncated).any()):
                self.extras["log"] = dict(self._episode_log(slot).items())
            else:
                self.extras.pop("log", None)
        else:
            self.extras["log"] = self._episode_log(slot)
        self.obs_buf = {"policy": obs}
        return self.obs_buf, rew, terminated, truncated, self.extras

    def collect_step(self, actor_critic, storage, k: int):
        """One { policy step -> env.step } of the runner's collection loop (modified_rsl_rl_runner.py:70-80) with every
        output written IN PLACE into the storage: actions / mu / log-prob / value of row k from observation row k
        (wl_actor_critic_act, any observation width), then the fused step's observation into row k + 1 and reward / flags /
        dones into row k -- two or three launches per step, no copies.  Counters, curriculum and the episode log behave as
        in step(); call finish_collection() after the last step."""
        if self._has_custom_rewards:
            raise ValueError("custom (torch) reward terms run between steps; use step()")
        b = self._batch
        a = storage.actions[k]
        if self._task == "elevation" and b.n <= 32768 and os.environ.get("WL_ELEV_COLLECT", "0") == "1":
            # policy step, env.step() and the height scan in ONE launch (wl_elev_collect_step): measured 5 % SLOWER than the
            # two launches below (the 16-row blocks double the first-layer operand traffic), so it is opt-in
            b.collect_step(actor_critic, storage, k)
            self.action_manager.prev_action = a
            self.common_step_counter += 1
            self._sim_step_counter += self.cfg.decimation
            if self._has_curriculum and self.common_step_counter % self.max_episode_length == 0:
                if bool(storage.dones[k].any()):
                    for name, term in self._flat.curriculum:
                        term.func(self, None, **term.params)
            return
        # (running only the actor's half here and the critic's on a side stream next to the env's launches was measured
        # SLOWER -- elevation 7.1e7 -> 5.8e7 env-steps/s: the event / stream hand-off per step costs more than the overlap gains)
        # (planes_fresh: the parameters only change between collections, so the bf16 form's weight planes built at k = 0 hold)
        actor_critic.act(storage.observations[k], a, storage.mu[k], storage.actions_log_prob[k], storage.values[k], b.seed,
                         b.step_count, b.env_offset, planes_fresh=k > 0)
        self.action_manager.prev_action = a
        if self._task == "visual" and self._flat.extra.get("augment"):
            b.sample_augmentation()
        b.rollout(storage.actions[k:k + 1], storage.observations[k + 1:k + 2], storage.rewards[k:k + 1],
                  storage.terminated[k:k + 1], storage.time_outs[k:k + 1], dones_out=storage.dones[k:k + 1])
        self.common_step_counter += 1
        self._sim_step_counter += self.cfg.decimation
        if self._has_curriculum and self.common_step_counter % self.max_episode_length == 0:
            if bool(storage.dones[k].any()):
                for name, term in self._flat.curriculum:
                    term.func(self, None, **term.params)

    def can_collect_rollout(self) -> bool:
        """the whole collection loop as one launch per curriculum segment (collect_rollout): elevation task, quad form"""
        return (self._task == "elevation" and self._batch.n <= 32768 and not self._has_custom_rewards
                and os.environ.get("WL_ELEV_PERSISTENT_COLLECT", "1") != "0")

    def collect_rollout(self, actor_critic, storage):
        """storage.n_steps x { actor -> sample -> env.step } of the runner's collection loop (modified_rsl_rl_runner.py:70-80) in ONE
        launch per curriculum segment (wl_elev_collect_rollout: the actor's first layer in the blocks' registers, the
        observation rows in LDS); observation row 0 of the storage must hold the current observation.  The critic's values
        are NOT filled in (not needed to step): evaluate storage.observations in one batched pass afterwards.  The rollout is
        cut at curriculum boundaries exactly as rollout_policy() cuts the drift task's; call finish_collection() after it."""
        if not self.can_collect_rollout():
            raise NotImplementedError("the persistent collector exists for the elevation task (quad form, built-in reward terms)")
        b, K, k = self._batch, storage.n_steps, 0
        while k < K:
            seg = K - k
            if self._has_curriculum:
                seg = min(seg, self.max_episode_length - self.common_step_counter % self.max_episode_length)
            b.collect_rollout(actor_critic, storage, start=k, count=seg)
            k += seg
            self.common_step_counter += seg
            self._sim_step_counter += seg * self.cfg.decimation
            if self._has_curriculum and self.common_step_counter % self.max_episode_length == 0:
                if bool(storage.dones[k - 1].any()):
                    for name, term in self._flat.curriculum:
                        term.func(self, None, **term.params)
        self.action_manager.prev_action = storage.actions[K - 1]

    def finish_collection(self, storage, n_steps: int | None = None):
        """after the last collect_step(): the env's own observation buffer and episode log catch up with the storage"""
        b, K = self._batch, storage.n_steps if n_steps is None else n_steps
        b.obs.copy_(storage.observations[K])
        self.obs_buf = {"policy": b.obs}
        self.extras["log"] = self._episode_log(None) if b.metrics_slots == 1 else EpisodeLog(
            self.episode_metrics(window=K, reduce_ranks=False), None, self._log_keys, self.max_episode_length_s)

    def rollout_policy(self, actor_critic, storage):
        """storage.n_steps x { actor -> sample -> env.step } in fused launches (drift task): the collection loop of the
        reference's runner (modified_rsl_rl_runner.py:70-80).  `actor_critic` exposes `.actor`, `.critic` (policy.Mlp)
        and `.std`.  The rollout is cut at curriculum boundaries (common_step_counter % max_episode_length == 0), where
        IsaacLab's _reset_idx would evaluate the curriculum terms, so reward weights change on the same step as they
        do when stepping."""
        if self._task != "drift":
            raise NotImplementedError("fused policy rollouts exist for the drift task (14-dim observation)")
        if self._has_custom_rewards:
            raise ValueError("custom (torch) reward terms run between steps; use step()")
        b, K, k = self._batch, storage.n_steps, 0
        while k < K:
            seg = K - k
            if self._has_curriculum:
                seg = min(seg, self.max_episode_length - self.common_step_counter % self.max_episode_length)
            b.rollout_policy(actor_critic, storage, evaluate_critic=False, start=k, count=seg)
            k += seg
            self.common_step_counter += seg
            self._sim_step_counter += seg * self.cfg.decimation
            if self._has_curriculum and self.common_step_counter % self.max_episode_length == 0:
                if bool(storage.dones[k - 1].any()):
                    for name, term in self._flat.curriculum:
                        term.func(self, None, **term.params)
        storage.values.copy_(actor_critic.critic(storage.observations).squeeze(-1))
        self.obs_buf = {"policy": b.obs}
        self.extras["log"] = self._episode_log(None) if b.metrics_slots == 1 else EpisodeLog(
            self.episode_metrics(window=K, reduce_ranks=False), None, self._log_keys, self.max_episode_length_s)
        return storage

    def episode_log_summary(self, window: int, reduce_ranks: bool = False) -> dict:
        """host floats of the episode log over the last `window` steps (one device->host copy; the runner's logging);
        reduce_ranks: over all ranks' envs (every rank must call it: one all-reduce)"""
        m = self.episode_metrics(window=window, reduce_ranks=reduce_ranks).tolist()
        resets = max(m[A.M_RESETS], 1.0)
        out = {}
        for key, (kind, i) in self._log_keys.items():
            out[key] = m[i] if kind == "c" else m[A.M_EPSUM0 + i] / resets / self.max_episode_length_s
        if self._custom_log:      # terms that run as torch behind the kernel: their latest episode means / counts -- ONE stacked copy
            keys = list(self._custom_log)
            vals = torch.stack([torch.as_tensor(self._custom_log[k], dtype=torch.float32, device=self.device).reshape(()) for k in keys])
            if reduce_ranks and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
                # rank-local means / counts -> the mean over ranks (counts: the sum), like the kernel's keys above
                counts = torch.tensor([k.startswith("Episode_Termination/") for k in keys], device=self.device)
                torch.distributed.all_reduce(vals)
                vals = torch.where(counts, vals, vals / torch.distributed.get_world_size())
            out.update(zip(keys, vals.tolist()))
        return out

    def _episode_log(self, slot):
        return EpisodeLog(self._batch.metrics_raw, 0 if slot is None else slot, self._log_keys, self.max_episode_length_s,
                          self._custom_log)

    # ---- torch fallback for terms without a HIP implementation (SURVEY.md 8(b) "term plugin API") -------------------
    def _eval_obs_term(self, term) -> torch.Tensor:
        """ObservationManager semantics for one term: f(env, **params) -> noise (if corruption is on) -> clip -> scale"""
        v = term.func(self, **term.params).to(torch.float32)
        if v.dim() == 1:
            v = v.unsqueeze(-1)
        v = v.reshape(self.num_envs, -1)
        n = getattr(term, "noise", None)
        if n is not None and getattr(self.cfg.observations.policy, "enable_corruption", False):
            if hasattr(n, "std"):
                v = v + n.mean + n.std * torch.randn_like(v)
            else:
                v = v + n.n_min + (n.n_max - n.n_min) * torch.rand_like(v)
        if getattr(term, "clip", None) is not None:
            v = v.clamp(term.clip[0], term.clip[1])
        if getattr(term, "scale", None) is not None:
            v = v * term.scale
        return v

    def _with_custom_obs(self, obs: torch.Tensor) -> torch.Tensor:
        if not self._custom_obs:
            return obs
        return torch.cat([obs] + [self._eval_obs_term(t) for _, t in self._custom_obs], dim=-1)

    def _apply_custom_terms(self, obs, rew, terminated, truncated, slot):
        """The fused kernel has already stepped, rewarded, terminated and RESET the envs its built-in terms ended.  Custom
        terms see the post-step state of every env that is still running (exactly what IsaacLab would show them); envs
        that a built-in term ended this step are already at their reset pose, so custom rewards / terminations are
        masked off for them (their last-step custom reward is the one approximation of this path).  A custom
        termination ends the episode through a masked wl_*_reset launch; its episode sums go to the metric ring first.
        Everything is device-side torch: no host synchronisation."""
        b, n = self._batch, self.num_envs
        live = ~(terminated | truncated)
        zero = torch.zeros((), device=self.device)
        for name, term in self._custom_rew:
            if term.weight == 0.0:       # RewardManager skips zero-weight terms
                continue
            val = torch.where(live, term.func(self, **term.params).to(torch.float32) * (term.weight * self.step_dt), zero)
            rew += val
            self._custom_epsum[name] += val
        newly = None
        if self._custom_term:
            c_term = torch.zeros(n, dtype=torch.bool, device=self.device)
            c_to = torch.zeros_like(c_term)
            for name, term in self._custom_term:
                flag = term.func(self, **term.params).to(torch.bool) & live
                self._custom_flags[name] = flag
                if getattr(term, "time_out", False):
                    c_to |= flag
                else:
                    c_term |= flag
            newly = c_term | c_to
            m = newly.to(torch.float32)
            ring = b.metrics_raw[0 if slot is None else slot, 0]                 # shard 0 of this step's slot
            ring[A.M_EPSUM0:A.M_EPSUM0 + A.WL_MAX_REW_TERMS] += (b.state[A.S_EPSUM0:A.S_EPSUM0 + A.WL_MAX_REW_TERMS, :n] * m).sum(1)
            ring[A.M_RESETS] += m.sum()
            ring[A.M_TIMEOUTS] += c_to.sum()
            ring[A.M_EPLEN] += (b.episode_len[:n].to(torch.float32) * m).sum()
            terminated |= c_term          # in place: the batch's own flag buffers (reset_buf, the RSL-RL wrapper's dones)
            truncated |= c_to
            b.dones |= newly.to(b.dones.dtype)
            kept = obs.clone()
            b.reset(newly)                                                        # masked launch (no-op without flags)
            obs.copy_(torch.where(newly.unsqueeze(-1), b.observe().clone(), kept))   # fresh observation for those envs only
        done = ~live if newly is None else (~live | newly)
        cnt = done.sum().to(torch.float32)
        prev, self._custom_log = self._custom_log, {}
        for name, acc in self._custom_epsum.items():
            # IsaacLab refreshes these keys only inside _reset_idx: on a step where no episode ends the previous value stands
            # (0 before the first one) -- never 0 / 0
            key = f"Episode_Reward/{name}"
            fresh = (acc * done).sum() / cnt.clamp_min(1.0) / self.max_episode_length_s
            self._custom_log[key] = torch.where(cnt > 0, fresh, torch.as_tensor(prev.get(key, 0.0), dtype=fresh.dtype, device=fresh.device))
            acc *= ~done
        for name, flag in self._custom_flags.items():
            self._custom_log[f"Episode_Termination/{name}"] = flag.sum()
        return self._with_custom_obs(obs), rew, terminated, truncated

    def episode_metrics(self, window: int | None = None, reduce_ranks: bool = True):
        """aggregate of the last `window` per-step metric slots as one [WL_M_COUNT] vector; across ranks it is ONE
        sum all-reduce (RCCL over xGMI) -- the only collective of the env-sharded multi-GPU path"""
        b = self._batch
        if b.metrics_slots > 1:
            R = b.metrics_slots
            w = min(window or R - 1, R - 1, b.step_count)
            idx = [(b.step_count - 1 - i) % R for i in range(w)]
            m = b.metrics_raw[idx].sum((0, 1)) if idx else torch.zeros(A.M_COUNT, device=self.device)
        else:
            m = b.metrics
        if reduce_ranks and torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.all_reduce(m)
        return m

    def render(self):
        return None

    def close(self):
        self._batch = None

    def _soa(self, t):
        b = self._batch
        out = torch.zeros(t.shape[1], b.stride, device=self.device)
        out[:, : b.n] = t.T
        return out

    def _eval_elev_terms(self):
        """elevation mdp terms through wl_elev_mdp on the current state -> dict(terms [4,N], flags [4,N] bool, goal_rel [N,2])"""
        b, d = self._batch, self.scene["robot"].data
        n, stride = b.n, b.stride
        ins = [self._soa(t) for t in (d.root_pos_w - self.scene.env_origins, d.root_quat_w, d.root_lin_vel_b, d.root_lin_vel_w,
                                      d.joint_vel[:, 2:6], b.state[A.S_CMD_BX:A.S_CMD_BX + 2, :n].T)]
        terms = torch.zeros(4, stride, device=self.device)
        flags = torch.zeros(4, stride, dtype=torch.bool, device=self.device)
        goal = torch.zeros(2, stride, device=self.device)
        A.check(b.lib.wl_elev_mdp(C.byref(b.p), n, stride, *[t.data_ptr() for t in ins], b.truncated.data_ptr(), 0, None,
                                  None, terms.data_ptr(), flags.data_ptr(), goal.data_ptr(), None, b._stream()), "wl_elev_mdp")
        return dict(terms=terms[:, :n], flags=flags[:, :n], goal_rel=goal[:, :n].T)

    def _eval_visual_terms(self):
        """visual mdp terms through wl_visual_mdp on the current state"""
        b, d = self._batch, self.scene["robot"].data
        n, stride = b.n, b.stride
        pos, vb = self._soa(d.root_pos_w - self.scene.env_origins), self._soa(d.root_lin_vel_b)
        terms = torch.zeros(2, stride, device=self.device)
        oom = torch.zeros(n, dtype=torch.bool, device=self.device)
        xi = torch.zeros(n, dtype=torch.int32, device=self.device)
        yi = torch.zeros(n, dtype=torch.int32, device=self.device)
        A.check(b.lib.wl_visual_mdp(C.byref(b.p), C.byref(b._map), n, stride, pos.data_ptr(), vb.data_ptr(), terms.data_ptr(),
                                    oom.data_ptr(), xi.data_ptr(), yi.data_ptr(), b._stream()), "wl_visual_mdp")
        return dict(terms=terms[:, :n], out_of_map=oom, x_idx=xi, y_idx=yi)

    # ---- plugin support: built-in mdp terms evaluate through the terms-only kernel on the current state ----------
    def _eval_drift_terms(self, overrides: dict):
        b = self._batch
        p = A.WlDriftParams.from_buffer_copy(b.p)
        for k, v in overrides.items():
            setattr(p, k, v)
        d = self.scene["robot"].data
        n, stride = b.n, b.stride

        def soa(t):
            out = torch.zeros(t.shape[1], stride, device=self.device)
            out[:, :n] = t.T
            return out
        pos, quat = soa(d.root_pos_w - self.scene.env_origins), soa(d.root_quat_w)
        vb, wb, ww = soa(d.root_lin_vel_b), soa(d.root_ang_vel_b), soa(d.root_ang_vel_w)
        steer, act = soa(d.joint_pos[:, 0:2]), soa(self.action_manager.action)
        terms = torch.zeros(A.WL_MAX_REW_TERMS, stride, device=self.device)
        rew = torch.zeros(n, device=self.device)
        term = torch.zeros(n, dtype=torch.bool, device=self.device)
        obs = torch.zeros(n, DriftBatch.OBS_DIM, device=self.device)
        A.check(b.lib.wl_drift_mdp(C.byref(p), n, stride, pos.data_ptr(), quat.data_ptr(), vb.data_ptr(), wb.data_ptr(),
                                   ww.data_ptr(), steer.data_ptr(), act.data_ptr(), b.truncated.data_ptr(),
                                   terms.data_ptr(), rew.data_ptr(), term.data_ptr(), obs.data_ptr(), b._stream()),
                "wl_drift_mdp")
        return terms[:, :n], term, rew, obs
