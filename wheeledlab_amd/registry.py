"""gym-style registry with the reference's four task ids (wheeledlab_tasks/__init__.py:14-63).  `gymnasium` is not
a dependency (it is absent from the target image); when it IS importable the same ids are also registered with it,
with this package's env class as the entry point, so `gym.make("Isaac-MushrDriftRL-v0", cfg=env_cfg)` works."""
from __future__ import annotations

import importlib
from dataclasses import dataclass, field


@dataclass
class EnvSpec:
    id: str
    entry_point: object
    kwargs: dict = field(default_factory=dict)
    disable_env_checker: bool = True


_REGISTRY: dict[str, EnvSpec] = {}


def _resolve(entry):
    if isinstance(entry, str):
        mod, _, attr = entry.partition(":")
        return getattr(importlib.import_module(mod), attr)
    return entry


def register(id: str, entry_point, kwargs=None, disable_env_checker=True, **_):
    _REGISTRY[id] = EnvSpec(id, entry_point, dict(kwargs or {}), disable_env_checker)
    try:  # mirror into gymnasium when present
        import gymnasium as gym
        if id not in gym.registry:
            gym.register(id=id, entry_point=entry_point, kwargs=kwargs, disable_env_checker=disable_env_checker)
    except Exception:
        pass


def spec(id: str) -> EnvSpec:
    if id not in _REGISTRY:
        raise KeyError(f"no registered env '{id}'; known: {sorted(_REGISTRY)}")
    return _REGISTRY[id]


def registered_ids():
    return sorted(_REGISTRY)


def load_cfg_from_registry(task: str, entry_key: str = "env_cfg_entry_point"):
    """isaaclab_tasks.utils.load_cfg_from_registry work-alike: returns a config INSTANCE"""
    entry = _resolve(spec(task).kwargs[entry_key])
    return entry() if isinstance(entry, type) else entry


def parse_env_cfg(task: str, device: str = "cuda:0", num_envs: int | None = None, play: bool = False):
    """isaaclab_tasks.utils.parse_env_cfg work-alike (reference: wheeledlab_tasks/test/create_and_step_env.py:26);
    play=True picks the task's `play_env_cfg_entry_point` when it registers one (scripts/play_policy.py)"""
    key = "play_env_cfg_entry_point" if play and "play_env_cfg_entry_point" in spec(task).kwargs else "env_cfg_entry_point"
    cfg = load_cfg_from_registry(task, key)
    cfg.sim.device = device
    if num_envs is not None:
        cfg.num_envs = num_envs
        cfg.scene.num_envs = num_envs
    return cfg


def make(id: str, cfg=None, render_mode=None, **kwargs):
    """gym.make(id, cfg=env_cfg, render_mode=...) (reference: scripts/train_rl.py:70)"""
    s = spec(id)
    if cfg is None:
        cfg = load_cfg_from_registry(id)
    return _resolve(s.entry_point)(cfg=cfg, render_mode=render_mode, **kwargs)
