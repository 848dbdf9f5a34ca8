"""Registered run configs (wheeledlab_rl/configs/runs/__init__.py:1-10, rss_cfgs.py, f1tenth_cfgs.py) and their resolution."""
from __future__ import annotations

import ast

from ...envs.configclass import MISSING, configclass
from .. import AgentSetup, EnvSetup, LogConfig, RLTrainConfig, RslRlRunConfig


@configclass
class RSS_DRIFT_CONFIG(RslRlRunConfig):
    env_setup = EnvSetup(num_envs=1024, task_name="Isaac-MushrDriftRL-v0")
    train = RLTrainConfig(num_iterations=5000, rl_algo_lib="rsl", rl_algo_class="ppo", log=LogConfig(video_interval=15000))
    agent_setup = AgentSetup(entry_point="rsl_rl_cfg_entry_point")


@configclass
class RSS_VISUAL_CONFIG(RslRlRunConfig):
    env_setup = EnvSetup(num_envs=512, task_name="Isaac-MushrVisualRL-v0")
    train = RLTrainConfig(num_iterations=5000, rl_algo_lib="rsl", rl_algo_class="ppo")
    agent_setup = AgentSetup(entry_point="rsl_rl_cfg_entry_point")


@configclass
class RSS_ELEV_CONFIG(RslRlRunConfig):
    env_setup = EnvSetup(num_envs=1024, task_name="Isaac-MushrElevationRL-v0")
    train = RLTrainConfig(num_iterations=5000, rl_algo_lib="rsl", rl_algo_class="ppo")
    agent_setup = AgentSetup(entry_point="rsl_rl_cfg_entry_point")


@configclass
class F1TENTH_DRIFT_CONFIG(RslRlRunConfig):
    env_setup = EnvSetup(num_envs=1024, task_name="Isaac-F1TenthDriftRL-v0")
    train = RLTrainConfig(num_iterations=5000, rl_algo_lib="rsl", rl_algo_class="ppo", log=LogConfig(video_interval=15000))
    agent_setup = AgentSetup(entry_point="rsl_rl_cfg_entry_point")


@configclass
class VISUAL_DEPTH_CONFIG(RslRlRunConfig):
    """EXTENSION (not a reference run): the visual task on a heightfield with the depth image as observation"""
    env_setup = EnvSetup(num_envs=512, task_name="Isaac-MushrVisualDepthRL-v0")
    train = RLTrainConfig(num_iterations=5000, rl_algo_lib="rsl", rl_algo_class="ppo")
    agent_setup = AgentSetup(entry_point="rsl_rl_cfg_entry_point")


_RUNS: dict[str, type] = {}


def register_run(name: str, node: type):
    """register_run_to_hydra work-alike (utils/hydra.py:68-97): the store is a dict"""
    _RUNS[name] = node


def registered_runs():
    return sorted(_RUNS)


for _n, _c in (("RSS_DRIFT_CONFIG", RSS_DRIFT_CONFIG), ("RSS_ELEV_CONFIG", RSS_ELEV_CONFIG),
               ("RSS_VISUAL_CONFIG", RSS_VISUAL_CONFIG), ("F1TENTH_DRIFT_CONFIG", F1TENTH_DRIFT_CONFIG),
               ("VISUAL_DEPTH_CONFIG", VISUAL_DEPTH_CONFIG)):
    register_run(_n, _c)


def _parse(text: str):
    low = text.strip().lower()
    if low in ("true", "false"):
        return low == "true"
    if low in ("null", "none"):
        return None
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def apply_override(root, dotted: str, value):
    """Hydra-style `a.b.c=value` on config objects / dicts / lists; unknown keys are errors (as with a structured config)"""
    *path, last = dotted.split(".")
    node = root
    for key in path:
        if isinstance(node, dict):
            node = node[key]
        elif isinstance(node, (list, tuple)):
            node = node[int(key)]
        else:
            if not hasattr(node, key):
                raise KeyError(f"override '{dotted}': '{type(node).__name__}' has no field '{key}'")
            node = getattr(node, key)
        if node is MISSING or node is None:
            raise KeyError(f"override '{dotted}': '{key}' is not set")
    if isinstance(node, dict):
        if last not in node:
            raise KeyError(f"override '{dotted}': no key '{last}'")
        node[last] = value
    elif isinstance(node, list):
        node[int(last)] = value
    else:
        if not hasattr(node, last):
            raise KeyError(f"override '{dotted}': '{type(node).__name__}' has no field '{last}'")
        cur = getattr(node, last)
        if isinstance(cur, float) and isinstance(value, int) and not isinstance(value, bool):
            value = float(value)
        setattr(node, last, value)


def resolve_run(name: str, overrides=()):
    """What `@hydra_run_config(run_config_name)` hands to `main(run_cfg)` (utils/hydra.py:101-167): the run config with
    `env` / `agent` resolved from the task registry, the `key=value` overrides applied, and the exposed overrides
    consolidated (env.scene.num_envs <- env_setup.num_envs, env.seed <- agent.seed, env.sim.device <- train.device)."""
    from ... import registry, tasks  # noqa: F401  (registers the task ids)
    if name not in _RUNS:
        raise KeyError(f"run config '{name}' is not registered; known: {registered_runs()}")
    run = _RUNS[name]()
    pairs = []
    for ov in overrides:
        if "=" not in ov:
            raise ValueError(f"override '{ov}' is not of the form key=value")
        k, v = ov.split("=", 1)
        pairs.append((k.lstrip("+"), _parse(v)))
    for k, v in pairs:                       # the task may be switched by an override before env / agent are resolved
        if k.startswith(("env_setup.", "agent_setup.")):
            apply_override(run, k, v)
    run.env = registry.load_cfg_from_registry(run.env_setup.task_name, "env_cfg_entry_point")
    run.agent = registry.load_cfg_from_registry(run.env_setup.task_name, run.agent_setup.entry_point)
    for k, v in pairs:
        if not k.startswith(("env_setup.", "agent_setup.")):
            apply_override(run, k, v)
    run.env.scene.num_envs = run.env_setup.num_envs
    if hasattr(run.env, "num_envs"):
        run.env.num_envs = run.env_setup.num_envs
    run.env.seed = run.agent.seed
    run.env.sim.device = run.train.device
    log = run.train.log
    if log.test_mode:
        log.no_log, log.no_wandb, log.video, log.no_checkpoints = True, True, False, True
    return run
