"""Run configurations: what the reference's `train_rl.py -r <RUN_CONFIG> key=value ...` resolves
(wheeledlab_rl/configs/common_cfg.py:12-88, rl_cfg.py:7-30, utils/hydra.py:24-40).  Hydra / OmegaConf are not in the
target image; `runs.resolve_run` applies the same dotted `key=value` overrides to plain config objects."""
from __future__ import annotations

import os
import random

from ..envs.configclass import MISSING, configclass

WHEELEDLAB_LOGS_DIR = os.environ.get("WHEELEDLAB_LOGS_DIR", os.path.join(os.getcwd(), "logs"))


@configclass
class LogConfig:
    """logging during training (common_cfg.py:12-45); video / wandb switches are accepted and ignored (no renderer)"""
    logs_dir: str = WHEELEDLAB_LOGS_DIR
    no_log: bool = False
    log_every: int = 10
    video: bool = False
    video_length: int = 500
    video_interval: int = 5000
    no_checkpoints: bool = False
    checkpoint_every: int = 1000
    no_wandb: bool = True
    wandb_project: str = "WheeledLab"
    test_mode: bool = False
    model_save_dirname: str = "models"
    run_name: str = f"run-{random.randint(0, int(1e7))}"

    @property
    def run_log_dir(self):
        return os.path.join(self.logs_dir, self.run_name)

    @property
    def model_save_path(self):
        return os.path.join(self.run_log_dir, self.model_save_dirname)


@configclass
class TrainConfig:
    seed: int = 0
    device: str = "cuda:0"
    load_run: str = None
    load_run_checkpoint: int = 0
    log: LogConfig = LogConfig()


@configclass
class EnvSetup:
    num_envs: int = 1024
    task_name: str = MISSING


@configclass
class AgentSetup:
    entry_point: str = "rsl_rl_cfg_entry_point"


@configclass
class RunConfig:
    train: TrainConfig = TrainConfig()
    env_setup: EnvSetup = EnvSetup()
    agent_setup: AgentSetup = AgentSetup()
    env: object = MISSING      # resolved from the task registry
    agent: object = MISSING


@configclass
class RLTrainConfig(TrainConfig):
    agent_n_steps: int = 200
    num_iterations: int = 2048
    rl_algo_lib: str = MISSING
    rl_algo_class: str = MISSING
    set_env_step: int = 0


@configclass
class RslRlRunConfig(RunConfig):
    train: RLTrainConfig = RLTrainConfig(rl_algo_lib="rsl", rl_algo_class="ppo", log=LogConfig(video_interval=15000))
    agent_setup: AgentSetup = AgentSetup(entry_point="rsl_rl_cfg_entry_point")
