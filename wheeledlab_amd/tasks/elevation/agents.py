"""PPO hyper-parameters of the elevation agent (reference:
wheeledlab_tasks/elevation/config/agents/mushr/rsl_rl_ppo_cfg.py): as the drift agent but ReLU MLPs and 4000 iterations."""
from ...envs.configclass import configclass
from ..drifting.agents import AlgorithmCfg, PolicyCfg


@configclass
class ReluPolicyCfg(PolicyCfg):
    activation: str = "relu"


@configclass
class MushrPPORunnerCfg:
    seed: int = 42
    num_steps_per_env: int = 128
    max_iterations: int = 4000
    save_interval: int = 50
    experiment_name: str = "ppo_mushr_elevation"
    empirical_normalization: bool = False
    policy: ReluPolicyCfg = ReluPolicyCfg()
    algorithm: AlgorithmCfg = AlgorithmCfg()
