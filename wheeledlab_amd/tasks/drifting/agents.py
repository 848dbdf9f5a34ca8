"""PPO hyper-parameters of the drift task as plain data (reference:
wheeledlab_tasks/drifting/config/agents/mushr/rsl_rl_ppo_cfg.py:5-31).  The learner itself is out of scope; these
values fix the rollout length (128 steps / env) the throughput harness reproduces."""
from ...envs.configclass import configclass


@configclass
class PolicyCfg:
    init_noise_std: float = 1.0
    actor_hidden_dims: list = [64, 64]
    critic_hidden_dims: list = [64, 64]
    activation: str = "elu"


@configclass
class AlgorithmCfg:
    value_loss_coef: float = 1.0
    use_clipped_value_loss: bool = True
    clip_param: float = 0.2
    entropy_coef: float = 0.005
    num_learning_epochs: int = 5
    num_mini_batches: int = 4
    learning_rate: float = 1.0e-3
    schedule: str = "adaptive"
    gamma: float = 0.99
    lam: float = 0.95
    desired_kl: float = 0.01
    max_grad_norm: float = 1.0


@configclass
class MushrPPORunnerCfg:
    num_steps_per_env: int = 128
    max_iterations: int = 5000
    save_interval: int = 100
    experiment_name: str = "mushr_drift"
    empirical_normalization: bool = False
    policy: PolicyCfg = PolicyCfg()
    algorithm: AlgorithmCfg = AlgorithmCfg()
