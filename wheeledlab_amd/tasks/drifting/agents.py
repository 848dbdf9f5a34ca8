"""PPO hyper-parameters of the drift agents as plain data (reference:
wheeledlab_tasks/drifting/config/agents/{mushr,f1tenth}/rsl_rl_ppo_cfg.py).  Consumed by wheeledlab_amd.rl.ppo."""
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
    seed: int = 42
    num_steps_per_env: int = 128
    max_iterations: int = 150
    save_interval: int = 50
    experiment_name: str = "ppo_mushr"
    empirical_normalization: bool = False
    policy: PolicyCfg = PolicyCfg()
    algorithm: AlgorithmCfg = AlgorithmCfg()


@configclass
class F1TenthPPORunnerCfg(MushrPPORunnerCfg):
    max_iterations: int = 1500
    experiment_name: str = "ppo_f1tenth"
