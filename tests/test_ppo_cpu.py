"""CPU checks of the learner (wheeledlab_amd.rl.ppo): PPO on a toy vectorised env through the runner's step-wise
collection path, checkpoint round trip with rsl_rl's keys, the adaptive-KL schedule, agent cfgs per task."""
import os

import torch

from wheeledlab_amd.rl.ppo import ActorCritic, OnPolicyRunner, PPO
from wheeledlab_amd.policy import RolloutStorage


class _ToyEnv:
    """one-step episodes: obs ~ N(0, I_4); reward = -|a - target(obs)|^2 with target = (obs0 - obs1, 0.5 obs2)"""

    def __init__(self, n=512, seed=0):
        self.num_envs, self.num_obs, self.num_actions = n, 4, 2
        self.device = torch.device("cpu")
        self.max_episode_length = 1
        self.episode_length_buf = torch.zeros(n, dtype=torch.int32)
        self.g = torch.Generator().manual_seed(seed)
        self.obs = torch.randn(n, 4, generator=self.g)

    @property
    def unwrapped(self):
        return self

    def get_observations(self):
        return self.obs, {"observations": {"policy": self.obs}}

    def step(self, a):
        tgt = torch.stack([self.obs[:, 0] - self.obs[:, 1], 0.5 * self.obs[:, 2]], -1)
        rew = -((a - tgt) ** 2).sum(-1)
        self.obs = torch.randn(self.num_envs, 4, generator=self.g)
        dones = torch.ones(self.num_envs, dtype=torch.long)
        return self.obs, rew, dones, {"observations": {"policy": self.obs}, "time_outs": torch.zeros(self.num_envs, dtype=torch.bool)}


CFG = dict(num_steps_per_env=8, save_interval=1000,
           policy=dict(init_noise_std=1.0, actor_hidden_dims=[64, 64], critic_hidden_dims=[64, 64], activation="elu"),
           algorithm=dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.005,
                          num_learning_epochs=5, num_mini_batches=4, learning_rate=1e-3, schedule="adaptive", gamma=0.99,
                          lam=0.95, desired_kl=0.01, max_grad_norm=1.0))


def test_ppo_improves_a_toy_problem_and_checkpoints_round_trip(tmp_path):
    torch.manual_seed(0)
    env = _ToyEnv()
    runner = OnPolicyRunner(env, CFG, log_dir=str(tmp_path), device="cpu")
    assert not runner.fused                                          # no fused collector off the drift task / off GPU
    hist = runner.learn(40, verbose=False)
    assert hist[-1]["mean_step_reward"] > hist[0]["mean_step_reward"] + 1.0, (hist[0], hist[-1])
    assert hist[-1]["mean_noise_std"] < 1.0 and all(h["kl"] >= 0 for h in hist)
    path = os.path.join(str(tmp_path), "models", "model_39.pt")
    assert os.path.exists(path)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "iter", "infos"} and ck["iter"] == 40
    assert {"std", "actor.0.weight", "actor.4.bias", "critic.4.weight"} <= set(ck["model_state_dict"])   # rsl_rl's names
    other = OnPolicyRunner(_ToyEnv(seed=1), CFG, device="cpu")
    ptr = other.actor_critic.actor[0].weight.data_ptr()
    other.load(path)
    assert other.actor_critic.actor[0].weight.data_ptr() == ptr                     # loaded in place
    x = torch.randn(16, 4)
    assert torch.equal(other.get_inference_policy()(x), runner.get_inference_policy()(x))
    assert other.current_learning_iteration == 40


def test_adaptive_kl_schedule_moves_the_learning_rate():
    torch.manual_seed(1)
    ac = ActorCritic(4, 4, 2)
    n, K = 64, 4
    st = RolloutStorage(K, n, obs_dim=4, device="cpu")
    st.observations.normal_()
    with torch.no_grad():
        ac.update_distribution(st.observations[:K].reshape(K * n, 4))
        a = ac.distribution.sample()
        st.actions.copy_(a.reshape(K, n, 2))
        st.mu.copy_(ac.action_mean.reshape(K, n, 2))
        st.actions_log_prob.copy_(ac.get_actions_log_prob(a).reshape(K, n))
        st.values.copy_(ac.evaluate(st.observations.reshape((K + 1) * n, 4)).reshape(K + 1, n))
    st.rewards.normal_()
    calm = PPO(ac, learning_rate=1e-3, desired_kl=1e9, num_learning_epochs=1, num_mini_batches=1)
    calm.update(st)
    assert calm.learning_rate > 1e-3                                   # KL far below the target: lr * 1.5
    hot = PPO(ac, learning_rate=1e-3, desired_kl=1e-12, num_learning_epochs=2, num_mini_batches=1)
    hot.update(st)
    assert hot.learning_rate < 1e-3                                    # KL above 2 x target after the first step: lr / 1.5
    assert all(g["lr"] == hot.learning_rate for g in hot.optimizer.param_groups)


def test_agent_cfgs_follow_the_reference_per_task():
    import wheeledlab_amd.tasks  # noqa: F401
    from wheeledlab_amd import registry
    want = {"Isaac-MushrDriftRL-v0": ("ppo_mushr", 150, "elu"), "Isaac-F1TenthDriftRL-v0": ("ppo_f1tenth", 1500, "elu"),
            "Isaac-MushrElevationRL-v0": ("ppo_mushr_elevation", 4000, "relu"),
            "Isaac-MushrVisualRL-v0": ("ppo_mushr_visual", 4000, "relu")}
    for task, (name, iters, act) in want.items():
        d = registry.load_cfg_from_registry(task, "rsl_rl_cfg_entry_point").to_dict()
        assert (d["experiment_name"], d["max_iterations"], d["policy"]["activation"]) == (name, iters, act)
        assert d["num_steps_per_env"] == 128 and d["save_interval"] == 50 and d["algorithm"]["num_mini_batches"] == 4


def test_vectorised_episode_bookkeeping_equals_the_runners_per_step_loop():
    from wheeledlab_amd.rl.ppo import _finished_episodes
    g = torch.Generator().manual_seed(5)
    K, n = 37, 11
    cr, cl = torch.rand(n, generator=g), torch.randint(0, 9, (n,), generator=g).float()
    want_r, want_l, r0, l0 = [], [], cr.clone(), cl.clone()
    got_r, got_l = [], []
    for _ in range(3):                                   # carries across three rollouts
        rew = torch.randn(K, n, generator=g)
        done = torch.rand(K, n, generator=g) < 0.08
        for k in range(K):                               # modified_rsl_rl_runner.py:88-98
            r0 += rew[k]
            l0 += 1
            ids = done[k].nonzero().flatten()
            want_r += r0[ids].tolist()
            want_l += l0[ids].tolist()
            r0[ids] = 0
            l0[ids] = 0
        a, b, cr, cl = _finished_episodes(rew, done, cr, cl)
        got_r += a.tolist()
        got_l += b.tolist()
    assert torch.allclose(torch.tensor(got_r), torch.tensor(want_r), atol=1e-5) and got_l == want_l
    assert torch.allclose(cr, r0, atol=1e-5) and torch.equal(cl, l0)


def test_runner_raises_on_non_finite_actions():
    """the reference runner's only runtime guard (modified_rsl_rl_runner.py:74-75)"""
    import pytest
    runner = OnPolicyRunner(_ToyEnv(seed=0), CFG, device="cpu")
    with torch.no_grad():
        runner.actor_critic.actor[-1].bias.fill_(float("nan"))
    with pytest.raises(ValueError, match="non-finite"):
        runner.learn(1, verbose=False)
