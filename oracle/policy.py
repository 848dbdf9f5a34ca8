"""TEST INFRASTRUCTURE (see oracle/__init__.py) -- numpy restatement of the policy-in-the-loop rollout.

What it restates: RSL-RL's ActorCritic.act and the runner's collection loop as the reference drives them
(wheeledlab_rl/utils/modified_rsl_rl_runner.py:70-80; agent cfg wheeledlab_tasks/drifting/config/agents/mushr/
rsl_rl_ppo_cfg.py:6,12-17: [64, 64] ELU MLPs, init_noise_std 1.0).  rsl-rl-lib (>= 2.3.0, wheeledlab_rl/setup.py:19) is
not vendored in /root/reference, so this follows its published definitions: MLP of nn.Linear + activation,
a ~ Normal(mu, std), log_prob summed over action dims.  PARITY UNPINNED against rsl_rl itself; pinned against
torch.nn.functional in tests/test_oracle_policy_cpu.py.
"""
import numpy as np

from . import drift_step as OS
from . import philox as PH

F = np.float32
S_POLICY = 7   # wl_rng.h WL_RS_POLICY
LOG_2PI = F(1.8378770664093453)


def act_fn(x, activation):
    x = x.astype(F)
    if activation == "relu":
        return np.maximum(x, F(0))
    return np.where(x > 0, x, np.expm1(np.minimum(x, F(0)))).astype(F)   # ELU, alpha = 1


def mlp(net, x):
    """net: dict w1,b1,w2,b2,w3,b3 (torch nn.Linear layout [out][in]) + activation; x [n, in] -> [n, out] float32"""
    h = act_fn(x.astype(F) @ net["w1"].T.astype(F) + net["b1"].astype(F), net["activation"])
    h = act_fn(h @ net["w2"].T.astype(F) + net["b2"].astype(F), net["activation"])
    return (h @ net["w3"].T.astype(F) + net["b3"].astype(F)).astype(F)


def policy_normals(env_ids, step, seed):
    """the two standard normals of the action sample: Box-Muller on the first two words of Philox stream 7"""
    u = PH.uniform4(env_ids, step, S_POLICY, seed)
    r = np.sqrt(F(-2.0) * np.log(F(1.0) - u[0])).astype(F)
    th = F(2.0 * np.pi) * u[1]
    return (r * np.cos(th)).astype(F), (r * np.sin(th)).astype(F)


def act(actor, std, obs, env_ids, step, seed):
    """-> actions [n,2], mu [n,2], log_prob [n]"""
    mu = mlp(actor, obs)
    z0, z1 = policy_normals(env_ids, step, seed)
    z = np.stack([z0, z1], -1)
    a = (mu + std.astype(F)[None] * z).astype(F)
    logp = (F(-0.5) * (z * z).sum(-1) - np.log(std.astype(F)).sum() - LOG_2PI).astype(F)
    return a, mu, logp


def rollout(p, state, episode_len, ref_table, actor, std, obs0, n_steps, seed, step0, env_offset=0):
    """K x { act -> env.step }; state / episode_len updated in place.  -> dict of [K(+1), n, ...] arrays"""
    n = obs0.shape[0]
    ids = np.arange(n) + env_offset
    out = dict(obs=[obs0.astype(F)], actions=[], mu=[], log_prob=[], reward=[], terminated=[], truncated=[])
    obs = obs0.astype(F)
    for k in range(n_steps):
        a, mu, logp = act(actor, std, obs, ids, step0 + k, seed)
        obs, rew, term, trunc, _ = OS.step(p, state, episode_len, ref_table, a, seed, step0 + k, env_offset=env_offset)
        out["actions"].append(a)
        out["mu"].append(mu)
        out["log_prob"].append(logp)
        out["reward"].append(rew)
        out["terminated"].append(term)
        out["truncated"].append(trunc)
        out["obs"].append(obs.astype(F))
    return {k: np.stack(v) if len(v) else np.zeros((0,)) for k, v in out.items()}


def compute_returns(rewards, values, dones, gamma=0.99, lam=0.95):
    """rsl_rl RolloutStorage.compute_returns (GAE); values has K + 1 rows.  -> returns [K, n], raw advantages [K, n]"""
    K = rewards.shape[0]
    adv = np.zeros_like(rewards, dtype=F)
    last = np.zeros(rewards.shape[1], dtype=F)
    nd = F(1.0) - dones.astype(F)
    for t in reversed(range(K)):
        delta = rewards[t] + nd[t] * F(gamma) * values[t + 1] - values[t]
        last = delta + nd[t] * F(gamma * lam) * last
        adv[t] = last
    return (adv + values[:-1]).astype(F), adv
