"""GPU parity of the policy-in-the-loop row: the matrix-pipe MLP and the fused
{ actor -> sample -> env.step } rollout, through the C ABI, against the numpy oracle and against the library's own
per-step path."""
import numpy as np
import pytest
import torch

from oracle import params as OP
from oracle import policy as OPOL

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    from wheeledlab_amd import _abi as A
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return A.load()


def _np_net(m):
    d = dict(activation=m.activation)
    for k in ("w1", "b1", "w2", "b2", "w3", "b3"):
        d[k] = getattr(m, k).cpu().numpy()
    return d


@pytest.mark.parametrize("activation", ["elu", "relu"])
@pytest.mark.parametrize("in_dim,out_dim", [(14, 2), (14, 1), (15, 4), (3, 3)])
def test_mlp_forward_matches_oracle(lib, activation, in_dim, out_dim):
    """fp32 MFMA is an exact fmaf chain; only the summation order differs from numpy's: tolerance 2e-5 abs+rel on
    O(1) activations.  Random (asymmetric) weights catch any row/column or k-order mix-up; ragged row counts cover
    partial 16-row tiles and the grid-stride loop."""
    from wheeledlab_amd.policy import Mlp
    g = torch.Generator().manual_seed(11)
    net = Mlp(in_dim, out_dim, activation, DEV, generator=g)
    for w in (net.w1, net.w2, net.w3, net.b1, net.b2, net.b3):
        w.mul_(2.0)                                            # push units into both branches of the activation
    for rows in (1, 15, 16, 17, 1000, 70001, 600000):
        x = (torch.randn(rows, in_dim, generator=g) * 1.5).to(DEV)
        y = net(x)
        torch.cuda.synchronize()
        want = OPOL.mlp(_np_net(net), x.cpu().numpy())
        np.testing.assert_allclose(y.cpu().numpy(), want, rtol=2e-5, atol=2e-5, err_msg=f"rows={rows}")


def _setup(n, seed, corruption=True):
    from wheeledlab_amd.core import DriftBatch
    from wheeledlab_amd.policy import ActorCritic
    env = DriftBatch(n, device=DEV, seed=seed)
    env.reset()
    if not corruption:
        env.p.enable_corruption = 0
    env.observe()
    ac = ActorCritic(device=DEV, seed=3)
    ac.std.copy_(torch.tensor([0.6, 0.9]))
    return env, ac


def _cmp_obs(got, want, ok, tol, tag):
    d = np.abs(got - want)[ok]
    d[:, 3:6] = np.minimum(d[:, 3:6], np.abs(2 * np.pi - d[:, 3:6]))   # euler angles live on the circle
    assert d.max() < tol, (tag, d.max())


@pytest.mark.parametrize("n", [200, 1024])
def test_policy_rollout_matches_oracle_teacher_forced(lib, n):
    """12 launches of K = 1, each compared with the oracle's { mlp -> sample -> step } from the device's own state and
    observation, so nothing accumulates: actor outputs to MFMA-vs-numpy summation order (2e-5), the env step to the
    fused-step test's tolerances (state 2e-4, reward 2e-3, obs 1e-3)."""
    from wheeledlab_amd.policy import RolloutStorage
    env, ac = _setup(n, seed=21)
    p = OP.drift_params()
    ref, actor, std = env.ref_table.cpu().numpy(), _np_net(ac.actor), ac.std.cpu().numpy()
    store = RolloutStorage(1, n, device=DEV)
    flips = 0
    for k in range(12):
        if k == 6:
            env.episode_len[: n // 4] = 249                          # time-outs and in-kernel resets
        st, ep, obs0 = env.state.cpu().numpy()[:, :n].copy(), env.episode_len.cpu().numpy()[:n].copy(), env.obs.cpu().numpy().copy()
        env.rollout_policy(ac, store)
        torch.cuda.synchronize()
        want = OPOL.rollout(p, st, ep, ref, actor, std, obs0, 1, env.seed, k)
        np.testing.assert_allclose(store.mu[0].cpu().numpy(), want["mu"][0], rtol=2e-5, atol=2e-5, err_msg=f"mu {k}")
        np.testing.assert_allclose(store.actions[0].cpu().numpy(), want["actions"][0], rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(store.actions_log_prob[0].cpu().numpy(), want["log_prob"][0], rtol=1e-5, atol=2e-5)
        np.testing.assert_array_equal(store.time_outs[0].cpu().numpy(), want["truncated"][0])
        bad = store.terminated[0].cpu().numpy() != want["terminated"][0]
        flips += int(bad.sum())
        ok = ~bad
        np.testing.assert_allclose(store.rewards[0].cpu().numpy()[ok], want["reward"][0][ok], rtol=2e-3, atol=2e-3)
        _cmp_obs(store.observations[1].cpu().numpy(), want["obs"][1], ok, 1e-3, k)
        np.testing.assert_allclose(env.state.cpu().numpy()[:23, :n][:, ok], st[:23][:, ok], rtol=2e-4, atol=2e-4)
        np.testing.assert_array_equal(env.episode_len.cpu().numpy()[:n][ok], ep[ok])
        np.testing.assert_array_equal(store.observations[0].cpu().numpy(), obs0)
        assert torch.equal(store.dones[0] != 0, store.terminated[0] | store.time_outs[0])
        assert torch.equal(env.obs, store.observations[1]) and env.step_count == k + 1
        v = OPOL.mlp(_np_net(ac.critic), store.observations.cpu().numpy().reshape(-1, 14)).reshape(2, n)
        np.testing.assert_allclose(store.values.cpu().numpy(), v, rtol=2e-5, atol=2e-5)
    assert flips <= 2


def test_policy_rollout_multi_step_tracks_oracle(lib):
    """K = 4 in one launch vs the oracle loop: differences now pass through policy and physics each step, so the bars
    are trajectory-divergence bars (the tight per-step bars are in the teacher-forced test above).  The observation
    carries Euler angles wrapped to [0, 2 pi) (IsaacLab's euler_xyz_from_quat; roll and pitch of a car on flat ground
    sit at the seam), so an angle of -1e-7 on one side and +1e-7 on the other are 2 pi apart as policy INPUTS: envs
    where the wrap decision flips leave the comparison, like termination flips."""
    from wheeledlab_amd.policy import RolloutStorage
    n, K = 600, 4
    env, ac = _setup(n, seed=23)
    env.episode_len[: n // 4] = 248
    p = OP.drift_params()
    st, ep, obs0 = env.state.cpu().numpy()[:, :n].copy(), env.episode_len.cpu().numpy()[:n].copy(), env.obs.cpu().numpy().copy()
    store = RolloutStorage(K, n, device=DEV)
    env.rollout_policy(ac, store)
    torch.cuda.synchronize()
    want = OPOL.rollout(p, st, ep, env.ref_table.cpu().numpy(), _np_net(ac.actor), ac.std.cpu().numpy(), obs0, K, env.seed, 0)
    ok = np.ones(n, bool)
    for k in range(K):
        tol = 2e-5 if k == 0 else 1e-2
        np.testing.assert_allclose(store.mu[k].cpu().numpy()[ok], want["mu"][k][ok], rtol=tol, atol=tol, err_msg=f"mu {k}")
        np.testing.assert_allclose(store.actions[k].cpu().numpy()[ok], want["actions"][k][ok], rtol=tol, atol=tol)
        np.testing.assert_array_equal(store.time_outs[k].cpu().numpy(), want["truncated"][k])
        ok &= store.terminated[k].cpu().numpy() == want["terminated"][k]
        got_obs = store.observations[k + 1].cpu().numpy()
        _cmp_obs(got_obs, want["obs"][k + 1], ok, 1e-3 if k == 0 else 2e-2, k)
        ok &= ~(np.abs(got_obs - want["obs"][k + 1])[:, 3:6] > 3.0).any(-1)      # wrap flips
    assert ok.sum() >= 20 and store.time_outs.any() and env.step_count == K
    np.testing.assert_array_equal(env.episode_len.cpu().numpy()[:n][ok], ep[ok])


def test_policy_rollout_full_size_self_consistency(lib):
    """BASELINE size (4096 envs, the reference's 128 steps per env): the stored transitions are mutually consistent
    with the library's own per-step path -- mu is the actor on the stored observation, the log-prob is that of the
    stored action, replaying the stored actions through wl_drift_step reproduces the stored observations."""
    from wheeledlab_amd.core import DriftBatch
    from wheeledlab_amd.policy import RolloutStorage
    n, K = 4096, 128
    env, ac = _setup(n, seed=9)
    twin = DriftBatch(n, device=DEV, seed=9)
    twin.reset()
    twin.observe()
    assert torch.equal(twin.state, env.state) and torch.equal(twin.obs, env.obs)
    ep0 = torch.randint(0, 250, (n,), generator=torch.Generator().manual_seed(1), dtype=torch.int32).to(DEV)
    env.episode_len[:n] = ep0                                          # the runner's init_at_random_ep_len
    twin.episode_len[:n] = ep0
    store = RolloutStorage(K, n, device=DEV)
    env.rollout_policy(ac, store)
    torch.cuda.synchronize()
    for t in (store.observations, store.actions, store.mu, store.actions_log_prob, store.rewards, store.values):
        assert torch.isfinite(t).all()
    mu = ac.actor(store.observations[:K].contiguous())
    assert (mu - store.mu).abs().max() < 1e-5
    want_lp = torch.distributions.Normal(store.mu, ac.std).log_prob(store.actions).sum(-1)
    assert (want_lp - store.actions_log_prob).abs().max() < 2e-3      # (a - mu) / std re-derived from rounded a
    z = (store.actions - store.mu) / ac.std
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
    assert torch.equal(store.dones != 0, store.terminated | store.time_outs)
    assert store.time_outs.any() and store.terminated.any()
    same = torch.ones(n, dtype=torch.bool, device=DEV)
    for k in range(24):                                               # replay through the per-step kernel
        obs, rew, term, trunc = twin.step(store.actions[k])
        assert torch.equal(trunc, store.time_outs[k])
        same &= term == store.terminated[k]
        d = (obs - store.observations[k + 1]).abs()
        d[:, 3:6] = torch.minimum(d[:, 3:6], (2 * np.pi - d[:, 3:6]).abs())
        tol = 2e-5 if k < 3 else 5e-3
        assert d[same].max() < tol, (k, float(d[same].max()))
        assert (rew - store.rewards[k])[same].abs().max() < (1e-3 if k < 3 else 0.5)
    assert same.float().mean() > 0.99
    # episode metrics of the whole rollout: resets counted == dones stored
    assert abs(float(env.metrics[8]) - float(store.dones.sum())) < 0.5


def test_policy_rollout_shards_by_env_offset(lib):
    """the action noise is keyed by the GLOBAL env id: two half-size shards with env_offset reproduce the big batch"""
    from wheeledlab_amd.core import DriftBatch
    from wheeledlab_amd.policy import ActorCritic, RolloutStorage
    n, K = 512, 6
    ac = ActorCritic(device=DEV, seed=3)
    big = DriftBatch(n, device=DEV, seed=4)
    big.reset()
    big.observe()
    halves = []
    for r in range(2):
        h = DriftBatch(n // 2, device=DEV, seed=4, env_offset=r * (n // 2))
        h.state.copy_(big.state[:, r * (n // 2):(r + 1) * (n // 2)])
        h.episode_len.copy_(big.episode_len[r * (n // 2):(r + 1) * (n // 2)])
        h.obs.copy_(big.obs[r * (n // 2):(r + 1) * (n // 2)])
        halves.append(h)
    sb = RolloutStorage(K, n, device=DEV)
    big.rollout_policy(ac, sb)
    for r, h in enumerate(halves):
        sh = RolloutStorage(K, n // 2, device=DEV)
        h.rollout_policy(ac, sh)
        sl = slice(r * (n // 2), (r + 1) * (n // 2))
        assert torch.equal(sh.actions, sb.actions[:, sl]) and torch.equal(sh.observations, sb.observations[:, sl])
        assert torch.equal(sh.rewards, sb.rewards[:, sl]) and torch.equal(sh.dones, sb.dones[:, sl])
