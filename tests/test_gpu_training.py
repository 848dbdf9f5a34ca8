"""GPU: the learner end to end on the drift task -- fused collection through the env surface (curriculum cuts, metric
ring, counters) and a short PPO run that must improve the policy."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make(n, seed=42, **cfg_over):
    import wheeledlab_amd.tasks  # noqa: F401
    from wheeledlab_amd import registry
    from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper
    cfg = registry.parse_env_cfg("Isaac-MushrDriftRL-v0", device=DEV, num_envs=n)
    cfg.seed = seed
    for k, v in cfg_over.items():
        setattr(cfg, k, v)
    env = registry.make("Isaac-MushrDriftRL-v0", cfg=cfg)
    env.action_space.low, env.action_space.high = -1.0, 1.0
    return RslRlVecEnvWrapper(ClipAction(env))


def test_env_level_fused_rollout_equals_stepping_with_the_same_actions():
    """env.rollout_policy (fused, cut at curriculum boundaries) vs env.step() fed the actions it stored: same counters,
    same curriculum weight changes at the same step, same observations to per-step tolerance, same episode log."""
    from wheeledlab_amd.policy import RolloutStorage
    from wheeledlab_amd.rl.ppo import ActorCritic
    n, K = 1024, 300                                            # crosses the 250-step episode boundary once
    a, b = _make(n), _make(n)
    torch.manual_seed(0)
    ac = ActorCritic(14, 14, 2).to(DEV)
    st = RolloutStorage(K, n, device=DEV)
    w0 = a.unwrapped.reward_manager.get_term_cfg("side_slip").weight
    a.unwrapped.rollout_policy(ac.fused(), st)
    assert a.unwrapped.common_step_counter == K and a.unwrapped._batch.step_count == K
    same = torch.ones(n, dtype=torch.bool, device=DEV)
    for k in range(K):
        obs, rew, dones, infos = b.step(st.actions[k])
        same &= dones == st.dones[k]
        if k < 3 or k == 251:
            d = (obs - st.observations[k + 1]).abs()
            d[:, 3:6] = torch.minimum(d[:, 3:6], (2 * np.pi - d[:, 3:6]).abs())
            if k < 3:
                assert d[same].max() < 2e-5
    assert same.float().mean() > 0.9
    wa = a.unwrapped.reward_manager.get_term_cfg("side_slip").weight
    wb = b.unwrapped.reward_manager.get_term_cfg("side_slip").weight
    assert wa == wb and b.unwrapped.common_step_counter == K
    assert torch.equal(st.dones != 0, st.terminated | st.time_outs)
    assert st.time_outs[249].float().mean() > 0.3 and not st.time_outs[:249].any()      # first time-outs at step 250
    log = a.unwrapped.episode_log_summary(K)
    assert log["Metrics/resets"] == float(st.dones.sum()) and "Episode_Reward/side_slip" in log
    assert w0 == 10.0


def test_short_ppo_run_improves_the_drift_policy():
    """60 iterations x 128 steps x 2048 envs (~16 M env-steps): the mean per-step reward must rise clearly and the
    fused collector must be the path in use (the adaptive-KL rule first drops the learning rate for ~10 iterations)"""
    from wheeledlab_amd.rl.ppo import OnPolicyRunner
    import wheeledlab_amd.tasks  # noqa: F401
    from wheeledlab_amd import registry
    torch.manual_seed(0)
    env = _make(2048)
    runner = OnPolicyRunner(env, registry.load_cfg_from_registry("Isaac-MushrDriftRL-v0", "rsl_rl_cfg_entry_point"), device=DEV)
    assert runner.fused
    hist = runner.learn(60, init_at_random_ep_len=True, verbose=False)
    first = np.mean([h["mean_step_reward"] for h in hist[:3]])
    last = np.mean([h["mean_step_reward"] for h in hist[-3:]])
    assert last > first + 0.03 * abs(first), (first, last)
    assert all(np.isfinite(h["value_function"]) and np.isfinite(h["surrogate"]) for h in hist)


def test_gpu_ppo_update_equals_the_cpu_update():
    """the device-resident learning-rate rule + capturable Adam (GPU) against the plain host-side rule + Adam (CPU) on
    the same storage and permutations: same parameters to fp32 tolerance, same learning-rate trajectory"""
    import copy
    from wheeledlab_amd.policy import RolloutStorage
    from wheeledlab_amd.rl.ppo import ActorCritic, PPO
    torch.manual_seed(3)
    n, K = 256, 8
    ac_c = ActorCritic(14, 14, 2)
    ac_g = copy.deepcopy(ac_c).to(DEV)
    st_c = RolloutStorage(K, n, device="cpu")
    st_c.observations.normal_()
    with torch.no_grad():
        ac_c.update_distribution(st_c.observations[:K].reshape(K * n, 14))
        a = ac_c.distribution.sample()
        st_c.actions.copy_(a.reshape(K, n, 2))
        st_c.mu.copy_(ac_c.action_mean.reshape(K, n, 2))
        st_c.actions_log_prob.copy_(ac_c.get_actions_log_prob(a).reshape(K, n))
        st_c.values.copy_(ac_c.evaluate(st_c.observations.reshape((K + 1) * n, 14)).reshape(K + 1, n))
    st_c.rewards.normal_()
    st_c.dones.copy_((torch.rand(K, n) < 0.05).long())
    st_g = RolloutStorage(K, n, device=DEV)
    for name in ("observations", "actions", "mu", "actions_log_prob", "values", "rewards", "dones"):
        getattr(st_g, name).copy_(getattr(st_c, name))
    pc, pg = PPO(ac_c), PPO(ac_g)

    class _Perm:      # the same permutations on both devices
        def __init__(self, dev):
            self.g, self.dev = torch.Generator().manual_seed(7), dev

    orig = torch.randperm
    perms = [orig(K * n, generator=torch.Generator().manual_seed(100 + i)) for i in range(10)]
    for dev_name, ppo, st in (("cpu", pc, st_c), (DEV, pg, st_g)):
        it = iter(perms)
        torch.randperm = lambda m, device=None, generator=None, _it=it: next(_it).to(device)
        try:
            for _ in range(2):
                ppo.update(st)
        finally:
            torch.randperm = orig
    for a_, b_ in zip(ac_c.parameters(), ac_g.parameters()):
        assert torch.allclose(a_, b_.cpu(), rtol=2e-3, atol=2e-4), float((a_ - b_.cpu()).abs().max())
    assert abs(pc.learning_rate - pg.learning_rate) < 1e-7


def test_checkpoint_resume_and_stepwise_collection(tmp_path):
    """save -> load into a fresh runner -> the next iteration is the same as without the round trip (fused learner state
    travels through rsl_rl's optimizer format); the step-wise collector (torch actor, one launch per env.step) feeds
    the same learner"""
    import wheeledlab_amd.tasks  # noqa: F401
    from wheeledlab_amd import registry
    from wheeledlab_amd.rl.ppo import OnPolicyRunner
    cfg = registry.load_cfg_from_registry("Isaac-MushrDriftRL-v0", "rsl_rl_cfg_entry_point")
    torch.manual_seed(1)
    a = OnPolicyRunner(_make(512, seed=5), cfg, log_dir=str(tmp_path), device=DEV)
    a.learn(3, verbose=False)
    path = str(tmp_path / "models" / "model_2.pt")
    ck = torch.load(path, weights_only=False)
    assert ck["iter"] == 3 and len(ck["optimizer_state_dict"]["state"]) == 13      # std + 6 actor + 6 critic tensors
    b = OnPolicyRunner(_make(512, seed=5), cfg, device=DEV)
    b.load(path)
    assert b.current_learning_iteration == 3 and abs(b.alg.learning_rate - a.alg.learning_rate) < 1e-9
    for p, q in zip(a.actor_critic.parameters(), b.actor_critic.parameters()):
        assert torch.equal(p, q)
    st_a = a.alg.optimizer_state_dict()["state"]
    b.alg.update(_filled_storage(b))                                                # creates the fused step from the loaded state
    st_b = b.alg.optimizer_state_dict()["state"]
    assert float(st_b[0]["step"]) == float(st_a[0]["step"]) + 20                    # 5 epochs x 4 minibatches more
    c = OnPolicyRunner(_make(256, seed=6), cfg, device=DEV, fused=False)
    assert not c.fused and c.alg.fused_update
    hist = c.learn(2, verbose=False)
    assert len(hist) == 2 and np.isfinite(hist[-1]["value_function"])


def _filled_storage(runner):
    runner.env.unwrapped.rollout_policy(runner.actor_critic.fused(), runner.storage)
    return runner.storage


@pytest.mark.parametrize("task,n,K", [("Isaac-MushrElevationRL-v0", 192, 12), ("Isaac-MushrVisualRL-v0", 96, 6),
                                      ("Isaac-MushrDriftRL-v0", 320, 260)])
def test_in_place_collection_equals_stepping_through_the_wrapper(task, n, K):
    """env.collect_step (policy kernel + fused step writing straight into the storage rows) against the same loop driven
    through RslRlVecEnvWrapper.step with copies: identical storage (same kernels, same (seed, env, step) keys), identical
    counters and curriculum state"""
    import wheeledlab_amd.tasks  # noqa: F401
    from wheeledlab_amd import registry
    from wheeledlab_amd.policy import RolloutStorage
    from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper
    from wheeledlab_amd.rl.ppo import ActorCritic

    def make():
        cfg = registry.parse_env_cfg(task, device=DEV, num_envs=n)
        cfg.seed = 7
        env = registry.make(task, cfg=cfg)
        env.action_space.low, env.action_space.high = -1.0, 1.0
        return RslRlVecEnvWrapper(ClipAction(env))

    ea, eb = make(), make()
    D = ea.num_obs
    torch.manual_seed(1)
    ac = ActorCritic(D, D, 2, activation="relu").to(DEV)
    view = ac.fused()
    sa, sb = RolloutStorage(K, n, D, 2, DEV), RolloutStorage(K, n, D, 2, DEV)
    obs_a, _ = ea.get_observations()
    obs_b, _ = eb.get_observations()
    assert torch.equal(obs_a, obs_b)
    base = ea.unwrapped
    sa.observations[0].copy_(obs_a)
    torch.manual_seed(99)     # the visual task draws its per-step augmentation from the host generator
    for k in range(K):
        base.collect_step(view, sa, k)
    base.finish_collection(sa)
    bb = eb.unwrapped._batch
    obs = obs_b
    torch.manual_seed(99)
    for k in range(K):
        sb.observations[k].copy_(obs)
        view.act(sb.observations[k], sb.actions[k], sb.mu[k], sb.actions_log_prob[k], sb.values[k], bb.seed, bb.step_count,
                 bb.env_offset)
        obs, rew, dones, infos = eb.step(sb.actions[k])
        sb.rewards[k].copy_(rew)
        sb.dones[k].copy_(dones)
        sb.time_outs[k].copy_(infos["time_outs"])
    sb.observations[K].copy_(obs)
    torch.cuda.synchronize()
    for name in ("observations", "actions", "mu", "actions_log_prob", "rewards", "dones", "time_outs"):
        assert torch.equal(getattr(sa, name), getattr(sb, name)), name
    assert torch.equal(sa.values[:K], sb.values[:K])
    # the kernel's `terminated` is the raw flag (an env may run out of bounds on its time-out step)
    assert torch.equal(sa.terminated | sa.time_outs, sb.dones != 0) and bool((sa.terminated >= ((sb.dones != 0) & ~sb.time_outs)).all())
    assert ea.unwrapped.common_step_counter == eb.unwrapped.common_step_counter == K
    assert torch.equal(ea.get_observations()[0], eb.get_observations()[0])
    if task == "Isaac-MushrDriftRL-v0":   # the curriculum fired at the 250-step boundary in both
        wa = ea.unwrapped.reward_manager.get_term_cfg("side_slip").weight
        assert wa == eb.unwrapped.reward_manager.get_term_cfg("side_slip").weight


def test_chunked_weight_gradients_equal_the_blas_ones():
    """_TallLinear (dW as one batched GEMM over 64 row chunks + a sum) against plain nn.Linear autograd on the same
    minibatch: same loss, gradients equal to fp32 summation order (1e-5 of each tensor's scale)"""
    from wheeledlab_amd.rl.ppo import ActorCritic
    B, D = 32768, 689
    torch.manual_seed(0)
    ac = ActorCritic(D, D, 2, activation="relu").to(DEV)
    obs = torch.randn(B, D, device=DEV)
    act = torch.randn(B, 2, device=DEV)
    grads = []
    for tall in (True, False):
        ac.tall_linear = tall
        ac.zero_grad(set_to_none=True)
        ac.update_distribution(obs)
        loss = -ac.get_actions_log_prob(act).mean() + ac.evaluate(obs).square().mean()
        loss.backward()
        grads.append((float(loss), [p.grad.clone() for p in ac.parameters()]))
    assert abs(grads[0][0] - grads[1][0]) < 1e-6 * abs(grads[1][0])
    for a, b in zip(grads[0][1], grads[1][1]):
        torch.testing.assert_close(a, b, rtol=0, atol=1e-5 * float(b.abs().max()) + 1e-9)


@pytest.mark.parametrize("n,activation", [(4096, "elu"), (1000, "relu")])
def test_one_launch_elevation_collector_equals_policy_step_plus_env_step(n, activation):
    """wl_elev_collect_step (policy step, env.step() and the height scan in ONE launch) against wl_actor_critic_act followed by
    wl_elev_step on a twin batch: every storage row and the env state bit for bit (the collector's layer 1 repeats the
    four-way feature split of the policy kernel at these row counts), 10 steps incl. resets; n = 1000 leaves the last
    16-env block partly empty"""
    from wheeledlab_amd.core import ElevBatch
    from wheeledlab_amd.policy import RolloutStorage
    from wheeledlab_amd.rl.ppo import ActorCritic
    K, D = 10, 689
    torch.manual_seed(3)
    ac = ActorCritic(D, D, 2, activation=activation).to(DEV)
    view = ac.fused()
    view.planes = False
    ea, eb = ElevBatch(n, device=DEV, seed=11), ElevBatch(n, device=DEV, seed=11)
    for e in (ea, eb):
        e.reset()
        e.episode_len[:n] = torch.randint(0, 245, (n,), device=DEV, dtype=torch.int32, generator=torch.Generator(device=DEV).manual_seed(1))
    sa, sb = RolloutStorage(K, n, D, 2, DEV), RolloutStorage(K, n, D, 2, DEV)
    sa.observations[0].copy_(ea.observe())
    sb.observations[0].copy_(eb.observe())
    for k in range(K):
        ea.collect_step(view, sa, k)
        view.act(sb.observations[k], sb.actions[k], sb.mu[k], sb.actions_log_prob[k], sb.values[k], eb.seed, eb.step_count, eb.env_offset)
        eb.rollout(sb.actions[k:k + 1], sb.observations[k + 1:k + 2], sb.rewards[k:k + 1], sb.terminated[k:k + 1], sb.time_outs[k:k + 1],
                   dones_out=sb.dones[k:k + 1])
    torch.cuda.synchronize()
    for name in ("actions", "mu", "actions_log_prob", "observations", "rewards", "terminated", "time_outs", "dones"):
        assert torch.equal(getattr(sa, name), getattr(sb, name)), name
    assert torch.equal(sa.values[:K], sb.values[:K])
    assert torch.equal(ea.state, eb.state) and torch.equal(ea.episode_len, eb.episode_len) and ea.step_count == eb.step_count == K
    assert int(sa.dones.sum()) > 0
    # the play policy: a = mu
    ea.collect_step(view, sa, 0, deterministic=True)
    torch.cuda.synchronize()
    assert torch.equal(sa.actions[0], sa.mu[0])


@pytest.mark.parametrize("n,activation,slots", [(4096, "relu", 1), (1000, "elu", 1), (256, "relu", 3), (256, "relu", 8)])
def test_persistent_elevation_collector(n, activation, slots):
    """wl_elev_collect_rollout (the runner's collection loop as ONE launch: actor layer 1 from the blocks' registers, observation
    rows in LDS; the critic's values come from a batched pass afterwards).  (1) A K-step launch equals K one-step launches of
    itself bit for bit (storage rows, env state).  (2) Each step against the per-step path -- wl_actor_critic_act + wl_elev_step
    on a twin batch re-synchronised to the same state and observation row before the step: policy outputs to fp32 rounding
    (layer 1 is summed in eight partial sums instead of four), the same random draws, and an env.step that agrees wherever the
    actions do.  n = 1000 leaves the last 16-env block partly empty."""
    from wheeledlab_amd.core import ElevBatch
    from wheeledlab_amd.policy import RolloutStorage
    from wheeledlab_amd.rl.ppo import ActorCritic
    K, D = 8, 689
    torch.manual_seed(5)
    ac = ActorCritic(D, D, 2, activation=activation).to(DEV)
    view = ac.fused()
    view.planes = False
    ea, eb, ec = (ElevBatch(n, device=DEV, seed=13, metrics_slots=s) for s in (slots, 1, 1))   # slots > 1: the episode-metric ring
    for e in (ea, eb, ec):
        e.reset()
        e.episode_len[:n] = torch.randint(0, 198, (n,), device=DEV, dtype=torch.int32, generator=torch.Generator(device=DEV).manual_seed(2))
    sa, sb, sc = (RolloutStorage(K, n, D, 2, DEV) for _ in range(3))
    for e, s in ((ea, sa), (eb, sb), (ec, sc)):
        s.observations[0].copy_(e.observe())
    ea.collect_rollout(view, sa)                      # one launch
    for k in range(K):                                # K launches of one step
        eb.collect_rollout(view, sb, start=k, count=1)
    torch.cuda.synchronize()
    names = ("actions", "mu", "actions_log_prob", "observations", "rewards", "terminated", "time_outs", "dones")
    for name in names:
        assert torch.equal(getattr(sa, name), getattr(sb, name)), name
    assert float(sa.values.abs().sum()) == 0.0        # not the collector's job
    assert torch.equal(ea.state, eb.state) and torch.equal(ea.episode_len, eb.episode_len) and ea.step_count == eb.step_count == K
    assert int(sa.dones.sum()) > 0 and bool(torch.isfinite(sa.observations).all())
    if slots == K:
        # a rollout whose length is a multiple of the ring (num_steps_per_env == metrics_slots): the C ABI refuses it as
        # ONE launch (slot aliasing) and the host layer runs 1 + (K - 1) steps; the ring is left holding steps 1 .. K - 1
        assert float(ea.metrics_raw[2:].abs().sum()) == 0.0 and float(ea.metrics_raw[0].abs().sum()) == 0.0
        return
    torch.testing.assert_close(ea.metrics_raw.sum((0, 1)), eb.metrics_raw.sum((0, 1)), rtol=1e-5, atol=1e-3)
    if slots > 1:                                     # the one launch books all K steps into the first step's slot
        assert float(ea.metrics_raw[1:].abs().sum()) == 0.0 and float(ea.metrics_raw[0].abs().sum()) > 0.0
    # (2) step by step against the two-launch path
    ea2 = ElevBatch(n, device=DEV, seed=13)
    ea2.reset()
    ea2.state.copy_(ec.state)
    ea2.episode_len.copy_(ec.episode_len)
    sa2 = RolloutStorage(1, n, D, 2, DEV)
    for k in range(K):
        ea2.state.copy_(ec.state)                     # same start, same observation row
        ea2.episode_len.copy_(ec.episode_len)
        ea2.step_count = ec.step_count
        sa2.observations[0].copy_(sc.observations[k])
        ea2.collect_rollout(view, sa2, start=0, count=1)
        view.act(sc.observations[k], sc.actions[k], sc.mu[k], sc.actions_log_prob[k], sc.values[k], ec.seed, ec.step_count, ec.env_offset)
        ec.rollout(sc.actions[k:k + 1], sc.observations[k + 1:k + 2], sc.rewards[k:k + 1], sc.terminated[k:k + 1], sc.time_outs[k:k + 1],
                   dones_out=sc.dones[k:k + 1])
        torch.cuda.synchronize()
        torch.testing.assert_close(sa2.mu[0], sc.mu[k], rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(sa2.actions[0], sc.actions[k], rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(sa2.actions_log_prob[0], sc.actions_log_prob[k], rtol=1e-5, atol=1e-5)
        assert torch.equal(sa2.time_outs[0], sc.time_outs[k])
        same = sa2.terminated[0] == sc.terminated[k]
        assert int((~same).sum()) <= 1                # an action 1e-6 apart may tip a termination threshold
        torch.testing.assert_close(sa2.rewards[0][same], sc.rewards[k][same], rtol=1e-3, atol=1e-2)
        d = (sa2.observations[1] - sc.observations[k + 1]).abs()[same]
        assert float(d[:, :13].max()) < 1e-3 and float((d[:, 13:] > 1e-3).float().mean()) < 1e-3
        torch.testing.assert_close(ea2.state[:13, :n][:, same], ec.state[:13, :n][:, same], rtol=1e-3, atol=1e-3)
        torch.testing.assert_close(ea2.state[17:21, :n][:, same], ec.state[17:21, :n][:, same], rtol=1e-3, atol=1e-3)
        # wheel spin = contact speed / r (r = 0.05 m): 1e-3 m/s is 2e-2 rad/s (tests/parity_predicates.py)
        torch.testing.assert_close(ea2.state[13:17, :n][:, same], ec.state[13:17, :n][:, same], rtol=1e-3, atol=2e-2)
    # the play policy: a = mu
    ea.collect_rollout(view, sa, start=0, count=1, deterministic=True)
    torch.cuda.synchronize()
    assert torch.equal(sa.actions[0], sa.mu[0])


def test_rollout_bookkeeping_kernel_equals_the_torch_bookkeeping():
    """wl_rollout_bookkeeping against the torch form of the runner's per-rollout bookkeeping (_finished_episodes, the raw-reward
    mean, the action guard, bootstrap_time_outs): finished episodes' returns / lengths in time order, updated carries, bootstrapped
    rewards; two consecutive rollouts so that the carries matter; a NaN action is reported"""
    from types import SimpleNamespace
    from wheeledlab_amd.policy import RolloutStorage
    from wheeledlab_amd.rl.ppo import OnPolicyRunner, _finished_episodes
    K, n = 24, 1000
    g = torch.Generator(device=DEV).manual_seed(0)
    runner = SimpleNamespace(device=torch.device(DEV), alg=SimpleNamespace(gamma=0.99))
    ca, cb = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)      # kernel carries
    ra, rb = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)      # torch carries
    for it in range(2):
        st = RolloutStorage(K, n, 14, 2, DEV)
        st.rewards.copy_(torch.randn(K, n, device=DEV, generator=g))
        st.values.copy_(torch.randn(K + 1, n, device=DEV, generator=g))
        st.actions.copy_(torch.randn(K, n, 2, device=DEV, generator=g))
        done = torch.rand(K, n, device=DEV, generator=g) < 0.04
        st.time_outs.copy_(done & (torch.rand(K, n, device=DEV, generator=g) < 0.5))
        st.dones.copy_(done.long())
        raw = st.rewards.clone()
        ret, length, ra, rb = _finished_episodes(raw, done, ra, rb)
        want_rew = raw + 0.99 * st.values[:-1] * st.time_outs
        rets, lens, mean, finite = OnPolicyRunner._bookkeeping(runner, st, ca, cb)
        torch.cuda.synchronize()
        assert finite and abs(mean - float(raw.mean())) < 1e-6
        torch.testing.assert_close(torch.tensor(rets), ret[-100:].cpu(), rtol=1e-5, atol=1e-5)
        assert lens == length[-100:].tolist()
        torch.testing.assert_close(ca, ra, rtol=1e-5, atol=1e-5)
        assert torch.equal(cb, rb)
        torch.testing.assert_close(st.rewards, want_rew, rtol=1e-6, atol=1e-6)
    st.actions[3, 7, 1] = float("nan")
    assert not OnPolicyRunner._bookkeeping(runner, st, ca, cb)[3]
