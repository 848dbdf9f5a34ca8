"""GPU parity of the fused PPO step (csrc/wl_ppo.hip) against torch autograd / torch.optim.Adam on the same minibatch."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(B, activation="elu", seed=0):
    from wheeledlab_amd.rl.ppo import ActorCritic
    torch.manual_seed(seed)
    ac = ActorCritic(14, 14, 2, activation=activation).to(DEV)
    with torch.no_grad():
        ac.std.copy_(torch.tensor([0.8, 1.1]))
        for p in ac.parameters():
            if p.dim() == 2:
                p.mul_(1.5)
    g = torch.Generator(device=DEV).manual_seed(seed + 1)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    obs = r(B, 14)
    if activation == "relu":
        # ReLU's derivative jumps at 0.  Layer 2 (forward and its transpose backward) runs on the bf16 pipe with operands
        # split into two planes (16 mantissa bits): a hidden pre-activation within ~1e-5 of zero can take the other branch
        # than torch's f32 GEMM and move a weight gradient by |delta . h| -- a different (equally valid) subgradient, not a
        # rounding-sized difference.  The comparison therefore runs on observations whose pre-activations keep clear of 0.
        with torch.no_grad():
            for _ in range(20):
                risky = torch.zeros(B, dtype=torch.bool, device=DEV)
                for net in (ac.actor, ac.critic):
                    z1 = obs.double() @ net[0].weight.double().t() + net[0].bias.double()
                    z2 = torch.relu(z1) @ net[2].weight.double().t() + net[2].bias.double()
                    risky |= (z1.abs() < 1e-5).any(1) | (z2.abs() < 1e-4).any(1)
                if not bool(risky.any()):
                    break
                obs[risky] = r(int(risky.sum()), 14)
            assert not bool(risky.any())
    with torch.no_grad():
        ac.update_distribution(obs)
        actions = ac.distribution.sample()
        mu = ac.action_mean + 0.05 * r(B, 2)                       # "old" policy slightly off the current one
        logp = ac.get_actions_log_prob(actions) + 0.1 * r(B)
        values = ac.evaluate(obs).squeeze(-1) + 0.3 * r(B)
    flat = dict(obs=obs, actions=actions.contiguous(), mu=mu.contiguous(), logp=logp.contiguous(), adv=r(B), returns=r(B),
                values=values.contiguous())
    sigma_old = torch.tensor([0.85, 1.05], device=DEV)
    return ac, flat, sigma_old


def _close_up_to_adam_noise(p, q, lr, steps, what):
    """parameters of two learners after `steps` Adam steps: all but a few elements within (rtol 1e-3, atol 2e-4); the
    elements whose gradient is within rounding of zero may differ by a fraction of lr per step (see the trajectory test)"""
    d = (p - q).abs()
    bad = d > 2e-4 + 1e-3 * q.abs()
    assert float(bad.float().mean()) < 2e-3, (what, float(bad.float().mean()))
    assert float(d.max()) < 0.25 * lr * steps + 2e-4, (what, float(d.max()))


def _torch_loss(ac, ppo, b, sigma_old):
    ac.update_distribution(b["obs"])
    logp = ac.get_actions_log_prob(b["actions"])
    value = ac.evaluate(b["obs"]).squeeze(-1)
    ratio = torch.exp(logp - b["logp"])
    surrogate = torch.max(-b["adv"] * ratio, -b["adv"] * torch.clamp(ratio, 1 - ppo.clip_param, 1 + ppo.clip_param)).mean()
    v_clip = b["values"] + (value - b["values"]).clamp(-ppo.clip_param, ppo.clip_param)
    value_loss = torch.max((value - b["returns"]).square(), (v_clip - b["returns"]).square()).mean()
    mu, sigma = ac.action_mean, ac.action_std
    kl = torch.sum(torch.log(sigma / sigma_old + 1e-5) + (sigma_old.square() + (b["mu"] - mu).square()) / (2 * sigma.square()) - 0.5, -1)
    return surrogate, value_loss, kl.mean()


@pytest.mark.parametrize("activation", ["elu", "relu"])
@pytest.mark.parametrize("B,mb_start,mb_size", [(4096, 0, 4096), (5000, 700, 3001), (40, 3, 21)])
def test_fused_gradients_match_autograd(activation, B, mb_start, mb_size):
    """every parameter gradient of the surrogate + value loss (no entropy term, no clipping) vs torch autograd on the
    permuted minibatch: f32 / split-bf16 MFMA sums vs rocBLAS sums -> 2e-4 relative to the gradient's scale; ragged sizes cover
    partial 16-sample tiles and a minibatch that starts inside the permutation"""
    from wheeledlab_amd.rl.ppo import FusedPpoStep, PPO
    ac, flat, sigma_old = _problem(B, activation)
    ppo = PPO(ac)
    fused = FusedPpoStep(ac, ppo)
    perm = torch.randperm(B, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)).to(torch.int32)
    grad = fused.gradients(flat, perm, mb_start, mb_size, sigma_old).clone()
    torch.cuda.synchronize()
    idx = perm[mb_start:mb_start + mb_size].long()
    b = {k: v[idx] for k, v in flat.items()}
    surrogate, value_loss, kl = _torch_loss(ac, ppo, b, sigma_old)
    ac.zero_grad()
    (surrogate + ppo.value_loss_coef * value_loss).backward()
    want = torch.cat([p.grad.reshape(-1) for p in ac.parameters()])
    names = [n for n, _ in ac.named_parameters()]
    off = 0
    for name, p in zip(names, ac.parameters()):
        k = p.numel()
        got_p, want_p = grad[off:off + k], want[off:off + k]
        scale = float(want_p.abs().max()) + 1e-12
        err = float((got_p - want_p).abs().max())
        assert err < 2e-4 * scale + 1e-7, (name, err, scale)
        off += k
    assert off == 10437
    stats = grad[10437:10440] / mb_size
    assert abs(float(stats[0]) - float(value_loss)) < 1e-4 * (1 + abs(float(value_loss)))
    assert abs(float(stats[1]) - float(surrogate)) < 1e-4 * (1 + abs(float(surrogate)))
    assert abs(float(stats[2]) - float(kl)) < 1e-4 * (1 + abs(float(kl)))


def test_fused_step_tracks_the_torch_step():
    """12 consecutive minibatch steps (gradients -> entropy term -> norm clipping -> adaptive-KL learning rate -> Adam) by
    the fused kernels and by the torch implementation on copies of the same nets and the same minibatches: parameters
    stay equal to accumulated fp32 rounding, the learning-rate trajectory is identical"""
    from wheeledlab_amd.rl.ppo import FusedPpoStep, PPO
    B, mb = 8192, 2048
    ac_f, flat, sigma_old = _problem(B, "elu", seed=3)
    ac_t = copy.deepcopy(ac_f)
    pf, pt = PPO(ac_f, desired_kl=0.002), PPO(ac_t, desired_kl=0.002)   # the synthetic KL (~0.01) is far above 2 x target
    fused = FusedPpoStep(ac_f, pf)
    g = torch.Generator(device=DEV).manual_seed(9)
    lrs_f, lrs_t = [], []
    for step in range(12):
        if step % 4 == 0:
            perm = torch.randperm(B, device=DEV, generator=g).to(torch.int32)
        start = (step % 4) * mb
        fused.minibatch(flat, perm, start, mb, sigma_old)
        idx = perm[start:start + mb].long()
        pt._step({k: v[idx] for k, v in flat.items()}, sigma_old)
        lrs_f.append(fused.learning_rate)
        lrs_t.append(pt.learning_rate)
        for (name, a), b in zip(ac_f.named_parameters(), ac_t.parameters()):
            # Adam normalises every element's move to ~lr whatever the size of its gradient, so the few elements whose
            # gradient is within the split-bf16 rounding of zero (|g| ~ eps = 1e-8: their update lr g / (|g| + eps) is
            # ill-conditioned in g) move differently by a fraction of lr; everything else stays within the f32 bound
            d = (a - b).abs().flatten()
            tight = 2e-5 * (step + 1) + 2e-3 * lrs_t[-1] * (step + 1)
            assert float((d > tight).float().mean()) < 2e-3, (step, name, float((d > tight).float().mean()))
            assert float(d.max()) < 0.25 * max(lrs_t) * (step + 1), (step, name, float(d.max()))
    assert np.allclose(lrs_f, lrs_t, rtol=1e-6), (lrs_f, lrs_t)
    assert len(set(lrs_t)) > 1                                     # the adaptive rule actually moved the learning rate


def test_split_step_for_the_data_parallel_learner_equals_the_fused_step():
    """wl_ppo_gradients -> (all-reduce of the gradient row; a no-op with one rank) -> squared norm -> wl_ppo_apply is the
    same step as wl_ppo_minibatch: 8 steps on copies of the same nets; the only difference is the summation order of the
    squared norm (torch reduction vs the reduce kernel's atomics), i.e. the clipping coefficient to fp32 rounding"""
    from wheeledlab_amd.rl.ppo import FusedPpoStep, PPO
    B, mb = 8192, 2048
    ac_a, flat, sigma_old = _problem(B, "elu", seed=5)
    ac_b = copy.deepcopy(ac_a)
    fa, fb = FusedPpoStep(ac_a, PPO(ac_a, desired_kl=0.002)), FusedPpoStep(ac_b, PPO(ac_b, desired_kl=0.002))
    perm = torch.randperm(B, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)).to(torch.int32)
    for step in range(8):
        start = (step % 4) * mb
        fa.minibatch(flat, perm, start, mb, sigma_old)
        fb.minibatch(flat, perm, start, mb, sigma_old, split=True)
        assert fa.learning_rate == fb.learning_rate
        for (name, a), b in zip(ac_a.named_parameters(), ac_b.parameters()):
            torch.testing.assert_close(b, a, rtol=0, atol=2e-6 * (step + 1), msg=f"step {step} {name}")
    torch.testing.assert_close(fb.adam_m, fa.adam_m, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(fb.ctrl[4:7], fa.ctrl[4:7], rtol=1e-6, atol=0)
    assert fa.adam_step == fb.adam_step == 8 and fa.parity == fb.parity


def test_update_with_the_fused_step_equals_the_torch_update_and_checkpoints_round_trip(tmp_path):
    """PPO.update(fused_update=True) vs PPO.update(fused_update=False) on one storage with the same permutations; the
    optimizer state written by the fused learner loads into the torch learner (rsl_rl's checkpoint format) and back"""
    from wheeledlab_amd.policy import RolloutStorage
    from wheeledlab_amd.rl.ppo import ActorCritic, PPO
    torch.manual_seed(4)
    n, K = 1024, 16
    ac_f = ActorCritic(14, 14, 2).to(DEV)
    ac_t = copy.deepcopy(ac_f)
    st = RolloutStorage(K, n, device=DEV)
    st.observations.normal_()
    with torch.no_grad():
        ac_t.update_distribution(st.observations[:K].reshape(K * n, 14))
        a = ac_t.distribution.sample()
        st.actions.copy_(a.reshape(K, n, 2))
        st.mu.copy_(ac_t.action_mean.reshape(K, n, 2))
        st.actions_log_prob.copy_(ac_t.get_actions_log_prob(a).reshape(K, n))
        st.values.copy_(ac_t.evaluate(st.observations.reshape((K + 1) * n, 14)).reshape(K + 1, n))
    st.rewards.normal_()
    st.dones.copy_((torch.rand(K, n, device=DEV) < 0.05).long())
    pf, pt = PPO(ac_f, fused_update=True), PPO(ac_t, fused_update=False)
    for it in range(2):
        lf = pf.update(st, generator=torch.Generator(device=DEV).manual_seed(20 + it))
        lt = pt.update(st, generator=torch.Generator(device=DEV).manual_seed(20 + it))
        for (name, p), q in zip(ac_f.named_parameters(), ac_t.parameters()):
            _close_up_to_adam_noise(p, q, max(1e-3, lt["learning_rate"]), (it + 1) * 20, (it, name))   # the schedule starts at 1e-3
        assert abs(lf["learning_rate"] - lt["learning_rate"]) < 1e-9
        assert abs(lf["kl"] - lt["kl"]) < 1e-4 and abs(lf["surrogate"] - lt["surrogate"]) < 1e-4
        assert abs(lf["value_function"] - lt["value_function"]) < 1e-3 * (1 + abs(lt["value_function"]))
    # fused -> checkpoint format -> torch learner: the next torch update continues from the fused moments
    sd = copy.deepcopy(pf.optimizer_state_dict())
    ac_r = copy.deepcopy(ac_f)
    pr = PPO(ac_r, fused_update=False)
    pr.load_optimizer_state(sd)
    pf.update(st, generator=torch.Generator(device=DEV).manual_seed(77))
    pr.update(st, generator=torch.Generator(device=DEV).manual_seed(77))
    for (name, p), q in zip(ac_f.named_parameters(), ac_r.parameters()):
        _close_up_to_adam_noise(p, q, 1e-2, 20, name)


def test_gae_kernel_matches_the_oracle():
    from oracle import policy as OPOL
    from wheeledlab_amd.policy import RolloutStorage
    rng = np.random.default_rng(2)
    K, n = 37, 1000
    st = RolloutStorage(K, n, device=DEV)
    st.rewards.copy_(torch.from_numpy(rng.normal(size=(K, n)).astype(np.float32)))
    st.values.copy_(torch.from_numpy(rng.normal(size=(K + 1, n)).astype(np.float32)))
    st.dones.copy_(torch.from_numpy((rng.random((K, n)) < 0.1).astype(np.int64)))
    ret, adv_n = st.compute_returns(0.99, 0.95)
    ret_o, adv_o = OPOL.compute_returns(st.rewards.cpu().numpy(), st.values.cpu().numpy(), st.dones.cpu().numpy(), 0.99, 0.95)
    np.testing.assert_allclose(ret.cpu().numpy(), ret_o, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(adv_n.cpu().numpy(), (adv_o - adv_o.mean()) / (adv_o.std(ddof=1) + 1e-8), rtol=1e-4, atol=1e-4)
