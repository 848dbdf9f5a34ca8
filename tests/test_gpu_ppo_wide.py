"""GPU parity of the fused PPO step of the WIDE agents (csrc/wl_ppo_wide.hip: elevation D = 689, visual D = 3208; agent
configs wheeledlab_tasks/{elevation,visual}/config/agents, driven through modified_rsl_rl_runner.py:104-118) against torch
autograd / torch.optim.Adam on the same minibatch.  The first layer runs on the bf16 matrix pipe with every f32 operand
split into two bf16 planes (16 mantissa bits, f32 accumulation): the tolerance is the 2e-4 (relative to each parameter
tensor's gradient scale) the drift agents' f32 kernel is held to."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _problem(B, D, activation="elu", seed=0):
    from wheeledlab_amd.rl.ppo import ActorCritic
    torch.manual_seed(seed)
    ac = ActorCritic(D, D, 2, activation=activation).to(DEV)
    with torch.no_grad():
        ac.std.copy_(torch.tensor([0.8, 1.1]))
        for p in ac.parameters():
            if p.dim() == 2:
                p.mul_(1.5)
    g = torch.Generator(device=DEV).manual_seed(seed + 1)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)
    obs = r(B, D)
    obs[:, : D // 2] *= 0.05          # mixed magnitudes, like body rates next to a clipped height map
    obs[:, -7:] = obs[:, -7:].clamp(-0.5, 0.5) * 4.0
    with torch.no_grad():
        ac.update_distribution(obs)
        actions = ac.distribution.sample()
        mu = ac.action_mean + 0.05 * r(B, 2)
        logp = ac.get_actions_log_prob(actions) + 0.1 * r(B)
        values = ac.evaluate(obs).squeeze(-1) + 0.3 * r(B)
    flat = dict(obs=obs, actions=actions.contiguous(), mu=mu.contiguous(), logp=logp.contiguous(), adv=r(B), returns=r(B),
                values=values.contiguous())
    sigma_old = torch.tensor([0.85, 1.05], device=DEV)
    return ac, flat, sigma_old


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


def _close_up_to_adam_noise(p, q, lr, steps, what):
    """parameters of two learners after `steps` Adam steps: all but a few elements within (rtol 1e-3, atol 2e-4); the
    elements whose gradient is within rounding of zero may differ by a fraction of lr per step (see the trajectory test)"""
    d = (p - q).abs()
    bad = d > 2e-4 + 1e-3 * q.abs()
    assert float(bad.float().mean()) < 2e-3, (what, float(bad.float().mean()))
    assert float(d.max()) < 0.25 * lr * steps + 2e-4, (what, float(d.max()))


def _bf16_planes(hi, lo):
    f = lambda t: (t.to(torch.int32) & 0xFFFF).bitwise_left_shift(16).view(torch.float32)
    return f(hi) + f(lo)


def test_staging_splits_and_transposes_the_permuted_rows():
    """staged column k of X^T = obs[perm[k]] as hi + lo bf16 planes (relative error <= 2^-16), blocked by 64 rows, padding zero"""
    from wheeledlab_amd.rl.ppo import FusedWidePpoStep, PPO
    B, D = 2048, 689
    ac, flat, _ = _problem(B, D)
    fz = FusedWidePpoStep(ac, PPO(ac), B, 1024)
    assert fz.dp == 704 and fz.G == sum(p.numel() for p in ac.parameters())
    perm = torch.randperm(B, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)).to(torch.int32)
    fz.stage(flat["obs"], perm)
    torch.cuda.synchronize()
    xt = _bf16_planes(fz.xt_hi, fz.xt_lo)                              # [row / 64][feature][row % 64]
    x = xt.transpose(1, 2).reshape(B, fz.dp)
    want = flat["obs"][perm.long()]
    err = (x[:, :D] - want).abs()
    assert float((err / want.abs().clamp_min(1e-30)).max()) <= 2.0 ** -16
    assert float(x[:, D:].abs().max()) == 0.0


@pytest.mark.parametrize("D,B,mb_start,mb_size,activation", [
    (689, 16384, 0, 16384, "elu"),          # the elevation agent
    (689, 12288, 4096, 8192, "relu"),       # a minibatch that starts inside the staged block
    (3208, 4096, 1024, 2048, "elu"),        # the visual agent (dp = 3264: 25.5 row blocks of the dW1 contraction)
    (100, 1024, 512, 512, "elu"),           # a narrow "wide" net: two overlapping K chunks, a single partly-filled row block
    (128, 2048, 1024, 1024, "relu"),        # D a multiple of 64: no overlapped chunk
    (689, 8192, 1600, 6400, "elu"),         # 100 K chunks at 2 per split: 50 of the 85 partial-sum rows, grid padded to 56
])
def test_wide_gradients_match_autograd(D, B, mb_start, mb_size, activation):
    from wheeledlab_amd.rl.ppo import FusedWidePpoStep, PPO
    ac, flat, sigma_old = _problem(B, D, activation)
    ppo = PPO(ac)
    fz = FusedWidePpoStep(ac, ppo, B, mb_size)
    perm = torch.randperm(B, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    if activation == "relu":
        # ReLU's derivative jumps at 0: a pre-activation within rounding of 0 (1 in ~1e6 at these sizes) may take the other
        # branch than torch's own GEMM and shift a weight gradient by |delta . x| -- not a rounding-sized difference, and
        # not an error of either side.  Keep such samples out of the minibatch under test (they sit ahead of mb_start).
        with torch.no_grad():
            x64 = flat["obs"].double()
            risky = torch.zeros(B, dtype=torch.bool, device=DEV)
            for net in (ac.actor, ac.critic):
                z1 = x64 @ net[0].weight.double().t() + net[0].bias.double()
                z2 = torch.relu(z1) @ net[2].weight.double().t() + net[2].bias.double()
                risky |= (z1.abs() < 1e-4).any(1) | (z2.abs() < 1e-5).any(1)
        assert 0 < int(risky.sum()) <= mb_start
        perm = torch.cat([perm[risky[perm]], perm[~risky[perm]]])
    perm = perm.to(torch.int32)
    fz.stage(flat["obs"], perm)
    grad = fz.gradients(flat, perm, mb_start, mb_size, sigma_old).clone()
    torch.cuda.synchronize()
    idx = perm[mb_start:mb_start + mb_size].long()
    b = {k: v[idx] for k, v in flat.items()}
    ac.tall_linear = False
    surrogate, value_loss, kl = _torch_loss(ac, ppo, b, sigma_old)
    ac.zero_grad()
    (surrogate + ppo.value_loss_coef * value_loss).backward()
    # the layer-1 activations the tail kernel consumed, against torch
    with torch.no_grad():
        act = torch.nn.functional.elu if activation == "elu" else torch.relu
        h1 = torch.cat([act(ac.actor[0](b["obs"])), act(ac.critic[0](b["obs"]))], 1)
    torch.testing.assert_close(fz.h1[:mb_size], h1, rtol=2e-4, atol=2e-5)
    off = 0
    for name, p in ac.named_parameters():
        k = p.numel()
        got_p, want_p = grad[off:off + k], p.grad.reshape(-1)
        scale = float(want_p.abs().max()) + 1e-12
        err = float((got_p - want_p).abs().max())
        assert err < 2e-4 * scale + 1e-7, (name, err, scale)
        off += k
    assert off == fz.G
    stats = grad[fz.G:fz.G + 3] / mb_size
    assert abs(float(stats[0]) - float(value_loss)) < 1e-4 * (1 + abs(float(value_loss)))
    assert abs(float(stats[1]) - float(surrogate)) < 1e-4 * (1 + abs(float(surrogate)))
    assert abs(float(stats[2]) - float(kl)) < 1e-4 * (1 + abs(float(kl)))
    # the squared norm the clipping uses
    n2 = float(fz.ctrl[2 + fz.parity])
    assert abs(n2 - float(grad[:fz.G].square().sum())) < 1e-4 * n2 + 1e-12


def test_wide_step_tracks_the_torch_step_and_the_split_form_equals_it():
    """8 consecutive minibatch steps by the fused kernels, by their data-parallel split form (gradients -> all-reduce (one
    rank: a no-op) -> apply) and by the torch implementation, on copies of the same nets: the learning-rate trajectory is
    identical, parameters stay together"""
    from wheeledlab_amd.rl.ppo import FusedWidePpoStep, PPO
    B, mb, D = 16384, 4096, 689
    ac_f, flat, sigma_old = _problem(B, D, "elu", seed=3)
    ac_s, ac_t = copy.deepcopy(ac_f), copy.deepcopy(ac_f)
    pf, ps, pt = PPO(ac_f, desired_kl=0.002), PPO(ac_s, desired_kl=0.002), PPO(ac_t, desired_kl=0.002, fused_update=False)
    ff, fs = FusedWidePpoStep(ac_f, pf, B, mb), FusedWidePpoStep(ac_s, ps, B, mb)
    g = torch.Generator(device=DEV).manual_seed(9)
    lrs_f, lrs_t = [], []
    for step in range(8):
        if step % 4 == 0:
            perm = torch.randperm(B, device=DEV, generator=g).to(torch.int32)
            ff.stage(flat["obs"], perm)
            fs.stage(flat["obs"], perm)
        start = (step % 4) * mb
        ff.minibatch(flat, perm, start, mb, sigma_old)
        fs.minibatch(flat, perm, start, mb, sigma_old, split=True)
        idx = perm[start:start + mb].long()
        pt._step({k: v[idx] for k, v in flat.items()}, sigma_old)
        lrs_f.append(ff.learning_rate)
        lrs_t.append(pt.learning_rate)
        assert fs.learning_rate == ff.learning_rate
        for (name, a), b, c in zip(ac_f.named_parameters(), ac_t.parameters(), ac_s.parameters()):
            # Adam normalises every element's move to ~lr whatever the size of its gradient, so the few elements whose
            # gradient is within the bf16-split rounding of zero (|g| ~ eps = 1e-8: their update lr g / (|g| + eps) is
            # ill-conditioned in g) move differently by a fraction of lr; everything else stays within the f32 kernel's bound
            d = (a - b).abs().flatten()
            tight = 2e-5 * (step + 1) + 2e-3 * lrs_t[-1] * (step + 1)
            assert float((d > tight).float().mean()) < 2e-3, (step, name, float((d > tight).float().mean()))
            assert float(d.max()) < 0.25 * max(lrs_t) * (step + 1), (step, name, float(d.max()))
            torch.testing.assert_close(c, a, rtol=0, atol=2e-6 * (step + 1), msg=f"split step {step} {name}")
    assert np.allclose(lrs_f, lrs_t, rtol=1e-6), (lrs_f, lrs_t)
    assert len(set(lrs_t)) > 1


def test_update_of_the_elevation_agent_uses_the_wide_step_and_equals_the_torch_update():
    """PPO.update on a storage of 689-wide observations: the auto-selected learner is the wide fused step; two updates vs
    the torch learner with the same permutations, and the optimizer state round-trips through rsl_rl's checkpoint format"""
    from wheeledlab_amd.policy import RolloutStorage
    from wheeledlab_amd.rl.ppo import ActorCritic, FusedWidePpoStep, PPO
    torch.manual_seed(4)
    n, K, D = 512, 16, 689
    ac_f = ActorCritic(D, D, 2).to(DEV)
    ac_t = copy.deepcopy(ac_f)
    st = RolloutStorage(K, n, obs_dim=D, device=DEV)
    st.observations.normal_()
    with torch.no_grad():
        ac_t.update_distribution(st.observations[:K].reshape(K * n, D))
        a = ac_t.distribution.sample()
        st.actions.copy_(a.reshape(K, n, 2))
        st.mu.copy_(ac_t.action_mean.reshape(K, n, 2))
        st.actions_log_prob.copy_(ac_t.get_actions_log_prob(a).reshape(K, n))
        st.values.copy_(ac_t.evaluate(st.observations.reshape((K + 1) * n, D)).reshape(K + 1, n))
    st.rewards.normal_()
    st.dones.copy_((torch.rand(K, n, device=DEV) < 0.05).long())
    pf, pt = PPO(ac_f), PPO(ac_t, fused_update=False)
    assert pf.fused_update and pf._wide
    for it in range(2):
        lf = pf.update(st, generator=torch.Generator(device=DEV).manual_seed(20 + it))
        lt = pt.update(st, generator=torch.Generator(device=DEV).manual_seed(20 + it))
        assert isinstance(pf._fused, FusedWidePpoStep)
        for (name, p), q in zip(ac_f.named_parameters(), ac_t.parameters()):
            _close_up_to_adam_noise(p, q, max(1e-3, lt["learning_rate"]), (it + 1) * 20, (it, name))   # the schedule starts at 1e-3
        assert abs(lf["learning_rate"] - lt["learning_rate"]) < 1e-9
        assert abs(lf["kl"] - lt["kl"]) < 1e-4 and abs(lf["surrogate"] - lt["surrogate"]) < 1e-4
        assert abs(lf["value_function"] - lt["value_function"]) < 1e-3 * (1 + abs(lt["value_function"]))
    sd = copy.deepcopy(pf.optimizer_state_dict())
    ac_r = copy.deepcopy(ac_f)
    pr = PPO(ac_r, fused_update=False)
    pr.load_optimizer_state(sd)
    pf.update(st, generator=torch.Generator(device=DEV).manual_seed(77))
    pr.update(st, generator=torch.Generator(device=DEV).manual_seed(77))
    for (name, p), q in zip(ac_f.named_parameters(), ac_r.parameters()):
        _close_up_to_adam_noise(p, q, 1e-2, 20, name)


def test_entry_points_refuse_what_the_kernels_cannot_do():
    import ctypes as C

    from wheeledlab_amd.rl.ppo import FusedWidePpoStep, PPO
    ac, flat, sigma_old = _problem(1024, 100)
    fz = FusedWidePpoStep(ac, PPO(ac), 1024, 512)
    perm = torch.arange(1024, device=DEV, dtype=torch.int32)
    fz.stage(flat["obs"], perm)
    bt = fz._batch({k: v.contiguous() for k, v in flat.items()}, perm, sigma_old)
    call = lambda start, size: fz.lib.wl_ppo_wide_gradients(C.byref(fz._actor), C.byref(fz._critic), ac.std.data_ptr(), C.byref(bt),
                                                            start, size, C.byref(fz.hp), C.byref(fz.state), 0, fz._stream())
    assert call(0, 512) == 0
    assert call(32, 512) == -1       # minibatch starts are multiples of 64
    assert call(0, 500) == -1
    assert call(768, 512) == -1      # runs past the staged rows
    assert fz.lib.wl_ppo_wide_stage(flat["obs"].data_ptr(), perm.data_ptr(), 1000, C.byref(fz.state), fz._stream()) == -1
    with pytest.raises(ValueError):
        FusedWidePpoStep(ac, PPO(ac), 1000, 500)
    torch.cuda.synchronize()
