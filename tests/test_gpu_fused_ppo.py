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
    permuted minibatch: f32 MFMA sums vs rocBLAS sums -> 2e-4 relative to the gradient's scale; ragged sizes cover
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
