"""GPU parity of the one-launch policy step for wide observations (csrc/wl_actor.hip: wl_actor_critic_act) against the
numpy oracle of the same ActorCritic (oracle/policy.py: fp32 MLPs + the Philox / Box-Muller draw of the policy stream).
Tolerance: 3e-4 abs on means / values (fp32 dot products of up to 3208 terms in a different summation order), the draw
itself to 2e-5 (hardware log / sin / cos), log-prob 1e-3."""
import numpy as np
import pytest
import torch

from oracle import policy as OPOL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _nets(D, activation, seed):
    from wheeledlab_amd.policy import ActorCritic
    ac = ActorCritic(D, 2, activation, init_noise_std=1.0, device=DEV, seed=seed)
    ac.std.copy_(torch.tensor([0.7, 1.3], device=DEV))
    as_np = lambda m: dict(w1=m.w1.cpu().numpy(), b1=m.b1.cpu().numpy(), w2=m.w2.cpu().numpy(), b2=m.b2.cpu().numpy(),
                           w3=m.w3.cpu().numpy(), b3=m.b3.cpu().numpy(), activation=activation)
    return ac, as_np(ac.actor), as_np(ac.critic)


@pytest.mark.parametrize("activation", ["elu", "relu"])
@pytest.mark.parametrize("D,n", [(689, 4096), (3208, 512), (689, 100), (3208, 4096), (689, 16401), (14, 300), (16, 64), (33, 17), (1, 5)])
def test_policy_step_matches_oracle(D, n, activation):
    ac, actor_np, critic_np = _nets(D, activation, seed=D + n)
    g = torch.Generator(device=DEV).manual_seed(1)
    obs = torch.randn(n, D, device=DEV, generator=g)
    a, mu = torch.empty(n, 2, device=DEV), torch.empty(n, 2, device=DEV)
    logp, val = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    seed, step, off = 42, 1234567, 8192
    ac.act(obs, a, mu, logp, val, seed, step, env_offset=off)
    torch.cuda.synchronize()
    o_a, o_mu, o_logp = OPOL.act(actor_np, ac.std.cpu().numpy(), obs.cpu().numpy(), np.arange(n) + off, step, seed)
    o_val = OPOL.mlp(critic_np, obs.cpu().numpy())[:, 0]
    np.testing.assert_allclose(mu.cpu().numpy(), o_mu, rtol=0, atol=3e-4)
    np.testing.assert_allclose(val.cpu().numpy(), o_val, rtol=0, atol=3e-4)
    std = ac.std.cpu().numpy()
    np.testing.assert_allclose((a - mu).cpu().numpy() / std, (o_a - o_mu) / std, rtol=0, atol=2e-5)
    np.testing.assert_allclose(logp.cpu().numpy(), o_logp, rtol=0, atol=1e-3)
    # and against torch's own Normal on the kernel's numbers (what rsl_rl would compute for these actions)
    ref = torch.distributions.Normal(mu, ac.std).log_prob(a).sum(-1)
    torch.testing.assert_close(logp, ref, rtol=0, atol=1e-4)


def test_deterministic_step_and_strided_rows():
    """play policy (a = mu) on observation rows that are a column slice of a wider matrix (obs_stride > in_dim)"""
    D, n = 689, 1000
    ac, actor_np, _ = _nets(D, "relu", seed=5)
    wide = torch.randn(n, D + 7, device=DEV)
    obs = wide[:, 3:3 + D]
    a, mu = torch.empty(n, 2, device=DEV), torch.empty(n, 2, device=DEV)
    logp, val = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    ac.act(obs, a, mu, logp, val, 1, 2, deterministic=True)
    torch.cuda.synchronize()
    assert torch.equal(a, mu)
    np.testing.assert_allclose(mu.cpu().numpy(), OPOL.mlp(actor_np, obs.cpu().numpy()), rtol=0, atol=3e-4)


def test_matches_the_drift_rollout_draw():
    """for the 14-wide drift observation the step equals the first step of wl_drift_rollout_policy: same nets, same
    (seed, env, step) key -> same actions"""
    from wheeledlab_amd.core import DriftBatch
    from wheeledlab_amd.policy import ActorCritic, RolloutStorage
    n = 512
    env = DriftBatch(n, device=DEV, seed=9)
    env.reset()
    env.observe()
    ac = ActorCritic(device=DEV, seed=2)
    st = RolloutStorage(1, n, device=DEV)
    obs0 = env.obs.clone()
    a, mu = torch.empty(n, 2, device=DEV), torch.empty(n, 2, device=DEV)
    logp, val = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    ac.act(obs0, a, mu, logp, val, env.seed, env.step_count, env.env_offset)
    env.rollout_policy(ac, st)
    torch.cuda.synchronize()
    torch.testing.assert_close(st.mu[0], mu, rtol=0, atol=2e-5)
    torch.testing.assert_close(st.actions[0], a, rtol=0, atol=2e-5)
    torch.testing.assert_close(st.actions_log_prob[0], logp, rtol=0, atol=1e-5)
    torch.testing.assert_close(st.values[0], val, rtol=0, atol=2e-5)


def test_rejects_bad_arguments():
    import ctypes as C

    from wheeledlab_amd import _abi as A
    ac, _, _ = _nets(20, "elu", seed=0)
    lib = A.load()
    a, c = ac.actor.struct(), ac.critic.struct()
    obs = torch.zeros(4, 20, device=DEV)
    out2, out1 = torch.zeros(4, 2, device=DEV), torch.zeros(4, device=DEV)
    call = lambda n, stride, ap=a: lib.wl_actor_critic_act(C.byref(ap), C.byref(c), ac.std.data_ptr(), n, obs.data_ptr(), stride,
                                                          out2.data_ptr(), out2.data_ptr(), out1.data_ptr(), out1.data_ptr(), 0, 0, 0,
                                                          0, 3, None)
    assert call(4, 20) == 0
    assert call(0, 20) == -1 and call(4, 19) == -1      # WL_EINVAL
    bad = ac.critic.struct()       # an "actor" with one output
    assert call(4, 20, bad) == -1


def test_actor_and_critic_halves_equal_the_joint_launch():
    """nets = 1 (actor only) and nets = 2 (critic only, on a second stream) fill the same rows as the joint launch, bit for bit"""
    D, n = 689, 3000
    ac, _, _ = _nets(D, "relu", seed=11)
    obs = torch.randn(n, D, device=DEV)
    out = lambda: (torch.zeros(n, 2, device=DEV), torch.zeros(n, 2, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV))
    a0, m0, l0, v0 = out()
    ac.act(obs, a0, m0, l0, v0, 5, 77, env_offset=4096)
    a1, m1, l1, v1 = out()
    ac.act(obs, a1, m1, l1, v1, 5, 77, env_offset=4096, nets=1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ac.act(obs, a1, m1, l1, v1, 5, 77, env_offset=4096, nets=2)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(a0, a1) and torch.equal(m0, m1) and torch.equal(l0, l1) and torch.equal(v0, v1)
    assert float(v1.abs().max()) > 0 and float(a1.abs().max()) > 0
    v2 = torch.zeros(n, device=DEV)
    assert torch.equal(ac.values(obs, v2), v0)                       # the critic alone, no other outputs


@pytest.mark.parametrize("two_launches", [False, True])
@pytest.mark.parametrize("D,n,activation", [(689, 4096, "elu"), (3208, 1024, "elu"), (3208, 1000, "relu"), (100, 77, "elu"), (128, 300, "relu"),
                                            (3208, 2000, "elu"), (689, 1700, "relu")])
def test_bf16_plane_form_matches_oracle_and_the_f32_kernel(D, n, activation, two_launches):
    """wl_actor_critic_act_planes (layer 1 on the bf16 pipe, observation rows split hi + lo in registers: 16 mantissa bits), in
    its one-launch form (feature shares folded through LDS; row tiles per block 1 / 2 / 4 by the row count) and its two-launch
    form (split-K partial sums), against the numpy oracle at the f32 kernel's bars and against the f32 kernel itself; the
    draws are the same numbers.  Sizes cover a D that is a multiple of 64, D mod 64 != 0 (the overlapped last K chunk) and
    row counts off the tile grid."""
    ac, actor_np, critic_np = _nets(D, activation, seed=D + n)
    ac.planes_two_launch = two_launches
    g = torch.Generator(device=DEV).manual_seed(1)
    obs = torch.randn(n, D, device=DEV, generator=g)
    obs[:, : D // 3] *= 30.0                      # mixed magnitudes
    out = {}
    for planes in (False, True):
        ac.planes = planes
        a, mu = torch.empty(n, 2, device=DEV), torch.empty(n, 2, device=DEV)
        logp, val = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
        ac.act(obs, a, mu, logp, val, 42, 99, env_offset=512)
        torch.cuda.synchronize()
        out[planes] = (a, mu, logp, val)
    a, mu, logp, val = out[True]
    o_a, o_mu, o_logp = OPOL.act(actor_np, ac.std.cpu().numpy(), obs.cpu().numpy(), np.arange(n) + 512, 99, 42)
    o_val = OPOL.mlp(critic_np, obs.cpu().numpy())[:, 0]
    scale = max(1.0, float(np.abs(o_mu).max()), float(np.abs(o_val).max()))
    np.testing.assert_allclose(mu.cpu().numpy(), o_mu, rtol=0, atol=3e-4 * scale)
    np.testing.assert_allclose(val.cpu().numpy(), o_val, rtol=0, atol=3e-4 * scale)
    torch.testing.assert_close(mu, out[False][1], rtol=0, atol=2e-4 * scale)
    torch.testing.assert_close(val, out[False][3], rtol=0, atol=2e-4 * scale)
    torch.testing.assert_close(a - mu, out[False][0] - out[False][1], rtol=0, atol=2e-6)      # same draw, same std
    torch.testing.assert_close(logp, torch.distributions.Normal(mu, ac.std).log_prob(a).sum(-1), rtol=0, atol=1e-4)


@pytest.mark.parametrize("two_launches", [False, True])
def test_bf16_plane_form_is_independent_of_the_batch_a_row_arrives_in(two_launches):
    """two half-size calls with env_offset reproduce the full call bit for bit (the feature shares depend on the width
    only, whatever the row count), stale weight planes are refreshed unless the caller vouches for them, and the
    scratch is refused when it is too small"""
    import ctypes as C

    from wheeledlab_amd import _abi as A
    D, n = 3208, 4096            # the full call runs two row tiles per block, the halves one
    ac, _, _ = _nets(D, "elu", seed=3)
    ac.planes, ac.planes_two_launch = True, two_launches
    obs = torch.randn(n, D, device=DEV)
    full = [torch.empty(n, 2, device=DEV), torch.empty(n, 2, device=DEV), torch.empty(n, device=DEV), torch.empty(n, device=DEV)]
    ac.act(obs, *full, 7, 11, env_offset=0)
    half = [torch.empty_like(t) for t in full]
    for lo in range(0, n, n // 4):
        sl = slice(lo, lo + n // 4)
        ac.act(obs[sl], half[0][sl], half[1][sl], half[2][sl], half[3][sl], 7, 11, env_offset=lo)
    torch.cuda.synchronize()
    for x, y in zip(full, half):
        assert torch.equal(x, y)
    # parameters change behind the planes' back: the default call rebuilds them, planes_fresh=True keeps the stale ones
    ac.actor.w1.mul_(1.01)
    stale = [torch.empty_like(t) for t in full]
    ac.act(obs, *stale, 7, 11, planes_fresh=True)
    fresh = [torch.empty_like(t) for t in full]
    ac.act(obs, *fresh, 7, 11)
    torch.cuda.synchronize()
    assert torch.equal(stale[1], full[1]) and not torch.equal(fresh[1], full[1])
    assert torch.equal(stale[3], fresh[3])                                        # the critic did not change
    # too small a scratch
    sc = ac._scratch(n)
    a, c = ac._act_structs
    small = A.WlActScratch(sc.w_hi, sc.w_lo, sc.partials, sc.dp, sc.splits - 1, sc.rows_capacity, 0)
    call = lambda s, rows: A.load().wl_actor_critic_act_planes(
        C.byref(a), C.byref(c), ac.std.data_ptr(), rows, obs.data_ptr(), obs.stride(0), full[0].data_ptr(), full[1].data_ptr(),
        full[2].data_ptr(), full[3].data_ptr(), 0, 7, 11, 0, 3, C.byref(s), None)
    assert call(small, n) == -1 and call(sc, n) == 0
    rows_small = A.WlActScratch(sc.w_hi, sc.w_lo, sc.partials, sc.dp, sc.splits, n - 1, 0)
    assert call(rows_small, n) == -1
    torch.cuda.synchronize()


def test_batched_values_equal_the_policy_step_values():
    """ActorCritic.values_batched (a rollout's K + 1 observation rows through the streaming bf16 contraction, one sum per row and
    unit, in chunks) against the critic half of the f32 policy-step kernel: bf16-split rounding apart; independent of the chunking"""
    from wheeledlab_amd.policy import ActorCritic
    D, N = 689, 5000
    kac = ActorCritic(D, 2, "relu", device=DEV, seed=2)
    obs = torch.randn(N, D, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    want, got, got2 = (torch.zeros(N, device=DEV) for _ in range(3))
    kac.planes = False
    kac.values(obs, want)
    kac.values_batched(obs, got, chunk=2048)
    kac.values_batched(obs, got2, chunk=65536)
    torch.cuda.synchronize()
    torch.testing.assert_close(got, want, rtol=3e-4, atol=3e-4)
    assert torch.equal(got, got2)


@pytest.mark.parametrize("D,n_total", [(689, 8192), (3208, 4096), (3208, 8192)])
def test_default_form_of_a_shard_is_the_form_of_the_whole_batch(D, n_total):
    """ActorCritic.act picks its kernel (f32 / bf16 one launch / bf16 two launches) from the row count, and the
    crossovers sit between a shard's and the batch's size (D = 689: bf16 from 8192 rows, f32 below; D = 3208: two launches up to
    2048 rows, one up to 8192).  With `global_rows` set to the whole batch's row count a shard takes the batch's form and
    reproduces its rows bit for bit; and a larger batch on the same view (scratch reallocated) never runs on unbuilt planes."""
    ac, _, _ = _nets(D, "elu", seed=5)
    obs = torch.randn(n_total, D, device=DEV)
    full = [torch.empty(n_total, 2, device=DEV), torch.empty(n_total, 2, device=DEV), torch.empty(n_total, device=DEV), torch.empty(n_total, device=DEV)]
    ac.act(obs, *full, 7, 3)
    shard = [torch.empty_like(t) for t in full]
    ac.global_rows = n_total
    for lo in range(0, n_total, n_total // 4):
        sl = slice(lo, lo + n_total // 4)
        ac.act(obs[sl], shard[0][sl], shard[1][sl], shard[2][sl], shard[3][sl], 7, 3, env_offset=lo)
    torch.cuda.synchronize()
    for x, y in zip(full, shard):
        assert torch.equal(x, y)
    # the planes_fresh hazard: a small call builds the planes, a larger one reallocates the scratch -- the caller's
    # planes_fresh=True must not leave layer 1 running on zero-filled planes
    ac2, _, _ = _nets(D, "elu", seed=5)
    ac2.planes = True
    small = [t[:256].clone() for t in full]
    ac2.act(obs[:256], *small, 7, 3)
    big = [torch.empty_like(t) for t in full]
    ac2.act(obs, *big, 7, 3, planes_fresh=True)
    ref = [torch.empty_like(t) for t in full]
    ac2.act(obs, *ref, 7, 3)
    torch.cuda.synchronize()
    assert torch.equal(big[1], ref[1]) and torch.equal(big[3], ref[3])
