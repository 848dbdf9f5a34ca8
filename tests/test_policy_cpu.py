"""CPU checks of the policy-in-the-loop row: the oracle's MLP / Gaussian log-prob / GAE against
torch's own operators, the host storage against the oracle, the ABI structs against the header, misuse codes."""
import ctypes as C
import os
import subprocess

import numpy as np
import torch

from conftest import ROOT
from oracle import policy as OPOL
from wheeledlab_amd import _abi as A


def _net(in_dim, out_dim, activation, seed):
    g = torch.Generator().manual_seed(seed)
    lin = [torch.nn.Linear(in_dim, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, out_dim)]
    for m in lin:
        with torch.no_grad():
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.3)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.3)
    act = torch.nn.ELU() if activation == "elu" else torch.nn.ReLU()
    seq = torch.nn.Sequential(lin[0], act, lin[1], act, lin[2])
    net = dict(activation=activation)
    for i, m in enumerate(lin, 1):
        net[f"w{i}"], net[f"b{i}"] = m.weight.detach().numpy(), m.bias.detach().numpy()
    return seq, net


def test_oracle_mlp_matches_torch_sequential():
    for activation in ("elu", "relu"):
        seq, net = _net(14, 2, activation, 1)
        x = torch.randn(257, 14, generator=torch.Generator().manual_seed(2)) * 2
        with torch.no_grad():
            want = seq(x).numpy()
        np.testing.assert_allclose(OPOL.mlp(net, x.numpy()), want, rtol=1e-5, atol=1e-5)


def test_oracle_log_prob_matches_torch_normal():
    _, net = _net(14, 2, "elu", 3)
    obs = np.random.default_rng(0).normal(size=(64, 14)).astype(np.float32)
    std = np.array([0.7, 1.3], np.float32)
    a, mu, logp = OPOL.act(net, std, obs, np.arange(64), 5, 42)
    want = torch.distributions.Normal(torch.from_numpy(mu), torch.from_numpy(std)).log_prob(torch.from_numpy(a)).sum(-1)
    np.testing.assert_allclose(logp, want.numpy(), rtol=1e-4, atol=1e-4)
    # the sample really is mu + std * N(0,1): standardised residuals over many draws are ~ unit normal
    z = []
    for step in range(200):
        a, mu, _ = OPOL.act(net, std, obs, np.arange(64), step, 7)
        z.append((a - mu) / std)
    z = np.concatenate(z)
    assert abs(z.mean()) < 0.03 and abs(z.std() - 1.0) < 0.03


def test_gae_known_answer_and_host_storage_matches_oracle():
    from wheeledlab_amd.policy import RolloutStorage
    K, n, gamma, lam = 6, 5, 0.9, 0.8
    # known answer: reward 1, values 0, no dones -> advantage_t = sum_{j < K - t} (gamma lam)^j
    ret, adv = OPOL.compute_returns(np.ones((K, n), np.float32), np.zeros((K + 1, n), np.float32), np.zeros((K, n)), gamma, lam)
    want = np.array([sum((gamma * lam) ** j for j in range(K - t)) for t in range(K)], np.float32)
    np.testing.assert_allclose(adv[:, 0], want, rtol=1e-6)
    rng = np.random.default_rng(1)
    st = RolloutStorage(K, n, device="cpu")
    st.rewards.copy_(torch.from_numpy(rng.normal(size=(K, n)).astype(np.float32)))
    st.values.copy_(torch.from_numpy(rng.normal(size=(K + 1, n)).astype(np.float32)))
    st.dones.copy_(torch.from_numpy((rng.random((K, n)) < 0.3).astype(np.int64)))
    ret_h, adv_h = st.compute_returns(gamma, lam)
    ret_o, adv_o = OPOL.compute_returns(st.rewards.numpy(), st.values.numpy(), st.dones.numpy(), gamma, lam)
    np.testing.assert_allclose(ret_h.numpy(), ret_o, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(adv_h.numpy(), (adv_o - adv_o.mean()) / (adv_o.std(ddof=1) + 1e-8), rtol=1e-4, atol=1e-5)
    # time-out bootstrap (rsl_rl PPO.process_env_step)
    st.time_outs.copy_(torch.from_numpy(rng.random((K, n)) < 0.5))
    r0 = st.rewards.clone()
    st.bootstrap_time_outs(gamma)
    np.testing.assert_allclose(st.rewards.numpy(), (r0 + gamma * st.values[:-1] * st.time_outs).numpy(), rtol=1e-6)


def test_policy_structs_match_header_and_misuse_is_refused(tmp_path):
    probe = tmp_path / "probe3.c"
    probe.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "wheeledlab_amd.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %d %d\\n\", sizeof(WlMlp), sizeof(WlPolicyRollout), offsetof(WlMlp, in_dim),"
        " offsetof(WlMlp, activation), offsetof(WlPolicyRollout, dones), (int)WL_ACT_RELU, (int)WL_ACT_ELU);return 0;}\n")
    exe = tmp_path / "probe3"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [C.sizeof(A.WlMlp), C.sizeof(A.WlPolicyRollout), A.WlMlp.in_dim.offset, A.WlMlp.activation.offset,
                   A.WlPolicyRollout.dones.offset, A.ACT_RELU, A.ACT_ELU]
    import __graft_entry__ as g
    g.build()
    lib = A.load()
    buf = (C.c_float * 4096)()
    base = C.addressof(buf)
    ok = dict(w1=base, b1=base, w2=base, b2=base, w3=base, b3=base)
    good = A.WlMlp(in_dim=14, out_dim=2, hidden=64, activation=A.ACT_ELU, **ok)
    assert lib.wl_mlp_forward(None, 16, base, base, None) == -1
    assert lib.wl_mlp_forward(C.byref(good), 0, base, base, None) == -1
    assert lib.wl_mlp_forward(C.byref(A.WlMlp(in_dim=14, out_dim=2, hidden=32, activation=0, **ok)), 16, base, base, None) == -1
    assert lib.wl_mlp_forward(C.byref(A.WlMlp(in_dim=16, out_dim=2, hidden=64, activation=0, **ok)), 16, base, base, None) == -1
    assert lib.wl_mlp_forward(C.byref(A.WlMlp(in_dim=14, out_dim=5, hidden=64, activation=0, **ok)), 16, base, base, None) == -1
    assert lib.wl_mlp_forward(C.byref(A.WlMlp(in_dim=14, out_dim=2, hidden=64, activation=7, **ok)), 16, base, base, None) == -1
    from wheeledlab_amd import params as PP
    p = PP.drift_params()
    bufs = A.WlEnvBuffers(state=base, episode_len=base, ref_poses=base, metrics=base, stride=128, n_envs=100, env_offset=0,
                          metrics_slots=1)
    io = A.WlPolicyRollout(base, base, base, base, base, base, base, None)
    critic_shaped = A.WlMlp(in_dim=14, out_dim=1, hidden=64, activation=A.ACT_ELU, **ok)
    assert lib.wl_drift_rollout_policy(C.byref(p), C.byref(bufs), C.byref(critic_shaped), base, C.byref(io), 4, 0, 0, None) == -1
    assert lib.wl_drift_rollout_policy(C.byref(p), C.byref(bufs), C.byref(good), None, C.byref(io), 4, 0, 0, None) == -1
    assert lib.wl_drift_rollout_policy(C.byref(p), C.byref(bufs), C.byref(good), base, None, 4, 0, 0, None) == -1
    io_misaligned = A.WlPolicyRollout(base + 4, base, base, base, base, base, base, None)
    assert lib.wl_drift_rollout_policy(C.byref(p), C.byref(bufs), C.byref(good), base, C.byref(io_misaligned), 4, 0, 0, None) == -3


def test_ppo_structs_match_header_and_misuse_is_refused(tmp_path):
    probe = tmp_path / "probe4.c"
    probe.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "wheeledlab_amd.h"\n'
        "int main(){printf(\"%zu %zu %zu %zu %zu %zu %d %d %d %d %d %d %d\\n\", sizeof(WlPpoBatch), sizeof(WlPpoParams),"
        " sizeof(WlPpoState), offsetof(WlPpoBatch, sigma_old), offsetof(WlPpoParams, use_clipped_value_loss),"
        " offsetof(WlPpoState, operands), (int)WL_PPO_NUM_PARAMS, (int)WL_PPO_PARTIAL_STRIDE, (int)WL_PPO_BLOCKS,"
        " (int)WL_PPO_OPERAND_FLOATS, (int)WL_PPO_CTRL_LR, (int)WL_PPO_CTRL_NORM2, (int)WL_PPO_CTRL_STATS);return 0;}\n")
    exe = tmp_path / "probe4"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert got == [C.sizeof(A.WlPpoBatch), C.sizeof(A.WlPpoParams), C.sizeof(A.WlPpoState), A.WlPpoBatch.sigma_old.offset,
                   A.WlPpoParams.use_clipped_value_loss.offset, A.WlPpoState.operands.offset, A.PPO_NUM_PARAMS,
                   A.PPO_PARTIAL_STRIDE, A.PPO_BLOCKS, A.PPO_OPERAND_FLOATS, A.PPO_CTRL_LR, A.PPO_CTRL_NORM2, A.PPO_CTRL_STATS]
    # the flat parameter count is the ActorCritic's
    from wheeledlab_amd.rl.ppo import ActorCritic
    assert sum(p.numel() for p in ActorCritic(14, 14, 2).parameters()) == A.PPO_NUM_PARAMS
    assert [n for n, _ in ActorCritic(14, 14, 2).named_parameters()][:3] == ["std", "actor.0.weight", "actor.0.bias"]
    import __graft_entry__ as g
    g.build()
    lib = A.load()
    buf = (C.c_float * 4096)()
    base = C.addressof(buf)
    ok = dict(w1=base, b1=base, w2=base, b2=base, w3=base, b3=base)
    actor = A.WlMlp(in_dim=14, out_dim=2, hidden=64, activation=A.ACT_ELU, **ok)
    critic = A.WlMlp(in_dim=14, out_dim=1, hidden=64, activation=A.ACT_ELU, **ok)
    bt = A.WlPpoBatch(*([base] * 9))
    hp = A.WlPpoParams()
    st = A.WlPpoState(base, base, base, base, base, base)
    args = lambda a, c, b, s, par=0: (C.byref(a), C.byref(c), base, C.byref(b), 0, 16, C.byref(hp), C.byref(s), par)
    assert lib.wl_ppo_gradients(*args(critic, critic, bt, st), None) == -1            # actor must have 2 outputs
    relu_critic = A.WlMlp(in_dim=14, out_dim=1, hidden=64, activation=A.ACT_RELU, **ok)
    assert lib.wl_ppo_gradients(*args(actor, relu_critic, bt, st), None) == -1       # one activation for both nets
    assert lib.wl_ppo_gradients(*args(actor, critic, A.WlPpoBatch(), st), None) == -1
    assert lib.wl_ppo_gradients(*args(actor, critic, bt, A.WlPpoState(base, base, base, base, base, None)), None) == -1
    assert lib.wl_ppo_gradients(*args(actor, critic, bt, st, par=2), None) == -1
    assert lib.wl_ppo_minibatch(*args(actor, critic, bt, st), 0, None) == -1          # adam_step counts from 1
    assert lib.wl_gae(0, 16, base, base, base, 0.99, 0.95, base, base, None) == -1
    assert lib.wl_gae(4, 16, base, base, None, 0.99, 0.95, base, base, None) == -1
