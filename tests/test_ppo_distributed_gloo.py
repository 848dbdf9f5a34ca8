"""world_size-2 test of the data-parallel learner on CPU (gloo): with the gradient (and the adaptive rule's KL mean) averaged
once per minibatch step, both ranks end an update with identical parameters and learning rate; fed the SAME rollout they
reproduce the single-process update ((g + g) / 2 == g; advantage moments over the global batch), fed different rollouts they differ from it."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _storage(seed, K=16, n=64):
    from wheeledlab_amd.policy import RolloutStorage
    g = torch.Generator().manual_seed(seed)
    st = RolloutStorage(K, n, 14, 2, "cpu")
    st.observations.copy_(torch.randn(K + 1, n, 14, generator=g))
    st.actions.copy_(torch.randn(K, n, 2, generator=g))
    st.mu.copy_(st.actions + 0.1 * torch.randn(K, n, 2, generator=g))
    st.actions_log_prob.copy_(-2.0 + 0.3 * torch.randn(K, n, generator=g))
    st.rewards.copy_(torch.randn(K, n, generator=g))
    st.values.copy_(torch.randn(K + 1, n, generator=g))
    st.dones.copy_((torch.rand(K, n, generator=g) < 0.05).to(st.dones.dtype))
    return st


def _update(seed_data, distributed):
    from wheeledlab_amd.rl.ppo import PPO, ActorCritic
    torch.manual_seed(0)
    ac = ActorCritic(14, 14, 2, actor_hidden_dims=[64, 64], critic_hidden_dims=[64, 64], activation="elu")
    alg = PPO(ac, num_learning_epochs=2, num_mini_batches=2, desired_kl=0.002, distributed=distributed)
    stats = alg.update(_storage(seed_data), generator=torch.Generator().manual_seed(7))
    return torch.cat([p.detach().reshape(-1) for p in ac.parameters()]), stats["learning_rate"], alg.world


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from wheeledlab_amd import dist as D
    D.init_from_env("gloo")
    same, lr_same, w = _update(11, None)            # both ranks see rollout 11
    assert w == world
    diff, lr_diff, _ = _update(20 + rank, None)     # rollouts 20 / 21
    # the end-of-run sync check of scripts/train_rl.py
    assert D.ranks_agree(same) and D.ranks_agree(diff) and not D.ranks_agree(torch.full((3,), float(rank)))
    torch.save({"same": same, "lr_same": lr_same, "diff": diff, "lr_diff": lr_diff}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_learner_keeps_parameters_in_sync(tmp_path):
    world = 2
    mp.start_processes(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True, start_method="spawn")
    r0, r1 = (torch.load(tmp_path / f"r{r}.pt") for r in range(world))
    torch.set_num_threads(1)
    single, lr_single, w = _update(11, False)
    assert w == 1
    # identical rollouts on both ranks: the averaged gradient IS the local one -> the single-process trajectory, up to the
    # advantage normalisation, whose moments are taken over the GLOBAL batch (two copies: the unbiased std divides by
    # 2 n - 1 instead of n - 1, a 2.4e-4 relative change of the advantages at n = 1024)
    assert torch.equal(r0["same"], r1["same"])
    torch.testing.assert_close(r0["same"], single, rtol=0, atol=2e-5)
    assert r0["lr_same"] == r1["lr_same"] == lr_single
    # different rollouts: ranks stay in lockstep with each other and take a step neither would take alone
    assert torch.equal(r0["diff"], r1["diff"]) and r0["lr_diff"] == r1["lr_diff"]
    alone, _, _ = _update(20, False)
    assert not torch.allclose(r0["diff"], alone, atol=1e-6)
