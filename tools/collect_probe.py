"""us per collection step of the elevation task: the persistent collector (wl_elev_collect_rollout: K steps per launch) and the
one-launch-per-step collector (wl_elev_collect_step) against policy step + env step (wl_actor_critic_act; wl_elev_step).
usage: collect_probe.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
if os.environ.get("WL_LIB"):
    A.load(os.environ["WL_LIB"])   # a variant build (gpurun_variants/lib_*.so)
from wheeledlab_amd.core import ElevBatch
from wheeledlab_amd.policy import RolloutStorage
from wheeledlab_amd.rl.ppo import ActorCritic

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K, D = 32, 689
ac = ActorCritic(D, D, 2).to(DEV)
view = ac.fused()
view.planes = False
env = ElevBatch(n, device=DEV, seed=11)
env.reset()
st = RolloutStorage(K, n, D, 2, DEV)
st.observations[0].copy_(env.observe())


def one():
    for k in range(K):
        env.collect_step(view, st, k)


def two():
    for k in range(K):
        view.act(st.observations[k], st.actions[k], st.mu[k], st.actions_log_prob[k], st.values[k], env.seed, env.step_count, env.env_offset)
        env.rollout(st.actions[k:k + 1], st.observations[k + 1:k + 2], st.rewards[k:k + 1], st.terminated[k:k + 1], st.time_outs[k:k + 1],
                    dones_out=st.dones[k:k + 1])


def persistent():
    env.collect_rollout(view, st)


for name, fn in (("persistent collector (one launch per rollout)", persistent), ("one launch per step", one), ("policy step + env step", two)):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 1e3 / (4 * K):.1f} us per collection step at {n} envs")
