"""Does running the collection loop of G independent env groups on G streams pay?  Each group's K x { policy step ->
env.step (+ scan / camera) } is captured as a HIP graph (no host in the loop) and the G graphs are replayed concurrently.
usage: python tools/group_probe.py [elev|visual] [n_total] [K]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wheeledlab_amd.core import ElevBatch, VisualBatch  # noqa: E402
from wheeledlab_amd.policy import RolloutStorage  # noqa: E402
from wheeledlab_amd.rl.ppo import ActorCritic  # noqa: E402

DEV = "cuda:0"


def main():
    task = sys.argv[1] if len(sys.argv) > 1 else "elev"
    n_total = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    Batch = ElevBatch if task == "elev" else VisualBatch
    D = Batch.OBS_DIM
    torch.manual_seed(0)
    ac = ActorCritic(D, D, 2).to(DEV)
    view = ac.fused()
    for G in (1, 2, 4, 8):
        n = n_total // G
        groups = []
        for g in range(G):
            b = Batch(n, device=DEV, seed=3, env_offset=g * n)
            b.reset()
            st = RolloutStorage(K, n, D, 2, DEV)
            st.observations[0].copy_(b.observe())
            groups.append((b, st, torch.cuda.Stream(device=DEV)))
        torch.cuda.synchronize()

        def loop(b, st):
            for k in range(K):
                view.act(st.observations[k], st.actions[k], st.mu[k], st.actions_log_prob[k], st.values[k], b.seed, b.step_count, b.env_offset)
                b.rollout(st.actions[k:k + 1], st.observations[k + 1:k + 2], st.rewards[k:k + 1], st.terminated[k:k + 1],
                          st.time_outs[k:k + 1], dones_out=st.dones[k:k + 1])

        graphs = []
        for b, st, s in groups:
            with torch.cuda.stream(s):
                loop(b, st)                      # warm-up (also outside capture)
            s.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                loop(b, st)
            graphs.append(gr)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for (b, st, s), gr in zip(groups, graphs):
                with torch.cuda.stream(s):
                    gr.replay()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{task} n_total {n_total} groups {G} x {n} envs, K {K}: {best / K * 1e6:8.1f} us per step of the whole batch "
              f"({n_total * K / best:.3e} env-steps/s)")


if __name__ == "__main__":
    main()
