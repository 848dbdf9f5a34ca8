"""Timing of the wide PPO step's launches (HIP events around each entry point) at the elevation / visual agents' sizes.
usage: python tools/ppo_wide_probe.py [D] [rows] [minibatch]"""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wheeledlab_amd import _abi  # noqa: E402
if os.environ.get("WL_PROBE_LIB"):          # A/B of another build of the library (gpurun_variants/lib_*.so)
    _abi.load(os.environ["WL_PROBE_LIB"])
from wheeledlab_amd.rl.ppo import ActorCritic, FusedWidePpoStep, PPO  # noqa: E402

DEV = "cuda:0"


def main():
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 689
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 524288
    mb = int(sys.argv[3]) if len(sys.argv) > 3 else B // 4
    torch.manual_seed(0)
    ac = ActorCritic(D, D, 2).to(DEV)
    ppo = PPO(ac)
    fz = FusedWidePpoStep(ac, ppo, B, mb)
    obs = torch.randn(B, D, device=DEV)
    flat = dict(obs=obs, actions=torch.randn(B, 2, device=DEV), mu=torch.randn(B, 2, device=DEV), logp=torch.randn(B, device=DEV) - 3,
                adv=torch.randn(B, device=DEV), returns=torch.randn(B, device=DEV), values=torch.randn(B, device=DEV))
    sigma_old = torch.ones(2, device=DEV)
    perm = torch.randperm(B, device=DEV).to(torch.int32)

    def timed(f, reps=10):
        f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    print(f"D {D} dp {fz.dp} rows {B} minibatch {mb} splits {fz.splits} params {fz.G}")
    print(f"stage                 {timed(lambda: fz.stage(obs, perm), 5):9.1f} us   ({B * fz.dp * 12 / 1e6:.0f} MB moved)")
    print(f"gradients (minibatch) {timed(lambda: fz.gradients(flat, perm, 0, mb, sigma_old)):9.1f} us")
    print(f"minibatch step        {timed(lambda: fz.minibatch(flat, perm, 0, mb, sigma_old)):9.1f} us")
    t0 = time.perf_counter()
    n = 0
    for _ in range(5):
        for i in range(B // mb):
            fz.minibatch(flat, perm, i * mb, mb, sigma_old)
            n += 1
    torch.cuda.synchronize()
    print(f"{n} steps back to back   {(time.perf_counter() - t0) / n * 1e6:9.1f} us / step (host clock)")


if __name__ == "__main__":
    main()
