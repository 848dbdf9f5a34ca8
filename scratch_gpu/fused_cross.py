import sys, os
sys.path.insert(0, os.getcwd())
import torch
from wheeledlab_amd.core import ElevBatch
def timed(fn, reps, per):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * per))
    return round(best, 2)
for n in (4096, 6144, 8192, 12288, 16384, 32768):
    row = {"n": n}
    for lanes in (4, 1):
        env = ElevBatch(n, device="cuda:0", seed=42)
        env.set_lanes(lanes)
        env.reset()
        K = 16
        a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
        env.rollout(a)
        row["fused" if lanes == 4 else "two_launches"] = timed(lambda: env.rollout(a), 4, K)
        del env
    print(row, flush=True)
