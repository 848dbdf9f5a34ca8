"""us per env.step() of the three task backends (raw C-ABI rollout path) at given env counts."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd.core import DriftBatch, ElevBatch, VisualBatch

dev = "cuda:0"
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096").split(",")]
res = {}
for name, cls, K in (("drift", DriftBatch, 128), ("elevation", ElevBatch, 32), ("visual", VisualBatch, 16)):
    for n in sizes:
        env = cls(n, device=dev, seed=42)
        env.reset()
        a = torch.rand(K, n, 2, device=dev) * 2 - 1
        env.rollout(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            env.rollout(a)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (4 * K)
        res[f"{name}@{n}"] = {"us_per_step": round(us, 2), "env_steps_per_s": round(n / us * 1e6), "obs_GBs": round(n * env.OBS_DIM * 4 / us / 1e3, 1)}
        if name == "drift" and n <= 32768:
            env.rollout(a, persistent=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(4):
                env.rollout(a, persistent=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (4 * K)
            res[f"drift_persistent@{n}"] = {"us_per_step": round(us, 2), "env_steps_per_s": round(n / us * 1e6)}
        del env, a
print(json.dumps(res))
