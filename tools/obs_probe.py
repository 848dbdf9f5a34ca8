"""HBM rate of the observation kernels alone (wl_*_observe) at several env counts."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd.core import ElevBatch, VisualBatch

dev = "cuda:0"
res = {}
for name, cls in (("elevation", ElevBatch), ("visual", VisualBatch)):
    for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,32768,131072").split(",")]:
        env = cls(n, device=dev, seed=42)
        env.reset()
        for _ in range(3):
            env.observe()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            env.observe()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        res[f"{name}@{n}"] = {"us": round(us, 1), "GBs": round(n * env.OBS_DIM * 4 / us / 1e3, 1)}
        if name == "visual":
            env.p.brightness, env.p.contrast, env.p.blur_sigma = 1.2, 0.9, 1.5
            env.observe(); torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                env.observe()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            res[f"{name}_aug@{n}"] = {"us": round(us, 1), "GBs": round(n * env.OBS_DIM * 4 / us / 1e3, 1)}
        del env
print(json.dumps(res))
