"""us per elevation / visual env.step() vs decimation (sub-step slope vs fixed part), 4096 envs.  usage: elev_probe.py [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd.core import ElevBatch, VisualBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = {}
for name, cls, K in (("elev", ElevBatch, 32), ("visual", VisualBatch, 16)):
    env = cls(n, device="cuda:0", seed=42)
    env.reset()
    a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
    dec0 = env.p.decimation
    for dec in (1, 2, 5, 10):
        env.p.decimation = dec
        env.rollout(a)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            env.rollout(a)
        e1.record()
        torch.cuda.synchronize()
        out[f"{name}_dec{dec}"] = round(e0.elapsed_time(e1) * 1e3 / (4 * K), 2)
    env.p.decimation = dec0
print(json.dumps(out))
