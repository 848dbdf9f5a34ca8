import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import ElevBatch, VisualBatch, VisualDepthBatch
dev = "cuda:0"
if len(sys.argv) > 1: A._lib = None; A.load(sys.argv[1])
def timed(fn, reps, per, warm=2, blocks=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); best = 1e30
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * per))
    return round(best, 2)
res = {}
for name, cls, n in (("elev", ElevBatch, 4096), ("visual", VisualBatch, 4096), ("vdtask", VisualDepthBatch, 4096), ("elev262k", ElevBatch, 262144)):
    env = cls(n, device=dev, seed=42); env.reset()
    K = 16 if n <= 32768 else 4
    a = torch.rand(K, n, 2, device=dev) * 2 - 1
    env.rollout(a)
    for dec in (10, 1):
        env.p.decimation = dec
        res[f"{name}_dec{dec}"] = timed(lambda: env.rollout(a), 4, K)
    env.p.decimation = 10
print(json.dumps(res))
