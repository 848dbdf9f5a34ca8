"""Same-box A/B of several builds of the library (gpurun_variants/lib_*.so): us per env.step() of the drift step kernel at the
given env counts, of the persistent drift rollout, and of the elevation / visual steps at 4096 envs (best of 3)."""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import DriftBatch, ElevBatch, VisualBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,1048576").split(",")]
all_tasks = len(sys.argv) > 2 and sys.argv[2] == "all"


def timed(env, a, **kw):
    best = 1e9
    for trial in range(3):
        env.rollout(a, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            env.rollout(a, **kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (6 * a.shape[0]))
    return round(best, 2)


res = {}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_*.so"))):
    name = os.path.basename(path)[4:-3]
    A._lib = None
    A.load(path)
    for n in sizes:
        env = DriftBatch(n, device="cuda:0", seed=42)
        env.reset()
        a = torch.rand(128 if n <= 65536 else 8, n, 2, device="cuda:0") * 2 - 1
        res[f"{name}@{n}"] = timed(env, a)
        if all_tasks and n <= 32768:
            res[f"{name}@{n}:persistent"] = timed(env, a, persistent=True)
        del env, a
    if all_tasks:
        for task, cls, K in (("elev", ElevBatch, 32), ("visual", VisualBatch, 16)):
            env = cls(4096, device="cuda:0", seed=42)
            env.reset()
            a = torch.rand(K, 4096, 2, device="cuda:0") * 2 - 1
            res[f"{name}:{task}@4096"] = timed(env, a)
            del env, a
print(json.dumps(res))
