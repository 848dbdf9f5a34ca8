"""Time several builds of the library (gpurun_variants/lib_*.so) on the same workloads: us per fused step."""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import DriftBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,1048576").split(",")]
res = {}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_*.so"))):
    name = os.path.basename(path)[4:-3]
    A._lib = None
    A.load(path)
    for n in sizes:
        env = DriftBatch(n, device="cuda:0", seed=42)
        env.reset()
        K = 128 if n <= 65536 else 8
        a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
        best = 1e9
        for trial in range(3):
            env.rollout(a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                env.rollout(a)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / (6 * K))
        res[f"{name}@{n}"] = round(best, 2)
        del env, a
print(json.dumps(res))
