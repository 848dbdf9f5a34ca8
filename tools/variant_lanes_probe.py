"""Same-box A/B of library builds (gpurun_variants/lib_*.so) x step-kernel forms (WlEnvBuffers.lanes) x env counts:
us per fused drift env.step() with rollout storage, best of 3.  usage: variant_lanes_probe.py 1048576,4194304 1,2 [name-filter]"""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import DriftBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1048576,4194304").split(",")]
forms = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2").split(",")]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
res = {}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_*.so"))):
    name = os.path.basename(path)[4:-3]
    if flt and flt not in name:
        continue
    A._lib = None
    A.load(path)
    for n in sizes:
        env = DriftBatch(n, device="cuda:0", seed=42)
        env.reset()
        K = 128 if n <= 65536 else 8
        a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
        obs = torch.zeros(K, n, 14, device="cuda:0")
        rew = torch.zeros(K, n, device="cuda:0")
        term = torch.zeros(K, n, dtype=torch.uint8, device="cuda:0")
        trunc = torch.zeros(K, n, dtype=torch.uint8, device="cuda:0")
        for lanes in forms:
            env.set_lanes(lanes)
            best = 1e9
            for trial in range(3):
                env.rollout(a, obs, rew, term, trunc)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    env.rollout(a, obs, rew, term, trunc)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / (4 * K))
            res[f"{name}:lanes{lanes}@{n}"] = round(best, 2)
        del env, a, obs, rew, term, trunc
print(json.dumps(res))
