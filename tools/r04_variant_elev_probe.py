"""same-box A/B of library builds (gpurun_variants/lib_*.so): us per elevation env.step() in the lane form (step kernel + scan) and of
the step launch alone (= total - observe), per env count.   usage: r04_variant_elev_probe.py [n1,n2,...]"""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import ElevBatch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536,262144,1048576").split(",")]
def timed(fn, reps):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_*.so"))):
    A._lib = None; A.load(path)
    for n in sizes:
        env = ElevBatch(n, device="cuda:0", seed=42); env.reset(); env.set_lanes(1)
        K = 4
        a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
        for _ in range(6): env.rollout(a)      # cars settle and drive
        tot = timed(lambda: env.rollout(a), 2) / K
        ob = timed(env.observe, 8)
        print(json.dumps({"build": os.path.basename(path), "n": n, "step_total_us": round(tot, 1), "observe_us": round(ob, 1), "step_kernel_us": round(tot - ob, 1)}), flush=True)
        del env, a; torch.cuda.empty_cache()
