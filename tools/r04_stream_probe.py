"""Round-4 probe: non-temporal output rows BELOW the size thresholds (WL_FLAG_STREAM against the default / WL_FLAG_NO_STREAM), us per
step: elevation (lane form + scan launch; the fused form has its own build switch), visual (step + camera launch)."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import ElevBatch, VisualBatch


def timed(fn, reps, warm=3, blocks=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(blocks):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return round(best, 2)


for task, cls in (("elev", ElevBatch), ("visual", VisualBatch)):
    for n in (4096, 16384, 32768, 65536, 131072):
        env = cls(n, device="cuda:0", seed=42)
        env.reset()
        if task == "visual":
            env.sample_augmentation(torch.Generator().manual_seed(0))
        K = 8
        a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
        env.rollout(a)
        res = {"task": task, "n": n}
        for lanes in ((0, 1) if n <= 32768 else (0,)):
            for name, fl in (("default", 0), ("stream", A.FLAG_STREAM), ("nostream", A.FLAG_NO_STREAM), ("default2", 0)):
                if lanes == 0 and n <= 32768 and fl == A.FLAG_STREAM and task == "elev":
                    continue     # the quad / fused form is refused under WL_FLAG_STREAM
                env.set_lanes(lanes)
                env.set_flags(fl)
                try:
                    res[f"lanes{lanes}_{name}"] = round(timed(lambda: env.rollout(a), 3) / K, 2)
                except Exception as ex:
                    res[f"lanes{lanes}_{name}"] = str(ex)[:40]
        print(json.dumps(res), flush=True)
        del env, a
        torch.cuda.empty_cache()
