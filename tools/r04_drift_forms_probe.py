"""Round-4 probe: the drift step's forms per env count after the draws went from six Philox blocks to three: quad (lanes 4), lane form with
interleaved wheels (lanes 1), fenced low-register lane form (lanes 2), default -- us per step."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from wheeledlab_amd.core import DriftBatch


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


for n in (4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152):
    env = DriftBatch(n, device="cuda:0", seed=42)
    env.reset()
    env.set_dones_output(False)
    K = 8 if n > 100000 else 64
    a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
    env.rollout(a)
    res = {"n": n}
    for lanes in (0, 4, 1, 2, 0):
        if lanes == 4 and n > 262144:
            continue
        env.set_lanes(lanes)
        try:
            res[f"lanes{lanes}" + ("_again" if f"lanes{lanes}" in res else "")] = round(timed(lambda: env.rollout(a), 4) / K, 2)
        except Exception as ex:
            res[f"lanes{lanes}"] = str(ex)[:30]
    print(json.dumps(res), flush=True)
    del env, a
    torch.cuda.empty_cache()
