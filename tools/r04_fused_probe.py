"""Round-4 A/B of the fused 4096-env elevation step (BASELINE config 3): scan phase from LDS patches (default when the kernel is
built with WL_FUSED_LDS_SCAN=1) against the gather scan (WL_FLAG_SCAN_GATHER), same box, same state.    usage: r04_fused_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import ElevBatch

dev = "cuda:0"


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


for n in (1024, 4096, 8192):
    env = ElevBatch(n, device=dev, seed=42)
    env.reset()
    K = 32
    a = torch.rand(K, n, 2, device=dev) * 2 - 1
    env.rollout(a)
    res = {"task": "elev_fused", "n": n}
    for rep in range(2):
        for name, fl in (("lds", 0), ("gather", A.FLAG_SCAN_GATHER)):
            env.set_flags(fl)
            res[f"step_{name}_us_{rep}"] = round(timed(lambda: env.rollout(a), 4) / K, 2)
    # the two forms must write the same observation bits
    env.set_flags(0); o0 = env.observe(torch.empty_like(env.obs)).clone()
    env.set_flags(A.FLAG_SCAN_GATHER); o1 = env.observe(torch.empty_like(env.obs)).clone()
    res["observe_equal"] = bool(torch.equal(o0, o1))
    s0 = env.state.clone()
    outs, sc = [], env.step_count
    for fl in (0, A.FLAG_SCAN_GATHER):
        env.state.copy_(s0)
        env.step_count = sc
        env.set_flags(fl)
        env.rollout(a[:4])
        outs.append(torch.cat([env.obs.flatten(), env.state.flatten()]).clone())
    res["rollout_equal"] = bool(torch.equal(outs[0], outs[1]))
    print(json.dumps(res), flush=True)
