"""Round-4 probe: what the drift step's random draws cost at large N.  Run-time switches (observation corruption, interval pushes) on
the shipped library, then every gpurun_variants/lib_rng*.so (tools/build_variants.sh wl_drift.hip rng7:"-DWL_PHILOX_ROUNDS=7" ...)."""
import glob, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import DriftBatch


def timed(fn, reps, warm=3, blocks=4):
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
    return best


def run(tag, n, corruption=1, pushes=1):
    env = DriftBatch(n, device="cuda:0", seed=42)
    env.p.enable_corruption, env.p.enable_pushes = corruption, pushes
    env.reset()
    env.set_dones_output(False)
    K = 8 if n > 100000 else 128
    a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
    for _ in range(10):
        env.rollout(a)
    us = timed(lambda: env.rollout(a), 6) / K
    print(json.dumps({"build": tag, "n": n, "corruption": corruption, "pushes": pushes, "us": round(us, 2),
                      "frac": round(334 * n / (us * 1e-6) / 8e12, 4)}), flush=True)
    del env, a
    torch.cuda.empty_cache()


libs = [("shipped", None)] + [(os.path.basename(q), q) for q in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_rng*.so")))]
for rep in range(2):
    for tag, path in libs:
        A._lib = None
        A.load(path) if path else A.load()
        for n in (4096, 65536, 1048576, 4194304):
            run(tag, n)
A._lib = None
A.load()
for n in (1048576, 4194304):
    for c, p in ((0, 1), (1, 0), (0, 0)):
        run("shipped", n, c, p)
