"""us per visual env.step(): a launch pair per step (wl_visual_rollout) against the persistent rollout
(wl_visual_rollout_persistent), with and without the augmentation; for every build in gpurun_variants/lib_*.so if there
are any (same-box A/B), else the installed library.  usage: visual_probe.py [n ...]"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import VisualBatch

K = 16
ns = [int(x) for x in sys.argv[1:]] or [1024, 4096]


def timed(fn, reps=6):
    best = 1e9
    for _ in range(3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * K))
    return round(best, 2)


for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_*.so"))) or [None]:
    if path:
        A._lib = None
        A.load(path)
    for n in ns:
        env = VisualBatch(n, device="cuda:0", seed=42)
        env.reset()
        a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
        obs = torch.zeros(K, n, env.OBS_DIM, device="cuda:0")
        rew = torch.zeros(K, n, device="cuda:0")
        term = torch.zeros(K, n, dtype=torch.bool, device="cuda:0")
        trunc = torch.zeros(K, n, dtype=torch.bool, device="cuda:0")
        bits = env._map.bits
        for aug, lds in (((1.0, 1.0, 0.0), True), ((1.0, 1.0, 0.0), False), ((1.3, 0.9, 1.2), True), ((1.3, 0.9, 1.2), False)):
            env._map.bits = bits if lds else None          # the camera's map in LDS (bits) / byte gathers from global memory
            env.p.brightness, env.p.contrast, env.p.blur_sigma = aug
            out = {"per_step_launches": timed(lambda: env.rollout(a, obs, rew, term, trunc)),
                   "camera_only": round(timed(lambda: [env.observe() for _ in range(K)]), 2),
                   "persistent": timed(lambda: env.rollout(a, obs, rew, term, trunc, persistent=True))}
            print(os.path.basename(path) if path else "installed", n, "plain" if aug[2] == 0.0 else "aug", "lds-bits" if lds else "byte-gathers", json.dumps(out), flush=True)
