"""Round-4 probe: the visual camera launch's rows as non-temporal stores at every size (gpurun_variants/lib_vnt.so) against the
256 MB threshold (lib_vbase.so): us per env.step() in the default forms."""
import glob, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import VisualBatch


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


for rep in range(2):
    for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_v*.so"))):
        A._lib = None
        A.load(path)
        res = {"build": os.path.basename(path)}
        for n in (1024, 4096, 8192, 16384):
            env = VisualBatch(n, device="cuda:0", seed=42)
            env.reset()
            env.sample_augmentation(torch.Generator().manual_seed(0))
            a = torch.rand(16, n, 2, device="cuda:0") * 2 - 1
            env.rollout(a)
            res[f"step_{n}"] = round(timed(lambda: env.rollout(a), 3) / 16, 2)
            res[f"observe_{n}"] = timed(env.observe, 16)
            del env, a
            torch.cuda.empty_cache()
        print(json.dumps(res), flush=True)
