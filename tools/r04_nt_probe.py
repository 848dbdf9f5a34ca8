"""Round-4 probe: non-temporal output rows in the launches whose counter traffic exceeds the algorithmic bytes (fused elevation step
at 4096 envs, depth render) and in the visual camera launch: us per launch per library of gpurun_variants/ (tools/build_variants.sh with
-DWL_FUSED_SCAN_NT=false|true, -DWL_VIS_STREAM_BYTES=..., the depth store edited by hand); the FETCH_SIZE passes are in tools/r04_nt_pmc.sh."""
import glob, json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import DepthCamera, ElevBatch, VisualBatch


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


libs = sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_*.so")))
for rep in range(2):
    for path in libs:
        A._lib = None
        A.load(path)
        n = 4096
        env = ElevBatch(n, device="cuda:0", seed=42)
        env.reset()
        a = torch.rand(32, n, 2, device="cuda:0") * 2 - 1
        env.rollout(a)
        res = {"build": os.path.basename(path), "elev_step_us": round(timed(lambda: env.rollout(a), 4) / 32, 2)}
        cam = DepthCamera((env.height, float(env._hf.x0), float(env._hf.y0), float(env._hf.cell)), "cuda:0")
        out = torch.empty(n, 60, 80, device="cuda:0")
        res["depth_us"] = timed(lambda: cam.render(env, 100.0, out), 4)
        del env, cam, out, a
        torch.cuda.empty_cache()
        env = VisualBatch(n, device="cuda:0", seed=42)
        env.reset()
        env.sample_augmentation(torch.Generator().manual_seed(0))
        a = torch.rand(16, n, 2, device="cuda:0") * 2 - 1
        env.rollout(a)
        res["visual_step_us"] = round(timed(lambda: env.rollout(a), 3) / 16, 2)
        print(json.dumps(res), flush=True)
        del env, a
        torch.cuda.empty_cache()
