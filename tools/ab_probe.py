"""Round-5 same-box A/B: us per launch / step of the named workloads for the in-tree library and every build in
gpurun_variants/lib_*.so (tools/build_variants.sh).   usage: ab_probe.py workload[,workload...] [repeat]
workloads: drift4096 drift65536 drift1m elev4096 elevobs262144 elevobsg262144 (gather form) elevstep262144 elevstep1m visual4096 depth4096 vdtask4096"""
import glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dev = "cuda:0"
want = sys.argv[1].split(",") if len(sys.argv) > 1 else ["elev4096"]
repeat = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def timed(fn, reps, per=1, warm=2, blocks=4):
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
        best = min(best, e0.elapsed_time(e1) * 1e3 / (reps * per))
    return round(best, 2)


def run(name):
    from wheeledlab_amd.core import DepthCamera, DriftBatch, ElevBatch, VisualBatch, VisualDepthBatch
    if name.startswith("drift"):
        n = {"drift4096": 4096, "drift65536": 65536, "drift1m": 1048576, "drift4m": 4194304}[name]
        env = DriftBatch(n, device=dev, seed=42)
        env.set_dones_output(False)
        env.reset()
        K = 128 if n <= 65536 else 8
        a = torch.rand(K, n, 2, device=dev) * 2 - 1
        return timed(lambda: env.rollout(a), 8 if n <= 65536 else 6, K, warm=3)
    if name.startswith("elev"):
        n = int(name.replace("elevobsg", "").replace("elevobs", "").replace("elevstep", "").replace("elev", "").replace("1m", "1048576"))
        env = ElevBatch(n, device=dev, seed=42)
        if name.startswith("elevobsg"):
            env.set_flags(A.FLAG_SCAN_GATHER)
        env.reset()
        K = 32 if n <= 32768 else 4
        a = torch.rand(K, n, 2, device=dev) * 2 - 1
        env.rollout(a)
        if name.startswith("elevobs"):
            return timed(env.observe, 8)
        return timed(lambda: env.rollout(a), 4 if n <= 32768 else 3, K)
    if name == "visual4096":
        env = VisualBatch(4096, device=dev, seed=42)
        env.reset()
        env.sample_augmentation(torch.Generator().manual_seed(0))
        a = torch.rand(16, 4096, 2, device=dev) * 2 - 1
        return timed(lambda: env.rollout(a), 4, 16)
    if name == "depth4096":
        env = ElevBatch(4096, device=dev, seed=42)
        env.reset()
        env.rollout(torch.rand(8, 4096, 2, device=dev) * 2 - 1)
        cam = DepthCamera(env.hf, dev)
        img = torch.empty(4096, 60, 80, device=dev)
        return timed(lambda: cam.render(env, 100.0, img), 8, warm=3)
    if name == "vdtask4096":
        env = VisualDepthBatch(4096, device=dev, seed=42)
        env.reset()
        a = torch.rand(16, 4096, 2, device=dev) * 2 - 1
        a[:, :, 0] = a[:, :, 0].abs()
        return timed(lambda: env.rollout(a), 4, 16, warm=3)
    raise SystemExit(f"unknown workload {name}")


libs = [("default", A.LIB_PATH)] + [(os.path.basename(p)[4:-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_*.so")))]
for r in range(repeat):
    for lname, path in libs:
        A._lib = None
        A.load(path)
        res = {"lib": lname, "round": r}
        for w in want:
            res[w] = run(w)
            torch.cuda.empty_cache()
        print(json.dumps(res), flush=True)
