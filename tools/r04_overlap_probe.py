"""Round-4 probe: the elevation / visual env.step() of a large batch as TWO half batches (env shards by env_offset) on two HIP streams, so
that one half's observation launch (scan / camera: memory + LDS) runs beside the other half's step launch (dependent sub-steps: latency):
us per step of the whole batch, one stream against two."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from wheeledlab_amd.core import ElevBatch, VisualBatch

dev = "cuda:0"


def timed(fn, reps, warm=2, blocks=4):
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


for task, cls in (("elev", ElevBatch), ("visual", VisualBatch)):
    for n in (65536, 262144, 1048576) if task == "elev" else (65536, 262144):
        K = 4
        res = {"task": task, "n": n}
        env = cls(n, device=dev, seed=42)
        env.reset()
        if task == "visual":
            env.sample_augmentation(torch.Generator().manual_seed(0))
        a = torch.rand(K, n, 2, device=dev) * 2 - 1
        env.rollout(a)
        res["one_stream_us"] = round(timed(lambda: env.rollout(a), 3) / K, 1)
        del env
        torch.cuda.empty_cache()
        for parts in (2, 4):
            h = n // parts
            envs = [cls(h, device=dev, seed=42, env_offset=i * h) for i in range(parts)]
            acts = [a[:, i * h:(i + 1) * h].contiguous() for i in range(parts)]
            streams = [torch.cuda.Stream(device=dev) for _ in range(parts)]
            for e in envs:
                e.reset()
                if task == "visual":
                    e.sample_augmentation(torch.Generator().manual_seed(0))
            torch.cuda.synchronize()

            def step_all():
                cur = torch.cuda.current_stream()
                for s in streams:
                    s.wait_stream(cur)
                for k in range(K):       # launch order: interleave the parts step by step
                    for e, s, ai in zip(envs, streams, acts):
                        with torch.cuda.stream(s):
                            e.rollout(ai[k:k + 1])
                for s in streams:
                    cur.wait_stream(s)
            res[f"{parts}_streams_us"] = round(timed(step_all, 3) / K, 1)
            del envs, acts, streams
            torch.cuda.empty_cache()
        print(json.dumps(res), flush=True)
        del a
        torch.cuda.empty_cache()
