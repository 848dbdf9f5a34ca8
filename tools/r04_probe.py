"""Round-4 same-box A/B through WlEnvBuffers.flags (no rebuilds): height scan forms (gathers / LDS patches, streaming or not) of
the elevation task per env count -- observation launch alone and the whole lane-form step -- and the drift step's streaming /
cache-allocating forms with and without the int64 `dones` row.    usage: r04_probe.py [elev|drift|all]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import DriftBatch, ElevBatch

dev = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "all"


def timed(fn, reps, warm=2, blocks=3):
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


if what in ("elev", "all"):
    forms = {"gather": A.FLAG_SCAN_GATHER, "lds": A.FLAG_SCAN_LDS, "gather_nostream": A.FLAG_SCAN_GATHER | A.FLAG_NO_STREAM,
             "lds_nostream": A.FLAG_SCAN_LDS | A.FLAG_NO_STREAM, "gather_stream": A.FLAG_SCAN_GATHER | A.FLAG_STREAM,
             "lds_stream": A.FLAG_SCAN_LDS | A.FLAG_STREAM}
    for n in (4096, 16384, 65536, 262144, 1048576):
        env = ElevBatch(n, device=dev, seed=42)
        env.reset()
        K = 4 if n > 100000 else 16
        a = torch.rand(K, n, 2, device=dev) * 2 - 1
        env.rollout(a)
        res = {"task": "elev", "n": n}
        for name, fl in forms.items():
            env.set_flags(fl)
            env.set_lanes(1)
            res[f"observe_{name}_us"] = timed(env.observe, 8 if n > 100000 else 32)
            res[f"step_lane_{name}_us"] = round(timed(lambda: env.rollout(a), 2) / K, 2)
        env.set_flags(0)
        env.set_lanes(0)
        res["step_default_us"] = round(timed(lambda: env.rollout(a), 2) / K, 2)
        print(json.dumps(res), flush=True)
        del env, a
        torch.cuda.empty_cache()

if what in ("drift", "all"):
    for n in (262144, 1048576, 2097152, 4194304):
        env = DriftBatch(n, device=dev, seed=42)
        env.reset()
        a = torch.rand(8, n, 2, device=dev) * 2 - 1
        for _ in range(10):
            env.rollout(a)
        res = {"task": "drift", "n": n}
        for dn in (True, False):
            env.set_dones_output(dn)
            for name, fl in (("default", 0), ("stream", A.FLAG_STREAM), ("nostream", A.FLAG_NO_STREAM)):
                env.set_flags(fl)
                us = round(timed(lambda: env.rollout(a), 6, warm=3, blocks=4) / 8, 2)
                b = 342 if dn else 334
                res[f"{name}_{'dones' if dn else 'nodones'}"] = {"us": us, "frac": round(b * n / (us * 1e-6) / 8e12, 4)}
        print(json.dumps(res), flush=True)
        del env, a
        torch.cuda.empty_cache()
