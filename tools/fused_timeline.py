"""Where the time goes INSIDE the fused elevation launch: a debug build of the library (tools/build_variants.sh wl_elev.hip
tl:"-DWL_FUSED_TIMELINE=1" [tl_base:"-DWL_FUSED_TIMELINE=1 -DWL_FUSED_SCAN_PIPE=0"]) stamps the 100 MHz wall clock at the phase
boundaries of every block; this script steps 4096 envs, reads the stamps of the LAST launch and prints, in microseconds from the
first block's entry: when blocks enter, when the physics wavefront starts / ends its sub-steps, when it reaches the barrier, when
the scan starts, when a wavefront has issued its last store and when that store is acknowledged.
usage: fused_timeline.py [n_envs]"""
import ctypes, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wheeledlab_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
names = {0: "entry", 1: "substeps start", 2: "substeps end", 3: "physics wave done (tail + frame)", 4: "wave0 past barrier",
         5: "wave0 last store issued", 6: "wave0 stores acknowledged", 8: "wave7 past barrier", 9: "wave7 last store issued",
         10: "wave7 stores acknowledged"}
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_tl*.so"))):
    A._lib = None
    lib = A.load(path)
    from wheeledlab_amd.core import ElevBatch
    env = ElevBatch(n, device="cuda:0", seed=42)
    env.reset()
    a = torch.rand(32, n, 2, device="cuda:0") * 2 - 1
    env.rollout(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        env.rollout(a)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 128
    raw = ctypes.CDLL(path)
    raw.wl_debug_fused_timeline.argtypes = [ctypes.c_void_p]
    raw.wl_debug_fused_timeline.restype = ctypes.c_int
    buf = np.zeros((2048, 16), dtype=np.uint64)
    rc = raw.wl_debug_fused_timeline(buf.ctypes.data)
    nb = (n + 15) // 16
    t = buf[:nb].astype(np.int64)
    t0 = t[:, 0].min()
    rel = (t - t0) * 0.01            # 100 MHz ticks -> us
    out = {"lib": os.path.basename(path), "n_envs": n, "us_per_launch": round(us, 2), "rc": rc, "blocks": nb, "phases": {}}
    for k, nm in names.items():
        col = rel[:, k]
        out["phases"][nm] = {"min": round(float(col.min()), 2), "median": round(float(np.median(col)), 2), "max": round(float(col.max()), 2)}
    d = lambda a_, b_: round(float(np.median(rel[:, b_] - rel[:, a_])), 2)
    out["per_block_median"] = {"entry -> substeps start": d(0, 1), "substeps": d(1, 2), "tail": d(2, 3), "barrier": d(3, 4),
                               "scan (issue)": d(4, 5), "store drain": d(5, 6), "wave7 scan (issue)": d(8, 9), "wave7 drain": d(9, 10),
                               "block total": d(0, 6)}
    print(json.dumps(out), flush=True)
    del env
    torch.cuda.empty_cache()
