import glob, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import ElevBatch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_e*.so"))):
    A._lib = None; A.load(path)
    for n in (65536, 262144):
        env = ElevBatch(n, device="cuda:0", seed=42); env.reset(); env.set_lanes(1)
        res = {"build": os.path.basename(path), "n": n}
        for name, fl in (("lds_stream", 4 | 1), ("gather_stream", 8 | 1)):
            env.set_flags(fl)
            for _ in range(3): env.observe()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): env.observe()
            e1.record(); torch.cuda.synchronize()
            res[name] = round(e0.elapsed_time(e1) * 1e3 / 8, 1)
        print(json.dumps(res), flush=True)
        del env; torch.cuda.empty_cache()
