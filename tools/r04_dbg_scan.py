import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import ElevBatch
for n in (1000, 1536, 1537, 4000):
    env = ElevBatch(n, device="cuda:0", seed=3); env.reset(); env.set_lanes(1)
    env.set_flags(A.FLAG_SCAN_GATHER | A.FLAG_NO_STREAM); ref = env.observe().clone()
    env.set_flags(A.FLAG_SCAN_LDS | A.FLAG_NO_STREAM); env.obs.fill_(-7); got = env.observe().clone()
    torch.cuda.synchronize()
    bad = (got != ref).any(1)
    idx = bad.nonzero().flatten()
    print(n, "bad envs", int(bad.sum()), idx[:10].tolist(), idx[-5:].tolist())
    if len(idx):
        e = int(idx[0]); d = (got[e] != ref[e]).nonzero().flatten()
        print("  env", e, "bad cols", len(d), d[:8].tolist(), got[e, d[:4]].tolist(), ref[e, d[:4]].tolist())
