"""Workload for rocprofv3 --pmc passes: K fused drift steps at n envs exactly as bench.py launches them (outputs into
[K, n, ...] rollout storage, no int64 `dones` row), nothing else on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd.core import DriftBatch
n, K = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
env = DriftBatch(n, device=dev, seed=42)
env.reset()
if len(sys.argv) > 3:
    env.set_lanes(int(sys.argv[3]))   # force a step-kernel form (WlEnvBuffers.lanes)
a = torch.rand(K, n, 2, device=dev) * 2 - 1
obs = torch.zeros(K, n, 14, device=dev)
rew = torch.zeros(K, n, device=dev)
term = torch.zeros(K, n, dtype=torch.uint8, device=dev)
trunc = torch.zeros(K, n, dtype=torch.uint8, device=dev)
env.rollout(a, obs, rew, term, trunc)
torch.cuda.synchronize()
