"""Workload for rocprofv3 --pmc passes: K fused drift steps at n envs (one rollout call), nothing else on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd.core import DriftBatch
n, K = int(sys.argv[1]), int(sys.argv[2])
env = DriftBatch(n, device="cuda:0", seed=42)
env.reset()
a = torch.rand(K, n, 2, device="cuda:0") * 2 - 1
env.rollout(a)
torch.cuda.synchronize()
