"""Workload for PMC passes on the observation kernels: `python obs_run.py elevation|visual n reps [aug]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd.core import ElevBatch, VisualBatch
name, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
env = (ElevBatch if name == "elevation" else VisualBatch)(n, device="cuda:0", seed=42)
if len(sys.argv) > 4:
    env.p.brightness, env.p.contrast, env.p.blur_sigma = 1.2, 0.9, 1.5
env.reset()
for _ in range(reps):
    env.observe()
torch.cuda.synchronize()
