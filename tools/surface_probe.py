#!/usr/bin/env python3
"""Host cost of one env.step() through the Python surface, before / after torch's GEMM libraries are initialised in the
same process (a rsl_rl training process always has them)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from wheeledlab_amd import registry, tasks  # noqa: F401,E402
from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper  # noqa: E402

n, dev = 4096, "cuda:0"
cfg = registry.parse_env_cfg("Isaac-MushrDriftRL-v0", device=dev, num_envs=n)
e = registry.make("Isaac-MushrDriftRL-v0", cfg=cfg)
e.action_space.low, e.action_space.high = -1.0, 1.0
w = RslRlVecEnvWrapper(ClipAction(e))
actions = torch.rand(128, n, 2, device=dev) * 2 - 1


def rate(tag, steps=1024):
    for i in range(64):
        w.step(actions[i % 128])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(steps):
        w.step(actions[i % 128])
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / steps * 1e6
    print(f"{tag}: {us:.1f} us/step  {n / us * 1e6:.3e} env-steps/s", flush=True)


rate("fresh process")
lin = torch.nn.Linear(14, 64).to(dev)
with torch.inference_mode():
    y = lin(torch.randn(n, 14, device=dev))
torch.cuda.synchronize()
rate("after one torch Linear (rocBLAS / hipBLASLt initialised)")
with torch.inference_mode():
    rate("inside inference_mode")
x = torch.randn(n, 14, device=dev)
for _ in range(200):
    y = x * 2 + 1
torch.cuda.synchronize()
rate("after elementwise torch ops")
