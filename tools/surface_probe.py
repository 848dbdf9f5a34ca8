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

# the bench's policy section: fused collection launch, then the per-step loop with a torch actor in inference mode
from wheeledlab_amd.core import DriftBatch  # noqa: E402
from wheeledlab_amd.policy import ActorCritic, RolloutStorage  # noqa: E402

b = DriftBatch(n, device=dev, seed=1)
b.reset()
b.observe()
ac, store = ActorCritic(device=dev), RolloutStorage(128, n, device=dev)
b.rollout_policy(ac, store)
torch.cuda.synchronize()
rate("after wl_drift_rollout_policy + wl_mlp_forward")
lin3 = torch.nn.Sequential(torch.nn.Linear(14, 64), torch.nn.ELU(), torch.nn.Linear(64, 64), torch.nn.ELU(),
                           torch.nn.Linear(64, 2)).to(dev)
obs_t = b.obs
with torch.inference_mode():
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(128):
            mu = lin3(obs_t)
            a_t = mu + ac.std * torch.randn_like(mu)
            obs_t, _, _, _ = b.step(a_t)
        torch.cuda.synchronize()
        print(f"torch-actor loop rep {rep}: {(time.perf_counter() - t) / 128 * 1e6:.1f} us/step", flush=True)
rate("after the torch-actor loop")
