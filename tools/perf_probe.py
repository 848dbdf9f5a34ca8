"""Timing probes of the fused drift kernel: us/launch vs decimation (physics slope vs fixed part) and vs n_envs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd.core import DriftBatch

dev = "cuda:0"


def time_rollout(env, actions, reps=6):
    env.rollout(actions)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        env.rollout(actions)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * actions.shape[0])


out = {}
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4096,1048576").split(",")]:
    env = DriftBatch(n, device=dev, seed=42)
    env.reset()
    K = 128 if n <= 65536 else 8
    a = torch.rand(K, n, 2, device=dev) * 2 - 1
    for dec in (0, 1, 2, 4, 8):
        env.p.decimation = dec if dec > 0 else 1
        if dec == 0:
            env.p.vehicle.substeps = 1
            env.p.sim_dt = 0.005
        # decimation 0 is emulated by p.decimation=1 with... (not expressible) -> skip
        if dec == 0:
            continue
        out[f"n{n}_dec{dec}"] = round(time_rollout(env, a), 2)
    env.p.decimation = 4
    for k, name in (("enable_corruption", "nonoise"), ("enable_pushes", "nopush"), ("log_episode_sums", "noepsum")):
        setattr(env.p, k, 0)
        out[f"n{n}_dec4_{name}"] = round(time_rollout(env, a), 2)
        setattr(env.p, k, 1)
    del env, a
print(json.dumps(out))
