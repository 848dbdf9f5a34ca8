"""Workload for rocprofv3 --pmc / --kernel-trace passes: K launches of one task's hot kernel(s) at n envs exactly as bench.py
launches them, nothing else on the GPU.
  pmc_run.py <n> <K> [lanes]                 fused drift step (rollout storage, no int64 `dones` row)  [round-1/2 form]
  pmc_run.py drift|elev|visual|depth <n> <K> [lanes]
elev / visual: K env.step()s (outputs overwritten in place, as bench.py's other_tasks section); depth: K renders of the
depth camera over n elevation-task cars standing on the synthetic terrain (bench.py's depth section)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wheeledlab_amd.core import DepthCamera, DriftBatch, ElevBatch, VisualBatch

argv = sys.argv[1:]
task = "drift"
if argv and not argv[0].isdigit():
    task, argv = argv[0], argv[1:]
n, K = int(argv[0]), int(argv[1])
lanes = int(argv[2]) if len(argv) > 2 else 0
flags = int(os.environ.get("WL_FLAGS", "0"))     # WlEnvBuffers.flags: force an instantiation (round 4: scan forms, streaming)
dev = "cuda:0"
if os.environ.get("WL_LIB"):     # a variant build of the library (tools/build_variants.sh)
    from wheeledlab_amd import _abi as _A
    _A.load(os.environ["WL_LIB"])
if task == "drift":
    env = DriftBatch(n, device=dev, seed=42)
    env.reset()
    if lanes:
        env.set_lanes(lanes)   # force a step-kernel form (WlEnvBuffers.lanes)
    if flags:
        env.set_flags(flags)
    a = torch.rand(K, n, 2, device=dev) * 2 - 1
    obs = torch.zeros(K, n, 14, device=dev)
    rew = torch.zeros(K, n, device=dev)
    term = torch.zeros(K, n, dtype=torch.uint8, device=dev)
    trunc = torch.zeros(K, n, dtype=torch.uint8, device=dev)
    env.rollout(a, obs, rew, term, trunc)
elif task in ("elev", "visual"):
    env = (ElevBatch if task == "elev" else VisualBatch)(n, device=dev, seed=42)
    env.reset()
    if lanes:
        env.set_lanes(lanes)
    if flags:
        env.set_flags(flags)
    if task == "visual":
        env.sample_augmentation(torch.Generator().manual_seed(0))
    a = torch.rand(K, n, 2, device=dev) * 2 - 1
    env.rollout(a)
elif task == "visual_depth":      # round 4: the visual-depth extension task (step on the heightfield + depth image observation)
    from wheeledlab_amd.core import VisualDepthBatch
    env = VisualDepthBatch(n, device=dev, seed=42)
    env.reset()
    a = torch.rand(K, n, 2, device=dev) * 2 - 1
    a[:, :, 0] = a[:, :, 0].abs()
    env.rollout(a)
elif task == "depth":
    env = ElevBatch(n, device=dev, seed=42)
    env.reset()
    a = torch.rand(8, n, 2, device=dev) * 2 - 1
    env.rollout(a)
    cam = DepthCamera(env.hf, dev)
    out = torch.empty(n, 60, 80, device=dev)
    for _ in range(K):
        cam.render(env, 100.0, out)
else:
    raise SystemExit(f"unknown task {task}")
torch.cuda.synchronize()
