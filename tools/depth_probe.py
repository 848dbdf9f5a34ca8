"""us per depth render (wl_visual_depth) over n elevation-task cars on the synthetic terrain.  usage: depth_probe.py [n]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glob
import torch
from wheeledlab_amd import _abi as A
from wheeledlab_amd.core import DepthCamera, ElevBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = sorted(glob.glob(os.path.join(ROOT, "gpurun_variants", "lib_*.so"))) or [None]   # same-box A/B of builds, if any
for path in variants:
  if path:
    A._lib = None
    A.load(path)
  env = ElevBatch(n, device=dev, seed=42)
  env.reset()
  env.rollout(torch.rand(8, n, 2, device=dev) * 2 - 1)
  cam = DepthCamera((env.height, float(env._hf.x0), float(env._hf.y0), float(env._hf.cell)), dev)
  out = torch.empty(n, 60, 80, device=dev)
  res = {"build": os.path.basename(path) if path else "installed", "n": n}
  for md in (100.0, 20.0, 5.0):
      for _ in range(3):
          cam.render(env, md, out)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(10):
          cam.render(env, md, out)
      e1.record()
      torch.cuda.synchronize()
      us = e0.elapsed_time(e1) * 100
      res[f"max_depth_{md:g}"] = {"us": round(us, 1), "Grays_per_s": round(n * 4800 / us / 1e3, 2), "hit": round(float((out < md).float().mean()), 3),
                                  "write_GBs": round(n * 19200 / us / 1e3, 1)}
  print(json.dumps(res))
