"""Developer probe: walk statistics of the depth ray-cast's device walk on the host (tools/depth_walk_stats.cpp) -- steps per ray,
wave-steps per 4 x 16 tile, lanes busy, share of wave-steps with a lane in a fine cell.  No GPU needed.
usage: depth_walk_stats.py [poses.npy [max_depth]]     poses.npy: [n, 7] root pos + quat (e.g. dumped from a bench run); default:
tests/depth_cases.py::poses(256, seed 7) on the synthetic terrain.  Extra compiler flags through WL_DWS_FLAGS (e.g. -DWL_DEPTH_START_LEVEL=3)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import depth_cases as DC             # noqa: E402
from wheeledlab_amd import _abi                 # noqa: E402

so = "/tmp/wl_depth_walk_stats.so"
subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-DWL_HOST_SIM",
                *os.environ.get("WL_DWS_FLAGS", "").split(), "-I", os.path.join(ROOT, "tests", "host_sim", "hip_stub"),
                "-I", os.path.join(ROOT, "wheeledlab_amd", "csrc"), os.path.join(ROOT, "tools", "depth_walk_stats.cpp"), "-o", so], check=True)
lib = C.CDLL(so)
lib.dws_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
hf = DC.terrain()
if len(sys.argv) > 1:
    st = np.load(sys.argv[1]).astype(np.float32)
    pos, quat = np.ascontiguousarray(st[:, :3]), np.ascontiguousarray(st[:, 3:7])
else:
    pos, quat = DC.poses(256, seed=7, hf=hf)
far = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
from wheeledlab_amd.params import visual_params as product_params   # noqa: E402
vp = product_params()          # the ctypes struct the kernels take (camera intrinsics and mounting pose)
hfs, keep = DC.hf_struct(hf)
out = (C.c_double * 6)()
lib.dws_stats(C.byref(vp), C.byref(hfs), len(pos), pos.ctypes.data, quat.ctypes.data, far, out)
rays, ray_steps, tiles, wave_steps, fine_ws, fine_rs = list(out)
print(f"images {len(pos)}  far {far:g} m   steps/ray {ray_steps / rays:.2f}   wave-steps/tile {wave_steps / tiles:.2f}   "
      f"lanes busy {ray_steps / (64 * wave_steps):.3f}   wave-steps with a fine lane {fine_ws / wave_steps:.3f}   "
      f"fine ray-steps {fine_rs / ray_steps:.3f} of ray-steps, lanes active in them {fine_rs / (64 * max(fine_ws, 1)):.3f}")
sp = (C.c_double * 2)()
lib.dws_shared_prefix.argtypes = lib.dws_stats.argtypes
lib.dws_shared_prefix(C.byref(vp), C.byref(hfs), len(pos), pos.ctypes.data, quat.ctypes.data, far, sp)
print(f"leading clear steps shared by ALL rays of a tile (what a tile-cooperative start could skip at most): {sp[0] / tiles:.2f} of {sp[1] / tiles:.2f} "
      f"wave-steps per tile ({100 * sp[0] / sp[1]:.1f} %)")
