"""Depth ray-cast against a heightfield (BASELINE.json config 5) -- TEST INFRASTRUCTURE.

`depth()` is the ctypes face of oracle/depth.c (exact per-cell ray / bilinear-patch intersection in double precision, grid
walk cell by cell; see its header for what it restates: wheeledlab_tasks/visual/mdp_sensors/observations.py:89-95 forwards
IsaacLab's `distance_to_image_plane`, camera cfg visual/mushr_visual_env_cfg.py:230-246).  `depth_bruteforce()` is an
independent, slow statement of the same definition -- dense marching over oracle/heightfield.py::sample plus bisection --
that pins the C code on small cases (tests/test_oracle_depth.py).  Parity unpinned against IsaacLab / RTX (absent)."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import heightfield as HF
from .mathlib import F, f32, matrix_from_quat

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libwl_oracle.so")
_lib = None

IMG_H, IMG_W = 60, 80


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "depth.c")
        if not os.path.exists(_LIB) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB)):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _lib = C.CDLL(_LIB)
        _lib.wl_oracle_depth.restype = C.c_int
        _lib.wl_oracle_depth.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_float] * 4 + [C.c_int] * 2 + [C.c_void_p] + [C.c_int] * 2 + \
            [C.c_float] * 5 + [C.c_void_p]
    return _lib


def depth(p, pos, quat, hf, max_depth, outside_z=0.0, img_h=IMG_H, img_w=IMG_W):
    """pos [n,3], quat [n,4] (w,x,y,z) of the root; p: visual params (cam_pos, fx, fy, cx, cy); hf = (height [ny,nx], x0, y0,
    cell) -> distance_to_image_plane [n, img_h, img_w] float32, clipped at max_depth"""
    h, x0, y0, cell = hf
    h = np.ascontiguousarray(h, dtype=np.float32)
    pos = np.ascontiguousarray(pos, dtype=np.float32)
    quat = np.ascontiguousarray(quat, dtype=np.float32)
    cam = np.ascontiguousarray(p.cam_pos, dtype=np.float32)
    n = pos.shape[0]
    out = np.empty((n, img_h, img_w), np.float32)
    rc = _load().wl_oracle_depth(n, pos.ctypes.data, quat.ctypes.data, cam.ctypes.data, float(p.fx), float(p.fy), float(p.cx),
                                 float(p.cy), img_h, img_w, h.ctypes.data, h.shape[1], h.shape[0], float(x0), float(y0), float(cell),
                                 float(outside_z), float(max_depth), out.ctypes.data)
    assert rc == 0
    return out


def pixel_rays(p, pos, quat, img_h=IMG_H, img_w=IMG_W):
    """camera origin [n,3] and world ray directions [n, img_h*img_w, 3] (float64) whose body-frame x component is 1"""
    R = matrix_from_quat(f32(quat)).astype(np.float64)
    o = f32(pos).astype(np.float64) + np.einsum("nij,j->ni", R, np.asarray(p.cam_pos, np.float64))
    rows, cols = np.arange(img_h, dtype=np.float64), np.arange(img_w, dtype=np.float64)
    by = -((cols + 0.5 - p.cx) / p.fx)
    bz = -((rows + 0.5 - p.cy) / p.fy)
    db = np.stack([np.ones((img_h, img_w)), np.broadcast_to(by[None, :], (img_h, img_w)),
                   np.broadcast_to(bz[:, None], (img_h, img_w))], -1).reshape(-1, 3)
    return o, np.einsum("nij,pj->npi", R, db)


def depth_bruteforce(p, pos, quat, hf, max_depth, outside_z=0.0, dt=0.004, img_h=IMG_H, img_w=IMG_W, pixels=None):
    """Independent slow definition: march every ray in steps of `dt` over heightfield.sample (float32 bilinear), the first
    sample at or below the surface is refined by 30 bisections.  pixels: optional flat pixel indices (default all)."""
    h, x0, y0, cell = hf
    o, d = pixel_rays(p, pos, quat, img_h, img_w)
    if pixels is not None:
        d = d[:, pixels]
    n, m = d.shape[:2]
    oo = np.repeat(o[:, None, :], m, 1).reshape(-1, 3)
    dd = d.reshape(-1, 3)

    def above(t):
        q = oo + t[:, None] * dd
        z, _, _ = HF.sample(h, x0, y0, cell, q[:, 0].astype(F), q[:, 1].astype(F), outside=outside_z)
        return q[:, 2] - z.astype(np.float64)

    res = np.full(n * m, float(max_depth))
    live = np.arange(n * m)
    t = np.zeros(n * m)
    g0 = above(t)
    hit0 = g0 <= 0
    res[hit0] = 0.0
    live = live[~hit0]
    k = 0
    while len(live) and k * dt < max_depth:
        k += 1
        tk = np.full(len(live), min(k * dt, float(max_depth)))
        q = oo[live] + tk[:, None] * dd[live]
        z, _, _ = HF.sample(h, x0, y0, cell, q[:, 0].astype(F), q[:, 1].astype(F), outside=outside_z)
        below = q[:, 2] - z <= 0
        if below.any():
            idx = live[below]
            lo, hi = np.full(len(idx), (k - 1) * dt), tk[below].copy()
            for _ in range(30):
                mid = 0.5 * (lo + hi)
                qm = oo[idx] + mid[:, None] * dd[idx]
                zm, _, _ = HF.sample(h, x0, y0, cell, qm[:, 0].astype(F), qm[:, 1].astype(F), outside=outside_z)
                b = qm[:, 2] - zm <= 0
                hi = np.where(b, mid, hi)
                lo = np.where(b, lo, mid)
            res[idx] = hi
            live = live[~below]
    return res.reshape(n, m).astype(np.float32)
