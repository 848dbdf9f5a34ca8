"""Heightfield terrain for the elevation task and the depth camera: a regular grid of 16-bit height CODES, z = code * z_scale --
the representation of IsaacLab's own height-field terrains (isaaclab.terrains.height_field: int16 x vertical_scale).  The
reference's terrain mesh (`Terrains/huge_compact.usd`, elevation/mushr_elevation_env_cfg.py:95-108) is missing from the snapshot;
SURVEY.md 8d config 3 prescribes a synthetic 800 x 800 grid at 0.05 m (40 x 40 m, ramps + sine hills, seed 0).  Users with a real
heightfield pass their own array to the env / ElevBatch instead: `(height, x0, y0, cell)` with float heights (quantised here,
`quantize_heights`) or `(codes int16, x0, y0, cell, z_scale)` as an IsaacLab generator produced them."""
from __future__ import annotations

import numpy as np

BASE_Z = 0.19   # root height on the flat base == `plane_init_value` of the reference's height map (:79)
Z_SCALE = 2.0 ** -13   # default metres per code: 0.122 mm steps, +-4 m of range (a power of two: codes decode exactly in fp32)


def default_z_scale(max_abs: float) -> float:
    """2^-13 m while +-32767 codes cover the heights, else the smallest power of two that does"""
    zs = Z_SCALE
    while max_abs > 32767 * zs:
        zs *= 2.0
    return zs


def quantize_heights(height, z_scale: float | None = None):
    """float heights [ny, nx] -> (codes int16 [ny, nx], z_scale): code = rint(h / z_scale), clipped to +-32767"""
    h = np.asarray(height, dtype=np.float64)
    if not np.isfinite(h).all():
        raise ValueError("heightfield with non-finite heights")
    hmax = float(np.abs(h).max(initial=0.0))
    zs = float(default_z_scale(hmax) if z_scale is None else z_scale)
    if not (np.isfinite(zs) and zs > 0):
        raise ValueError("z_scale must be positive and finite")
    if hmax > 32767 * zs:
        raise ValueError(f"heights up to {hmax:g} m do not fit 16-bit codes of z_scale {zs:g} m (+-{32767 * zs:g} m)")
    return np.clip(np.rint(h / zs), -32767, 32767).astype(np.int16), zs


def decode_heights(codes, z_scale: float):
    """codes -> float32 heights exactly as every kernel decodes them: (float) code * (float) z_scale, one fp32 multiply"""
    return np.asarray(codes, np.int16).astype(np.float32) * np.float32(z_scale)


def synthetic_heightfield(n: int = 800, cell: float = 0.05, seed: int = 0):
    """-> (height float32 [n, n] indexed [iy, ix], x0, y0, cell).  Flat base at BASE_Z, gaussian hills, smooth-step
    ramps onto plateaus, gentle undulation; fades to the base at the border; slopes <~ 25 deg.  The heights lie ON the code
    lattice of Z_SCALE (every value is code * 2^-13 exactly), so that quantising them is lossless."""
    rng = np.random.RandomState(seed)
    half = 0.5 * n * cell
    xs = np.arange(n) * cell - half
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    h = np.zeros((n, n))
    for _ in range(14):
        cx, cy = rng.uniform(-16, 16, 2)
        s, a = rng.uniform(1.5, 3.5), rng.uniform(0.3, 1.0)
        h += 0.6 * a * np.exp(-((X - cx) ** 2 + (Y - cy) ** 2) / (2 * s * s))
    for _ in range(6):
        cx, cy = rng.uniform(-15, 15, 2)
        w, top = rng.uniform(2.0, 4.0), rng.uniform(0.4, 0.9)
        d = np.maximum(np.abs(X - cx), np.abs(Y - cy))
        t = np.clip((w + 2.5 - d) / 2.5, 0, 1)
        h = np.maximum(h, top * t * t * (3 - 2 * t))
    h += 0.04 * np.sin(0.9 * X) * np.sin(1.1 * Y)
    edge = np.clip((half - np.maximum(np.abs(X), np.abs(Y))) / 1.0, 0, 1)
    h = BASE_Z + np.maximum(h, 0) * edge
    codes, zs = quantize_heights(h, Z_SCALE)
    return decode_heights(codes, zs), -half, -half, cell
