"""Synthetic heightfield terrain for the elevation task.  The reference's terrain mesh (`Terrains/huge_compact.usd`,
elevation/mushr_elevation_env_cfg.py:95-108) is missing from the snapshot; SURVEY.md 8d config 3 prescribes a
synthetic 800 x 800 fp32 grid at 0.05 m (40 x 40 m, ramps + sine hills, seed 0).  Users with a real heightfield pass
their own array to the env / ElevBatch instead."""
from __future__ import annotations

import numpy as np

BASE_Z = 0.19   # root height on the flat base == `plane_init_value` of the reference's height map (:79)


def synthetic_heightfield(n: int = 800, cell: float = 0.05, seed: int = 0):
    """-> (height float32 [n, n] indexed [iy, ix], x0, y0, cell).  Flat base at BASE_Z, gaussian hills, smooth-step
    ramps onto plateaus, gentle undulation; fades to the base at the border; slopes <~ 25 deg."""
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
    return h.astype(np.float32), -half, -half, cell
