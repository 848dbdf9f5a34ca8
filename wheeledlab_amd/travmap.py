"""Traversability map of the visual task: procedurally generated black/white plane (reference:
wheeledlab_tasks/visual/utils/__init__.py:8-147).  The reference generates it at config-import time from the GLOBAL
numpy RNG and writes a USD mesh; here the map is generated at env construction and stays a byte
grid.  Same draws in the same order => the same map as the reference for a given numpy seed (pinned by
tests/test_host_surface_cpu.py against the golden map)."""
from __future__ import annotations

import numpy as np
from scipy.ndimage import binary_dilation

_DILATE = np.array([[0, 1, 0], [0, 1, 1], [0, 0, 0]], dtype=bool)   # asymmetric L1 structure (:85)


def _random_walk(grid, start, end, rng):
    """monotone lattice path start -> end: |drow| row moves and |dcol| column moves in a random order (:122-147)"""
    (r, c), (er, ec) = start, end
    n_r, n_c = abs(er - r), abs(ec - c)
    step_r, step_c = (1 if er >= r else -1), (1 if ec >= c else -1)
    order = rng.permutation(n_r + n_c)          # index < n_r -> a row move, else a column move
    grid[r, c] = True
    for k in order:
        if k < n_r:
            r += step_r
        else:
            c += step_c
        grid[r, c] = True


def _tile(env_size, group_size, walkers, rng):
    """one env tile: a start point per sub-group, each walking to a random not-yet-traversable end point (:95-120)"""
    rows, cols = env_size
    g_r, g_c = group_size
    grid = np.zeros((rows, cols), dtype=bool)
    starts = [(rng.randint(0, g_r) + i * g_r, rng.randint(0, g_c) + j * g_c)
              for i in range(rows // g_r) for j in range(cols // g_c)]
    for start in starts:
        for _ in range(walkers):
            end = (rng.randint(0, rows), rng.randint(0, cols))
            while grid[end]:
                end = (rng.randint(0, rows), rng.randint(0, cols))
            _random_walk(grid, start, end, rng)
    return grid


def generate_traversability_map(map_size=(500, 500), env_size=(100, 100), sub_group_size=(50, 50), num_walkers=1, rng=None):
    """-> bool [rows, cols]; `rng` is a numpy RandomState-like (default: the global numpy RNG, as the reference)"""
    rng = np.random if rng is None else rng
    rows, cols = map_size
    if rows % env_size[0] or cols % env_size[1]:
        raise ValueError("Map size must be a multiple of the sub environment size.")
    grid = np.zeros((rows, cols), dtype=bool)
    for i in range(rows // env_size[0]):
        for j in range(cols // env_size[1]):
            grid[i * env_size[0]:(i + 1) * env_size[0], j * env_size[1]:(j + 1) * env_size[1]] = \
                _tile(env_size, sub_group_size, num_walkers, rng)
    return binary_dilation(grid, structure=_DILATE, iterations=1)


def spawn_cells(trav) -> np.ndarray:
    """traversable cells (iy, ix) in nonzero order: the spawn candidates (:192-195)"""
    ys, xs = np.asarray(trav).nonzero()
    return np.stack([ys, xs], -1).astype(np.int32)
