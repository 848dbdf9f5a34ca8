"""Visual-task host utilities and mdp terms, numpy.  PARITY PINNED by tests/golden/visual_trav.npz, visual_mdp.npz.
Citations: /root/reference/source/wheeledlab_tasks/wheeledlab_tasks/visual/."""
import numpy as np
from scipy.ndimage import binary_dilation

from .mathlib import F, f32


# ---- traversability map generation (utils/__init__.py:8-147): draws from the GLOBAL numpy RNG like the reference ----

def generate_path(sr, sc, er, ec, m):
    """utils/__init__.py:122-147: monotone lattice path from (sr, sc) to (er, ec), row moves and column moves shuffled"""
    r, c = sr, sc
    m[r, c] = 1
    dr, dc = er - r, ec - c
    moves = [(-1 if dr < 0 else 1, 0)] * abs(dr) + [(0, -1 if dc < 0 else 1)] * abs(dc)
    # the reference permutes a list of strings; the permutation indices depend only on the length
    order = np.random.permutation(len(moves)) if moves else []
    for k in order:
        m[r, c] = 1
        r, c = r + moves[k][0], c + moves[k][1]
        m[r, c] = 1


def generate_env_map(env_size, sub_group_size, num_walkers):
    """utils/__init__.py:95-120"""
    R, C = env_size
    gr, gc = sub_group_size
    m = np.zeros((R, C), bool)
    starts = []
    for i in range(R // gr):
        for j in range(C // gc):
            starts.append((np.random.randint(0, gr) + i * gr, np.random.randint(0, gc) + j * gc))
    for sr, sc in starts:
        for _ in range(num_walkers):
            er, ec = np.random.randint(0, R), np.random.randint(0, C)
            while m[er, ec] == 1:
                er, ec = np.random.randint(0, R), np.random.randint(0, C)
            generate_path(sr, sc, er, ec, m)
    return m


def generate_map(map_size=(500, 500), env_size=(100, 100), sub_group_size=(50, 50), num_walkers=1):
    """generated_colored_plane's hashmap (utils/__init__.py:60,75-86): tiles + asymmetric 3x3 dilation"""
    R, C = map_size
    er, ec = env_size
    m = np.zeros((R, C), bool)
    for i in range(R // er):
        for j in range(C // ec):
            m[i * er:(i + 1) * er, j * ec:(j + 1) * ec] = generate_env_map(env_size, sub_group_size, num_walkers)
    st = np.array([[0, 1, 0], [0, 1, 1], [0, 0, 0]], bool)
    return binary_dilation(m, structure=st, iterations=1)


def generate_random_poses(num_poses, row_spacing, col_spacing, trav):
    """utils/__init__.py:188-202 -> [(x, y, yaw_deg)]"""
    trav = np.asarray(trav)
    H, W = trav.shape
    cand = trav.nonzero()
    idxs = np.random.choice(len(cand[0]), num_poses)
    ys, xs = cand[0][idxs], cand[1][idxs]
    out = []
    for i in range(len(xs)):
        x = (float(xs[i]) - W // 2) * row_spacing
        y = (float(ys[i]) - H // 2) * col_spacing
        out.append((x, y, np.random.uniform(0, 360.0)))
    return out


# ---- traversability lookup (utils/traversability_utils.py:68-88) ----

def get_map_id(x, y, num_rows=500, num_cols=500, row_spacing=0.5, col_spacing=0.5):
    """:83-88 -- float32 arithmetic, `.long()` truncates toward zero, then clamp"""
    x, y = f32(x), f32(y)
    width, height = num_rows * row_spacing, num_cols * col_spacing
    xi = ((x + F(width / 2.0) + F(row_spacing / 2.0)) / F(row_spacing)).astype(np.int64)
    yi = ((y + F(height / 2) + F(col_spacing / 2)) / F(col_spacing)).astype(np.int64)
    return np.clip(xi, 0, num_rows - 1), np.clip(yi, 0, num_cols - 1)


def get_traversability(trav, xy, **kw):
    """:68-79: map[y_idx, x_idx]"""
    xi, yi = get_map_id(f32(xy)[:, 0], f32(xy)[:, 1], **kw)
    return np.asarray(trav)[yi, xi]


# ---- mdp terms (mushr_visual_env_cfg.py) ----

def traversable_reward(trav, pos):
    """:309-312"""
    return np.where(get_traversability(trav, f32(pos)[:, :2]), F(1), F(-1)).astype(F)


def forward_vel(v_b):
    """:370-371"""
    return f32(v_b)[:, 0]


def out_of_map(pos, width=250.0, height=250.0):
    """:390-398"""
    x, y = f32(pos)[:, 0], f32(pos)[:, 1]
    return (x > F(width / 2)) | (x < F(-width / 2)) | (y > F(height / 2)) | (y < F(-height / 2))
