"""Heightfield terrain: bilinear height + normal sampling, and a synthetic terrain generator.
DESIGNED (the reference's terrain mesh `Terrains/huge_compact.usd` is missing, .MISSING_LARGE_BLOBS): SURVEY.md 8d
config 3 prescribes a synthetic 800 x 800 grid at 0.05 m (40 x 40 m, ramps + sine hills, seed 0).
Round 5: the terrain is DEFINED as 16-bit height codes x a vertical scale (IsaacLab's own height-field terrains are int16 x
vertical_scale): `quantize` / `decode` restate the product's representation (wheeledlab_amd/terrain.py), every sampler below
works on the DECODED float32 grid -- exactly the values the kernels decode (one fp32 multiply per grid point)."""
import numpy as np

from .mathlib import F, f32

BASE_Z = 0.19   # root height on the flat base: `plane_init_value` of the reference's height map (elevation cfg :79)
Z_SCALE = 2.0 ** -13   # metres per height code of the synthetic terrain (0.122 mm; +-4 m of range)


def quantize(h, z_scale=Z_SCALE):
    """float heights -> int16 codes: rint(h / z_scale), clipped to +-32767"""
    return np.clip(np.rint(np.asarray(h, np.float64) / float(z_scale)), -32767, 32767).astype(np.int16)


def decode(codes, z_scale=Z_SCALE):
    """int16 codes -> the float32 heights the kernels see: (float) code * (float) z_scale"""
    return np.asarray(codes, np.int16).astype(F) * F(z_scale)


def make_terrain(n=800, cell=0.05, seed=0):
    """-> (height [n, n] float32 indexed [iy, ix], x0, y0, cell): flat base at BASE_Z with smooth hills, ramps and
    plateaus; slopes stay below ~25 deg so the 4WD car can climb them.  Heights on the code lattice: code * 2^-13 exactly."""
    rng = np.random.RandomState(seed)
    half = 0.5 * n * cell
    xs = (np.arange(n) * cell - half).astype(np.float64)
    X, Y = np.meshgrid(xs, xs, indexing="xy")
    h = np.zeros((n, n))
    for _ in range(14):                                   # gaussian hills
        cx, cy = rng.uniform(-16, 16, 2)
        s, a = rng.uniform(1.5, 3.5), rng.uniform(0.3, 1.0)
        h += 0.6 * a * np.exp(-((X - cx) ** 2 + (Y - cy) ** 2) / (2 * s * s))
    for _ in range(6):                                    # ramps up to plateaus (smoothstep)
        cx, cy = rng.uniform(-15, 15, 2)
        w, top = rng.uniform(2.0, 4.0), rng.uniform(0.4, 0.9)
        d = np.maximum(np.abs(X - cx), np.abs(Y - cy))
        t = np.clip((w + 2.5 - d) / 2.5, 0, 1)
        h = np.maximum(h, top * t * t * (3 - 2 * t))
    h += 0.04 * np.sin(0.9 * X) * np.sin(1.1 * Y)          # gentle undulation
    edge = np.clip((half - np.maximum(np.abs(X), np.abs(Y))) / 1.0, 0, 1)   # fade to the flat base at the border
    h = BASE_Z + np.maximum(h, 0) * edge
    return decode(quantize(h)), F(-half), F(-half), F(cell)


def sample(hf, x0, y0, cell, x, y, outside=0.0):
    """bilinear height and unit normal at world (x, y) [N]; outside the grid -> height `outside`, normal +z.
    -> z [N], n [N,3], inside [N] bool"""
    hf = f32(hf)
    ny, nx = hf.shape
    inv = F(1) / F(cell)
    u = (f32(x) - F(x0)) * inv
    v = (f32(y) - F(y0)) * inv
    inside = (u >= 0) & (v >= 0) & (u < nx - 1) & (v < ny - 1)
    uc = np.clip(u, 0, nx - 1 - 1e-3).astype(F)
    vc = np.clip(v, 0, ny - 1 - 1e-3).astype(F)
    i = np.floor(uc).astype(np.int32)
    j = np.floor(vc).astype(np.int32)
    fu, fv = (uc - i).astype(F), (vc - j).astype(F)
    h00, h10 = hf[j, i], hf[j, i + 1]
    h01, h11 = hf[j + 1, i], hf[j + 1, i + 1]
    a = h00 + fu * (h10 - h00)
    b = h01 + fu * (h11 - h01)
    z = a + fv * (b - a)
    dzdx = ((h10 - h00) + fv * ((h11 - h01) - (h10 - h00))) * inv
    dzdy = (b - a) * inv
    nrm = np.stack([-dzdx, -dzdy, np.ones_like(z)], -1)
    nrm = nrm / np.sqrt((nrm * nrm).sum(-1, keepdims=True))
    z = np.where(inside, z, F(outside)).astype(F)
    nrm = np.where(inside[:, None], nrm, f32([0, 0, 1])).astype(F)
    return z, nrm, inside
