"""Camera poses for the depth ray-cast tests (CPU and GPU): cars standing / tilted on the synthetic 800 x 800 terrain of SURVEY
8(d) config 3, plus the edge cases -- cameras outside the grid looking in (side wall of the terrain solid) and out (the z =
outside_z plane), at the border, high above, and underground."""
import numpy as np

from oracle import heightfield as HF
from oracle.mathlib import quat_from_euler_xyz

EDGE = np.array([
    # x, y, z, roll, pitch, yaw
    [-20.5, 3.0, 0.30, 0.0, 0.0, 0.0],        # outside, looking into the grid through the wall of the base (0.19 above z = 0)
    [-20.5, 3.0, -0.05, 0.0, 0.0, 0.0],       # same with the CAMERA (root + 0.18) below the top of the wall
    [19.7, -19.8, 0.50, 0.0, 0.1, -0.7],      # inside near the corner, looking out over the edge
    [25.0, 25.0, 1.00, 0.0, 0.2, 0.3],        # far outside looking away: only the outside plane
    [0.0, -20.02, 0.10, 0.0, 0.0, 1.5708],    # just outside, heading in along +y
    [0.0, 0.0, 30.0, 0.0, 1.3, 0.0],          # high above, looking steeply down
    [3.0, 4.0, -0.5, 0.0, 0.0, 0.0],          # underground: depth 0 everywhere
    [-7.0, 9.0, 0.45, 0.5, -0.4, 2.0],        # rolled over on a slope
    [12.0, -3.0, 0.60, 0.0, -0.6, -2.5],      # nose up (sky in most rows)
    [0.0, 0.0, 0.40, 0.0, 0.0, 0.0],          # axis-aligned
    # the ground track of column 39 / 40 almost parallel to a grid axis (|du| or |dv| ~ 1e-6 cells per metre): one of the walk's two
    # exit parameters is then ~1e6 m away and every step leaves through the other axis
    [3.3, -4.7, 0.45, 0.0, 0.0, 1.5707963 - 0.5 / 39.6304],
    [3.3, -4.7, 0.45, 0.0, 0.0, 1.5707963 + 0.5 / 39.6304],
    [-6.1, 2.2, 0.50, 0.0, 0.0, 0.5 / 39.6304],
    [-6.1, 2.2, 0.50, 0.0, 0.0, 3.1415927 - 0.5 / 39.6304],
], np.float64)


def terrain():
    return HF.make_terrain()


def on_lattice(field, z_scale=None):
    """the same field with its heights on the 16-bit code lattice -- code * z_scale exactly, what the product's quantisation
    (wheeledlab_amd/terrain.py::quantize_heights, its default scale unless one is given) leaves of it: the ORACLE is handed these
    floats, the product re-quantises them without loss, so both sides see one terrain"""
    from wheeledlab_amd.terrain import decode_heights, quantize_heights
    codes, zs = quantize_heights(field[0], z_scale)
    return (decode_heights(codes, zs),) + tuple(field[1:4])


def hf_struct(field, outside_z=0.0, z_scale=None):
    """-> (WlHeightField over the field's height codes and row-pair table, the arrays that must stay alive while the struct is used)"""
    from wheeledlab_amd import _abi
    from wheeledlab_amd.terrain import quantize_heights
    codes, zs = quantize_heights(field[0], z_scale)
    codes = np.ascontiguousarray(codes)
    # the row-pair table (ABI 23), the header's definition in numpy: pair[j][i] = code[j][i] | code[min(j + 1, ny - 1)][i] << 16
    up = np.concatenate([codes[1:], codes[-1:]], 0)
    pairs = np.ascontiguousarray(codes.astype(np.uint16).astype(np.uint32) | (up.astype(np.uint16).astype(np.uint32) << 16))
    return _abi.WlHeightField(codes.ctypes.data, codes.shape[1], codes.shape[0], float(field[1]), float(field[2]), float(field[3]),
                              float(outside_z), zs, pairs.ctypes.data), (codes, pairs)


def poses(n, seed, hf=None, span=17.5, tilt=0.15, edge=True):
    """-> pos [n,3], quat [n,4] float32.  Roots 0.04-0.3 m above the terrain with N(0, tilt) roll / pitch and uniform yaw; the
    first len(EDGE) are the edge cases (when n allows and `edge`)."""
    hf = hf or terrain()
    rng = np.random.RandomState(seed)
    xy = rng.uniform(-span, span, (n, 2)).astype(np.float32)
    z, _, _ = HF.sample(*hf, xy[:, 0], xy[:, 1])
    pos = np.concatenate([xy, (z + rng.uniform(0.04, 0.3, n))[:, None]], 1).astype(np.float32)
    eul = np.stack([rng.normal(0, tilt, n), rng.normal(0, tilt, n), rng.uniform(-np.pi, np.pi, n)], 1)
    if edge and n >= 2 * len(EDGE):
        pos[: len(EDGE)] = EDGE[:, :3]
        eul[: len(EDGE)] = EDGE[:, 3:]
    eul = eul.astype(np.float32)
    quat = quat_from_euler_xyz(eul[:, 0], eul[:, 1], eul[:, 2]).astype(np.float32)
    return pos, np.ascontiguousarray(quat)


def mismatch(got, want, max_depth, rtol=2e-4, atol=2e-4):
    """pixels whose depth differs beyond fp32-vs-double rounding of the same intersection: -> (bad mask, abs error).  A ray that
    grazes a crest resolves to the crest in one arithmetic and to what lies behind it in the other: those are counted, not
    excused silently (the callers bound their number)."""
    err = np.abs(got.astype(np.float64) - want.astype(np.float64))
    return err > atol + rtol * np.abs(want), err


def rough_fields(seed=3):
    """terrains the max-pyramid cannot skip much of -- white noise, isolated spikes, stair steps -- with cameras above, inside and
    on them, steeply tilted: (name, field, pos, quat) tuples"""
    rng = np.random.RandomState(seed)
    out = []
    for name, h, cell in (("white noise", on_lattice((rng.uniform(0, 2, (256, 256)),))[0], 0.1),
                          ("steep noise", on_lattice((np.random.RandomState(5).uniform(0.0, 1.5, (97, 131)),))[0], 0.05),   # check_pyramid's field: up to 30 m / m
                          ("spikes", np.where(rng.rand(300, 200) < 0.02, 3.0, 0.0).astype(np.float32), 0.07),
                          ("steps", (np.floor(np.arange(512)[None, :] / 32) * 0.25 + np.zeros((512, 1))).astype(np.float32), 0.05)):
        ny, nx = h.shape
        field = (h, np.float32(-0.5 * nx * cell), np.float32(-0.5 * ny * cell), np.float32(cell))
        n = 48
        xy = rng.uniform(-0.45 * min(nx, ny) * cell, 0.45 * min(nx, ny) * cell, (n, 2)).astype(np.float32)
        z, _, _ = HF.sample(*field, xy[:, 0], xy[:, 1])
        pos = np.concatenate([xy, (z + rng.uniform(-0.2, 1.0, n))[:, None]], 1).astype(np.float32)
        e = np.stack([rng.normal(0, .3, n), rng.normal(0, .3, n), rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
        quat = np.ascontiguousarray(quat_from_euler_xyz(e[:, 0], e[:, 1], e[:, 2]).astype(np.float32))
        out.append((name, field, pos, quat))
    return out


def check_pyramid(pyr, h, z_scale=2.0 ** -13):
    """the bound pyramid's contract (wheeledlab_amd/csrc/wl_depth_dev.h): entry (J, I) of level L = { uint16 c' | int8 a' | int8 b' } with
    a = a' qs, b = b' qs, c = c0 + c' qc (header behind the heights: field maximum, qs, c0, qc); the plane a (i - I 2^L) + b (j - J 2^L)
    + c lies on or above EVERY grid point of its 2^L x 2^L block of cells (that is all the walk's skips rely on), tightly (the largest
    residual of the block, one offset step and a rounding hair above), and never looser at the block's centre than the block's
    maximum; float 0 = the field's maximum; blocks that cover no cell: the word 0; behind the entries the walk's copy of the 16-bit
    height codes (h = code * z_scale)"""
    h = np.asarray(h, np.float32)
    pyr = np.asarray(pyr, np.float32)
    ny, nx = h.shape
    Pw = 2
    while Pw < nx - 1 or Pw < ny - 1:
        Pw *= 2
    lp = int(np.log2(Pw))
    h0 = max(Pw * Pw // 2, 4)
    nw = (nx * ny + 1) // 2
    assert len(pyr) == h0 + nw + 4 and pyr[0] == h.max()
    codes = pyr[h0: h0 + nw].view(np.int16)[: nx * ny].reshape(ny, nx)
    np.testing.assert_array_equal(codes.astype(np.float32) * np.float32(z_scale), h)      # the walk's copy of the height codes
    fmax, qs, c0, qc = (float(v) for v in pyr[h0 + nw:])
    steepest = max(np.abs(np.diff(h, axis=0)).max(), np.abs(np.diff(h, axis=1)).max())
    assert fmax == h.max() and c0 == h.min() and abs(qs * 127 - steepest) <= 1e-6 * steepest   # one slope quantum per field: its steepest cell edge / 127
    assert abs(qc * 65534 - 2 * (h.max() - h.min())) <= 1e-5 * (h.max() - h.min())                # offsets: 16 bits over twice the relief
    words = pyr.view(np.uint32)
    hd = h.astype(np.float64)
    for L in range(1, lp + 1):
        W, s = Pw >> L, 1 << L
        off = (Pw * Pw) >> (2 * L)
        ent = words[off: off + W * W].reshape(W, W)
        a = (ent >> 8).astype(np.uint8).view(np.int8).astype(np.float64) * qs
        b = ent.astype(np.uint8).view(np.int8).astype(np.float64) * qs
        c = c0 + (ent >> 16).astype(np.float64) * qc
        nJ, nI = min(W, (ny - 1 + s - 1) // s), min(W, (nx - 1 + s - 1) // s)
        assert (ent[nJ:] == 0).all() and (ent[:, nI:] == 0).all() and (ent[:nJ, :nI] >> 16 > 0).all()   # blocks that cover no cell
        step = max(1, (nJ * nI) // 4000)                                            # every block of the coarse levels, a sample of the fine
        for k in range(0, nJ * nI, step):
            J, I = divmod(k, nI)
            blk = hd[J * s: min((J + 1) * s, ny - 1) + 1, I * s: min((I + 1) * s, nx - 1) + 1]
            jj, ii = np.mgrid[0:blk.shape[0], 0:blk.shape[1]]
            slack = c[J, I] + a[J, I] * ii + b[J, I] * jj - blk
            one_step = 1.01 * qc                                                 # the offset is rounded UP to a whole step
            assert slack.min() >= 0.0, (L, J, I, slack.min())
            # tight: it (nearly) touches a point -- up to the rounding allowance of the walk's travelled extent (plane_entry)
            travel = 4.1e-7 * (abs(a[J, I]) * (blk.shape[1] - 1) + abs(b[J, I]) * (blk.shape[0] - 1))
            assert slack.min() < one_step + travel + 1e-5 * (1 + abs(c[J, I])), (L, J, I, slack.min())
            centre = c[J, I] + 0.5 * (a[J, I] * (blk.shape[1] - 1) + b[J, I] * (blk.shape[0] - 1))
            assert centre <= blk.max() + one_step + travel + 1e-5 * (1 + abs(blk.max())), (L, J, I)
