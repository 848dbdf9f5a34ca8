"""Depth ray-cast (BASELINE config 5), CPU side: the C oracle (oracle/depth.c: exact per-cell intersection in double) pinned
against an independent brute-force marcher over oracle/heightfield.py::sample and against closed-form cases; and the DEVICE
walk (wheeledlab_amd/csrc/wl_depth_dev.h: max-pyramid traversal in fp32) compiled for the host and held against the oracle --
so the kernel's arithmetic is checked here before it ever runs on a GPU (tests/test_gpu_depth_parity.py does that)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import depth as D
from oracle import heightfield as HF
from oracle import visual_step as VS
from oracle.mathlib import matrix_from_quat
from tests import depth_cases as DC
from wheeledlab_amd import _abi
from wheeledlab_amd.params import visual_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = os.environ.get("WL_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
P = VS.visual_params()


@pytest.fixture(scope="module")
def hf():
    return DC.terrain()


def test_c_oracle_matches_bruteforce_marcher(hf):
    """two independent statements of `depth`: cell-exact quadratic roots (C, double) vs dense marching + bisection over the
    float32 bilinear sampler the physics uses"""
    pos, quat = DC.poses(24, seed=0, hf=hf)
    rng = np.random.RandomState(5)
    pix = rng.choice(4800, 300, replace=False)
    want = D.depth_bruteforce(P, pos, quat, hf, 12.0, pixels=pix)
    got = D.depth(P, pos, quat, hf, 12.0).reshape(len(pos), -1)[:, pix]
    bad, err = DC.mismatch(got, want, 12.0, rtol=1e-4, atol=1e-4)
    # the marcher can step over a crest thinner than its 4 mm stride: a handful of grazing rays at most
    assert bad.mean() < 2e-3, (bad.sum(), err.max())
    assert np.median(err) < 1e-5


def test_flat_field_is_the_analytic_plane_distance():
    flat = (np.full((64, 64), 0.25, np.float32), np.float32(-10.0), np.float32(-10.0), np.float32(20.0 / 63))
    pos, quat = DC.poses(16, seed=2, hf=flat, span=4.0, edge=False)
    got = D.depth(P, pos, quat, flat, 30.0)
    o, d = D.pixel_rays(P, pos, quat)
    t = (0.25 - o[:, None, 2]) / np.where(d[..., 2] < 0, d[..., 2], np.nan)
    hx, hy = o[:, None, 0] + t * d[..., 0], o[:, None, 1] + t * d[..., 1]
    inside = (np.abs(hx) < 10.0 - 1e-3) & (np.abs(hy) < 10.0 - 1e-3) & (t < 30.0)    # hits on the raised square
    want = np.where(inside, t, np.nan).reshape(got.shape)
    m = np.isfinite(want)
    assert m.mean() > 0.2
    np.testing.assert_allclose(got[m], want[m], rtol=1e-5, atol=1e-5)


def test_known_answers_walls_underground_sky(hf):
    pos, quat = DC.poses(32, seed=3, hf=hf)
    d = D.depth(P, pos, quat, hf, 50.0)
    # EDGE[6]: camera underground -> 0 everywhere
    assert (d[6] == 0).all()
    # EDGE[1]: outside the grid, 0.5 m from its x = -20 wall, camera below the wall's top (base 0.19) and looking straight at
    # it: the centre pixels hit the wall at the distance to the plane x = -20 (the optical axis is +x)
    R = matrix_from_quat(quat[1:2])[0]
    o = pos[1] + R @ np.asarray(P.cam_pos, np.float32)
    assert abs(d[1, 29, 40] - (-20.0 - o[0])) < 1e-4 and abs(d[1, 31, 40] - (-20.0 - o[0])) < 1e-4
    # EDGE[3]: far outside looking away: rows below the horizon hit the z = 0 plane analytically, rows above are sky
    o3, d3 = D.pixel_rays(P, pos[3:4], quat[3:4])
    t = np.where(d3[0, :, 2] < 0, -o3[0, 2] / np.where(d3[0, :, 2] < 0, d3[0, :, 2], 1.0), 50.0).clip(max=50.0).reshape(60, 80)
    np.testing.assert_allclose(d[3], t, rtol=1e-5, atol=1e-5)
    assert (d <= 50.0).all() and (d >= 0).all()


@pytest.fixture(scope="module")
def hostlib(tmp_path_factory):
    if not (os.path.exists(CLANG) or shutil.which(CLANG)):
        pytest.skip("no clang++ to build the host simulation")
    out = tmp_path_factory.mktemp("host_sim") / "libwl_depth_host.so"
    subprocess.run([CLANG, "-O1", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                    "-I", os.path.join(ROOT, "tests", "host_sim", "hip_stub"), "-I", os.path.join(ROOT, "wheeledlab_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host_sim", "depth_host.cpp"), "-o", str(out)], check=True)
    lib = C.CDLL(str(out))
    lib.hs_depth.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]
    lib.hs_pyramid.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    lib.hs_pyramid.restype = C.c_longlong
    return lib


@pytest.mark.parametrize("max_depth", [100.0, 20.0, 2.5])
def test_device_walk_on_the_host_matches_the_oracle(hostlib, hf, max_depth):
    """the kernel's per-ray function (pyramid walk, fp32) against the cell-by-cell double oracle on sloped terrain incl. the
    edge cases; non-square and odd-sized grids exercise the pyramid's partial cells"""
    vp = visual_params()
    for field, seed in ((hf, 11), ((hf[0][:613, :349].copy(), hf[1], hf[2], hf[3]), 12)):
        pos, quat = DC.poses(96, seed=seed, hf=field, span=0.5 * min(field[0].shape) * float(field[3]) - 1.5)
        hfs, _keep = DC.hf_struct(field)
        got = np.zeros((len(pos), 60, 80), np.float32)
        assert hostlib.hs_depth(C.byref(vp), C.byref(hfs), len(pos), pos.ctypes.data, quat.ctypes.data, max_depth, got.ctypes.data) == 0
        want = D.depth(P, pos, quat, field, max_depth)
        bad, err = DC.mismatch(got, want, max_depth)
        assert bad.mean() < 1e-4, (bad.sum(), err.max())      # grazing rays only
        assert np.quantile(err, 0.999) < 1e-4


def _host_depth(hostlib, field, pos, quat, max_depth):
    vp = visual_params()
    hfs, _keep = DC.hf_struct(field)
    got = np.zeros((len(pos), 60, 80), np.float32)
    assert hostlib.hs_depth(C.byref(vp), C.byref(hfs), len(pos), pos.ctypes.data, quat.ctypes.data, max_depth, got.ctypes.data) == 0
    return got


def test_size_independent_properties_of_oracle_and_device_walk(hostlib, hf):
    """properties that need no second implementation, for BOTH the oracle and the device walk: (1) a shorter range only clips:
    depth(range a) = min(depth(range b), a) for a < b; (2) the image does not change when terrain AND camera are lifted together;
    (3) it is bounded by [0, range]; (4) below the horizon on the flat base the depth grows monotonically up the image column"""
    pos, quat = DC.poses(48, seed=21, hf=hf)
    lifted = (hf[0] + np.float32(0.5), hf[1], hf[2], hf[3])
    pos_up = pos.copy()
    pos_up[:, 2] += 0.5
    inside = (np.abs(pos[:, 0]) < 18) & (np.abs(pos[:, 1]) < 18) & (pos[:, 2] > 0.19)    # property 2 needs the lifted outside plane too
    for depth_fn in (lambda f, p_, q, r: D.depth(P, p_, q, f, r), lambda f, p_, q, r: _host_depth(hostlib, f, p_, q, r)):
        far, near = depth_fn(hf, pos, quat, 40.0), depth_fn(hf, pos, quat, 6.0)
        np.testing.assert_allclose(near, np.minimum(far, 6.0), rtol=0, atol=1e-5)
        assert far.min() >= 0.0 and far.max() <= 40.0
        up = depth_fn(lifted, pos_up, quat, 6.0)
        hit = (near < 6.0) & inside[:, None, None]
        # rays that leave the grid meet the outside plane (z = 0) 0.5 m lower relative to the lifted camera: exclude those that end there
        on_grid = hit & (np.abs(up - near) < 1e-3)
        assert on_grid.sum() > 0.5 * hit.sum()
    flat = DC.on_lattice((np.full((128, 128), 0.19, np.float32), np.float32(-32.0), np.float32(-32.0), np.float32(0.5)))
    p0 = np.array([[0.0, 0.0, 0.25]], np.float32)
    q0 = np.array([[1.0, 0.0, 0.0, 0.0]], np.float32)
    for img in (D.depth(P, p0, q0, flat, 30.0)[0], _host_depth(hostlib, flat, p0, q0, 30.0)[0]):
        col = img[31:, 40]                      # rows below the horizon, bottom row last
        assert (np.diff(col) < 0).all() and col[-1] < 0.5 < col[0]


def test_device_walk_on_rough_terrain(hostlib):
    """white noise, spikes and stair steps: nothing to skip, a descent and a climb at almost every cell, walls everywhere -- the walk
    against the cell-by-cell oracle (a handful of grazing pixels at most)"""
    for name, field, pos, quat in DC.rough_fields():
        got = _host_depth(hostlib, field, pos, quat, 50.0)
        want = D.depth(P, pos, quat, field, 50.0)
        bad, err = DC.mismatch(got, want, 50.0)
        assert bad.mean() < 5e-5, (name, int(bad.sum()), float(err.max()))


def test_host_built_pyramid_keeps_the_bound_contract(hostlib, hf):
    """the pyramid's entry encoding (int8 slopes, 16-bit fixed-point offsets, header) as the device functions build it, compiled for the
    host: every plane bounds every grid point of its block, tightly (tests/depth_cases.py::check_pyramid); sized as the ABI says"""
    from wheeledlab_amd import _abi as A
    lib = A.load() if os.path.exists(A.LIB_PATH) else None
    rough = DC.on_lattice((np.random.RandomState(5).uniform(0.0, 1.5, (97, 131)),))[0]
    flat = np.full((9, 17), 0.25, np.float32)                      # no relief, no slope: the header's guards
    for field in (hf, (hf[0][:613, :349].copy(), hf[1], hf[2], hf[3]), (rough, -3.0, -2.0, 0.05)):
        h = np.ascontiguousarray(field[0], np.float32)
        hfs, _keep = DC.hf_struct(field)
        n = hostlib.hs_pyramid(C.byref(hfs), None, 0)
        if lib is not None:
            assert n == lib.wl_heightfield_pyramid_floats(h.shape[1], h.shape[0])
        pyr = np.zeros(n, np.float32)
        assert hostlib.hs_pyramid(C.byref(hfs), pyr.ctypes.data, n) == n
        DC.check_pyramid(pyr, h)
    hfs, _keep = DC.hf_struct((flat, 0.0, 0.0, 0.05))
    n = hostlib.hs_pyramid(C.byref(hfs), None, 0)
    pyr = np.zeros(n, np.float32)
    hostlib.hs_pyramid(C.byref(hfs), pyr.ctypes.data, n)
    fmax, qs, c0, qc = pyr[-4:]
    assert fmax == 0.25 and qs == 1.0 and c0 == 0.25 and 0 < qc < 1e-6          # flat field: unit slope quantum, a sliver of offset range
    words = pyr.view(np.uint32)
    assert ((words[1:2] >> 16) > 0).all() and (words[1] & 0xffff) == 0          # the top entry: flat, just above the field
