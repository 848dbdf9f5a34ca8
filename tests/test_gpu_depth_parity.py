"""Depth ray-cast against a heightfield on the GPU (BASELINE.json config 5; reference hook
wheeledlab_tasks/visual/mdp_sensors/observations.py:89-95): wl_visual_depth through the C ABI against oracle/depth.c (exact
per-cell intersection in double precision, walking cell by cell -- pinned on the CPU side by tests/test_oracle_depth.py).
Tolerance: |d_gpu - d_oracle| <= 2e-4 + 2e-4 d (fp32 walk vs double); rays that graze a crest may resolve to the crest or to
what lies behind it -- those pixels are COUNTED and bounded (< 1e-4 of the image), never excused silently."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import depth as D
from oracle import visual_step as VS
from tests import depth_cases as DC

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
P = VS.visual_params()


@pytest.fixture(scope="module")
def hf():
    return DC.terrain()


def _posed_batch(pos, quat):
    """a VisualBatch whose state rows carry the given root poses (the depth entry reads rows WL_S_PX.. / WL_S_QW.. only)"""
    from wheeledlab_amd.core import VisualBatch
    trav = np.ones((500, 500), bool)
    env = VisualBatch(len(pos), device=DEV, seed=1, trav_map=trav)
    env.state[0:3, : env.n] = torch.from_numpy(np.ascontiguousarray(pos.T)).to(DEV)
    env.state[3:7, : env.n] = torch.from_numpy(np.ascontiguousarray(quat.T)).to(DEV)
    return env


def test_pyramid_planes_bound_every_grid_point_of_their_cells(hf):
    """the bound pyramid as the device builds it (wl_heightfield_build_pyramid) keeps its contract: tests/depth_cases.py::check_pyramid"""
    from wheeledlab_amd.core import DepthCamera
    rough = DC.on_lattice((np.random.RandomState(5).uniform(0.0, 1.5, (97, 131)), -3.0, -2.0, 0.05))       # nothing smooth about it
    coarse = DC.on_lattice((np.random.RandomState(6).uniform(-0.5, 1.0, (40, 56)), -1.0, -1.4, 0.05), z_scale=0.005)   # IsaacLab's default vertical_scale
    for field, zs in ((hf, None), ((hf[0][:613, :349].copy(), hf[1], hf[2], hf[3]), None), (rough, None), (coarse, 0.005)):
        cam = DepthCamera(field if zs is None else field + (zs,), DEV)
        torch.cuda.synchronize()
        assert cam.hf.codes.dtype == torch.int16 and torch.equal(cam.height.cpu(), torch.from_numpy(np.asarray(field[0], np.float32)))   # lossless
        DC.check_pyramid(cam.pyramid.cpu().numpy(), field[0], cam.hf.z_scale)


@pytest.mark.parametrize("max_depth", [100.0, 20.0])
def test_full_images_match_the_oracle_at_4096_envs(hf, max_depth):
    """BASELINE config 5 at full size: 4096 cameras on the sloped synthetic terrain, every pixel of every 60 x 80 image"""
    from wheeledlab_amd.core import DepthCamera
    n = 4096
    pos, quat = DC.poses(n, seed=7, hf=hf)
    env = _posed_batch(pos, quat)
    cam = DepthCamera(hf, DEV)
    got = cam.render(env, max_depth)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    want = D.depth(P, pos, quat, hf, max_depth)
    assert got.shape == want.shape == (n, 60, 80)
    bad, err = DC.mismatch(got, want, max_depth)
    hit = want < max_depth
    print(f"depth parity n={n} max_depth={max_depth}: grazing pixels {int(bad.sum())} of {bad.size} ({bad.mean():.2e}), "
          f"hit fraction {hit.mean():.3f}, median err {np.median(err):.2e}, p99.99 err {np.quantile(err, 0.9999):.2e}")
    assert bad.mean() < 1e-4, (int(bad.sum()), float(err.max()))
    assert np.quantile(err, 0.999) < 1e-4
    assert 0.3 < hit.mean() < 0.9          # both outcomes well represented
    # the same launch through the batch's own entry (cached camera) and into a caller's buffer
    out = torch.empty(n, 60, 80, device=DEV)
    env.depth(hf, max_depth, out=out)
    assert torch.equal(out.cpu(), torch.from_numpy(got))


def test_odd_grid_edge_cases_and_short_range(hf):
    """a 349 x 613 grid (partial pyramid cells on both axes), cameras incl. all edge cases, max_depth shorter than most hits"""
    from wheeledlab_amd.core import DepthCamera
    field = (hf[0][:613, :349].copy(), hf[1], hf[2], hf[3])
    span = 0.5 * 349 * float(hf[3]) - 1.5
    pos, quat = DC.poses(300, seed=9, hf=field, span=span)
    env = _posed_batch(pos, quat)
    cam = DepthCamera(field, DEV)
    for md in (2.5, 60.0):
        got = cam.render(env, md).cpu().numpy()
        want = D.depth(P, pos, quat, field, md)
        bad, err = DC.mismatch(got, want, md)
        assert bad.mean() < 1e-4, (md, int(bad.sum()), float(err.max()))
    assert (got[6] == 0).all()             # EDGE[6]: underground


def test_depth_of_driving_elevation_cars(hf):
    """the poses the elevation task's physics produces (cars settled on slopes after 30 steps), rendered from the
    ElevBatch's own state rows"""
    from wheeledlab_amd.core import DepthCamera, ElevBatch
    n = 512
    env = ElevBatch(n, device=DEV, seed=4)
    env.reset()
    g = torch.Generator(device=DEV).manual_seed(0)
    for _ in range(30):
        env.step(torch.rand(n, 2, device=DEV, generator=g) * 2 - 1)
    cam = DepthCamera(env.hf, DEV)
    got = cam.render(env, 30.0)
    torch.cuda.synchronize()
    st = env.state[:, :n].cpu().numpy()
    field = (env.height.cpu().numpy(), float(env._hf.x0), float(env._hf.y0), float(env._hf.cell))
    want = D.depth(P, st[0:3].T, st[3:7].T, field, 30.0)
    bad, err = DC.mismatch(got.cpu().numpy(), want, 30.0)
    assert bad.mean() < 1e-4, (int(bad.sum()), float(err.max()))


def test_invalid_arguments_are_refused(hf):
    from wheeledlab_amd import _abi as A
    from wheeledlab_amd.core import DepthCamera
    pos, quat = DC.poses(64, seed=1, hf=hf, edge=False)
    env = _posed_batch(pos, quat)
    cam = DepthCamera(hf, DEV)
    out = torch.empty(64, 60, 80, device=DEV)
    lib = env.lib
    args = lambda **kw: (C.byref(kw.get("p", env.p)), C.byref(kw.get("b", env._bufs)), C.byref(kw.get("hf", cam._hf)),
                         kw.get("pyr", cam.pyramid.data_ptr()), kw.get("md", 10.0), kw.get("out", out.data_ptr()), None)
    assert lib.wl_visual_depth(*args()) == 0
    assert lib.wl_visual_depth(*args(pyr=None)) == -1
    assert lib.wl_visual_depth(*args(out=None)) == -1
    assert lib.wl_visual_depth(*args(md=0.0)) == -1
    bad_hf = A.WlHeightField(cam._hf.height, 1, 800, 0.0, 0.0, 0.05, 0.0, cam._hf.z_scale)
    assert lib.wl_visual_depth(*args(hf=bad_hf)) == -1
    assert lib.wl_heightfield_pyramid_floats(1, 5) == 0 and lib.wl_heightfield_pyramid_floats(800, 800) == 1024 * 1024 // 2 + 800 * 800 // 2 + 4
    assert lib.wl_heightfield_build_pyramid(C.byref(bad_hf), cam.pyramid.data_ptr(), None) == -1
    for zs in (float("inf"), float("nan"), 0.0):      # a vertical scale that is not a positive finite number is refused at the ABI
        inf_hf = A.WlHeightField(cam._hf.height, 800, 800, -20.0, -20.0, 0.05, 0.0, zs)
        assert lib.wl_visual_depth(*args(hf=inf_hf)) == -1
        assert lib.wl_heightfield_build_pyramid(C.byref(inf_hf), cam.pyramid.data_ptr(), None) == -1
    torch.cuda.synchronize()


def test_the_scene_camera_cache_tells_vertical_scales_apart():
    """the same int16 codes tensor handed over with another z_scale is another terrain -- the cached DepthCamera
    (its pyramid is built from decoded heights) must not be reused"""
    from wheeledlab_amd.core import VisualBatch, _cached_depth_camera
    codes = torch.randint(-200, 200, (128, 128), dtype=torch.int16, device=DEV)
    env = VisualBatch(64, device=DEV, seed=1, trav_map=np.ones((500, 500), bool))
    a = _cached_depth_camera(env, (codes, -3.2, -3.2, 0.05, 2.0 ** -13))
    assert _cached_depth_camera(env, (codes, -3.2, -3.2, 0.05, 2.0 ** -13)) is a
    b = _cached_depth_camera(env, (codes, -3.2, -3.2, 0.05, 2.0 ** -11))
    assert b is not a and b._hf.z_scale == 2.0 ** -11 and float((b.height - 4 * a.height).abs().max()) == 0.0


def test_rough_terrain(hf):
    """white noise, isolated spikes, stair steps (tests/depth_cases.py::rough_fields): the pyramid skips little, the walk descends
    and climbs at almost every cell, cameras sit inside the terrain -- parity with the oracle and, above all, termination"""
    from wheeledlab_amd.core import DepthCamera
    for name, field, pos, quat in DC.rough_fields():
        env = _posed_batch(pos, quat)
        got = DepthCamera(field, DEV).render(env, 50.0)
        torch.cuda.synchronize()
        want = D.depth(P, pos, quat, field, 50.0)
        bad, err = DC.mismatch(got.cpu().numpy(), want, 50.0)
        assert bad.mean() < 5e-5, (name, int(bad.sum()), float(err.max()))
