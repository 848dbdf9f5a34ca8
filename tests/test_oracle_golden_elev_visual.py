"""Pin the elevation / visual oracle terms to the golden vectors produced by the reference's own functions."""
import numpy as np

from oracle import elev_mdp as E


def test_elevation_terms_match_reference(golden):
    g = golden("elevation_mdp")
    tol = dict(rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(E.world_height_map(g["sensor_pos_w"][:, 2], g["ray_hits_z"], g["pos"][:, 2]),
                               g["world_height_map"], rtol=1e-6, atol=4e-6)   # (z+20) - hit: fp32 cancellation at ~20
    np.testing.assert_allclose(E.goal_relative_xyz(g["pos"], g["command"]), g["goal_relative_xyz"], **tol)
    np.testing.assert_allclose(E.goal_progress_rate(g["pos"], g["lin_vel_w"], g["command"]), g["goal_progress_rate"],
                               rtol=1e-5, atol=1e-5, equal_nan=True)
    np.testing.assert_allclose(E.higher_elevation(g["pos"], g["lin_vel_b"]), g["higher_elevation"], **tol)
    np.testing.assert_array_equal(E.is_falling_penalty(g["lin_vel_b"]), g["is_falling_penalty"])
    np.testing.assert_allclose(E.forward_vel(g["lin_vel_b"]), g["forward_vel"], **tol)
    np.testing.assert_array_equal(E.stuck(g["lin_vel_b"], g["joint_vel"][:, 2:6]), g["stuck"])
    np.testing.assert_allclose(E.upright_penalty(g["quat"], 60.0), g["upright_penalty"], rtol=1e-4, atol=2e-3)  # acos, degrees
    ok = np.abs(g["upright_penalty"]) > 1e-2                                   # away from the 60 deg tie
    np.testing.assert_array_equal(E.upright_bool(g["quat"])[ok | (g["upright_penalty"] == 0)],
                                  g["upright_bool"][ok | (g["upright_penalty"] == 0)])
    np.testing.assert_array_equal(E.close_to_goal(g["pos"], g["command"]), g["close_to_goal"])
    np.testing.assert_array_equal(g["weights"], np.array([200.0, 5000.0, 0.0, -200.0], np.float32))
    assert g["close_to_goal"].sum() >= 4 and g["stuck"].sum() >= 1 and g["upright_bool"].sum() >= 10   # branches are exercised


def test_visual_map_generation_and_lookup_match_reference(golden):
    from oracle import visual_mdp as V
    g = golden("visual_trav")
    np.random.seed(0)
    np.testing.assert_array_equal(V.generate_env_map((20, 20), (10, 10), 1), g["env_map_20"])
    np.random.seed(0)
    full = V.generate_map()
    want = np.unpackbits(g["full_map_packed"])[: 500 * 500].reshape(500, 500).astype(bool)
    np.testing.assert_array_equal(full, want)                      # same numpy seed -> same 500 x 500 map, bit for bit
    np.random.seed(1)
    poses = np.array(V.generate_random_poses(64, 0.5, 0.5, full))
    np.testing.assert_allclose(poses, g["poses"], rtol=0, atol=1e-12)
    xi, yi = V.get_map_id(g["xy"][:, 0], g["xy"][:, 1])
    np.testing.assert_array_equal(xi, g["x_idx"])
    np.testing.assert_array_equal(yi, g["y_idx"])
    np.testing.assert_array_equal(V.get_traversability(full, g["xy"]), g["trav"])
    assert want[yi[:64], xi[:64]].all()                             # spawn poses sit on traversable cells


def test_visual_terms_match_reference(golden):
    from oracle import visual_mdp as V
    g = golden("visual_mdp")
    full = np.unpackbits(golden("visual_trav")["full_map_packed"])[: 500 * 500].reshape(500, 500).astype(bool)
    np.testing.assert_array_equal(V.traversable_reward(full, g["pos"]), g["traversable_reward"])
    np.testing.assert_array_equal(V.forward_vel(g["lin_vel_b"]), g["forward_vel"])
    np.testing.assert_array_equal(V.out_of_map(g["pos"]), g["out_of_map"])
    np.testing.assert_array_equal(g["weights"], np.array([5.0, 7.0], np.float32))


def test_camera_blur_matches_scipy_correlation():
    """the GaussianBlur(5, sigma) restated from torchvision's published definition (torchvision is not installed; kernel
    exp(-x^2 / 2 sigma^2) on x = -2..2, normalised; separable; reflect padding that does not repeat the edge) against an
    independent implementation: scipy.ndimage.correlate1d in 'mirror' mode, on rendered camera images"""
    import scipy.ndimage as ndi

    from oracle import visual_step as VS
    rng = np.random.RandomState(3)
    trav = rng.rand(500, 500) < 0.5
    n = 6
    p = VS.visual_params()
    st = VS.init_state(p, n)
    st[VS.PX, :n], st[VS.PX + 1, :n], st[VS.PX + 2, :n] = rng.uniform(-50, 50, n), rng.uniform(-50, 50, n), 0.1
    yaw = rng.uniform(0, 6.28, n)
    st[VS.QW, :n], st[VS.QW + 3, :n] = np.cos(yaw / 2), np.sin(yaw / 2)
    p.brightness = 0.9
    plain = VS.camera(p, st[:, :n], trav)
    img = ((plain * 0.5 + 0.5) / 0.9999).reshape(n, VS.IMG_H - VS.CROP, VS.IMG_W).astype(np.float64)
    assert 0.2 < img.mean() < 0.8 and img.std() > 0.2          # a real image: road / off-road / sky
    for sigma in (0.1, 0.7, 2.0, 5.0):
        p.blur_sigma = sigma
        got = VS.camera(p, st[:, :n], trav).reshape(n, VS.IMG_H - VS.CROP, VS.IMG_W)
        x = np.arange(-2, 3, dtype=np.float64)
        k = np.exp(-0.5 * (x / sigma) ** 2)
        k /= k.sum()
        want = ndi.correlate1d(ndi.correlate1d(img, k, axis=2, mode="mirror"), k, axis=1, mode="mirror")
        np.testing.assert_allclose(got, (want * 0.9999 - 0.5) / 0.5, atol=3e-6)


def test_heightfield_sampling_matches_scipy_interpolation():
    """the bilinear terrain sampler (designed: the reference's terrain mesh is missing) against scipy's independent
    RegularGridInterpolator on the synthetic terrain: heights to fp32 rounding, normals against central differences of the
    interpolant, outside-the-grid behaviour"""
    from scipy.interpolate import RegularGridInterpolator

    from oracle import heightfield as HF
    hf, x0, y0, cell = HF.make_terrain()
    n = hf.shape[0]
    axis = np.float64(x0) + np.arange(n) * np.float64(cell)
    interp = RegularGridInterpolator((axis, axis), hf.astype(np.float64), method="linear")      # indexed [iy, ix]
    rng = np.random.RandomState(4)
    x, y = rng.uniform(-19.9, 19.8, 5000), rng.uniform(-19.9, 19.8, 5000)
    z, nrm, inside = HF.sample(hf, x0, y0, cell, x, y)
    assert inside.all()
    np.testing.assert_allclose(z, interp(np.stack([y, x], -1)), atol=5e-5)
    # the normal is (-dz/dx, -dz/dy, 1) normalised: compare the slopes with differences taken inside one cell
    u = (x - np.float64(x0)) / np.float64(cell)
    v = (y - np.float64(y0)) / np.float64(cell)
    mid = (np.abs(u - np.floor(u) - 0.5) < 0.3) & (np.abs(v - np.floor(v) - 0.5) < 0.3)
    eps = 0.1 * np.float64(cell)
    dzdx = (interp(np.stack([y, x + eps], -1)) - interp(np.stack([y, x - eps], -1))) / (2 * eps)
    dzdy = (interp(np.stack([y + eps, x], -1)) - interp(np.stack([y - eps, x], -1))) / (2 * eps)
    np.testing.assert_allclose((-nrm[:, 0] / nrm[:, 2])[mid], dzdx[mid], atol=2e-3)
    np.testing.assert_allclose((-nrm[:, 1] / nrm[:, 2])[mid], dzdy[mid], atol=2e-3)
    np.testing.assert_allclose(np.linalg.norm(nrm, axis=1), 1.0, atol=1e-6)
    zo, no, ins = HF.sample(hf, x0, y0, cell, np.array([25.0, -21.0, 0.0]), np.array([0.0, 0.0, 30.0]), outside=0.19)
    assert not ins.any() and np.allclose(zo, 0.19) and np.allclose(no, [0, 0, 1])


def test_saturation_and_hue_on_a_grey_image():
    """ColorJitter(saturation=.8, hue=.5) of the reference's augmentation chain (mdp_sensors/observations.py:21) on the rendered
    image, which is grey (R = G = B).  numpy restatement of torchvision's published tensor ops: adjust_saturation = blend(img,
    rgb_to_grayscale(img), f) with grayscale weights (0.2989, 0.587, 0.114) -- they sum to 0.9999, so on a grey image it is the
    scale 0.9999 + 0.0001 f: within 1.8e-4 of the identity for f in [0.2, 1.8], NOT exactly the identity; adjust_hue = RGB -> HSV,
    h += shift, -> RGB: with zero saturation the hue drops out, the identity to rounding.  Neither is modelled by the camera
    kernel (its parity tolerance per pixel is 2e-3); the op ORDER of brightness / contrast, which does matter, is."""
    rng = np.random.RandomState(0)
    v = rng.rand(5, 1, 40, 80).astype(np.float32)
    img = np.repeat(v, 3, 1)

    def grayscale(x):
        return (0.2989 * x[:, 0] + 0.587 * x[:, 1] + 0.114 * x[:, 2])[:, None]

    def adjust_saturation(x, f):
        return np.clip(f * x + (1.0 - f) * grayscale(x), 0, 1)

    def adjust_hue(x, shift):      # torchvision _rgb2hsv / _hsv2rgb
        r, g, b = x[:, 0], x[:, 1], x[:, 2]
        maxc, minc = x.max(1), x.min(1)
        eqc = maxc == minc
        cr = maxc - minc
        ones = np.ones_like(maxc)
        s = cr / np.where(eqc, ones, maxc)
        crd = np.where(eqc, ones, cr)
        rc, gc, bc = (maxc - r) / crd, (maxc - g) / crd, (maxc - b) / crd
        hr = (maxc == r) * (bc - gc)
        hg = ((maxc == g) & (maxc != r)) * (2.0 + rc - bc)
        hb = ((maxc != g) & (maxc != r)) * (4.0 + gc - rc)
        h = np.fmod(np.fmod((hr + hg + hb) / 6.0 + 1.0, 1.0) + shift, 1.0)
        i = np.floor(h * 6.0)
        f = h * 6.0 - i
        i = i.astype(np.int32) % 6
        p_ = np.clip(maxc * (1.0 - s), 0, 1)
        q = np.clip(maxc * (1.0 - f * s), 0, 1)
        t = np.clip(maxc * (1.0 - (1.0 - f) * s), 0, 1)
        a1 = np.stack([maxc, q, p_, p_, t, maxc]), np.stack([t, maxc, maxc, q, p_, p_]), np.stack([p_, p_, t, maxc, maxc, q])
        pick = lambda a: np.take_along_axis(a, i[None], 0)[0]
        return np.stack([pick(a1[0]), pick(a1[1]), pick(a1[2])], 1)

    for f in (0.2, 1.0, 1.8):
        out = adjust_saturation(img, f)
        np.testing.assert_allclose(out, np.clip(img * (0.9999 + 0.0001 * f), 0, 1), rtol=0, atol=1e-6)
        assert np.abs(out - img).max() <= 1.8e-4 + 1e-7
    for shift in (-0.5, -0.2, 0.3, 0.5):
        np.testing.assert_allclose(adjust_hue(img, shift), img, rtol=0, atol=1e-6)
