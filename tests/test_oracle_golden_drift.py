"""Pin the oracle (numpy restatement) to the golden vectors produced by the reference's own functions
(tests/golden/gen_golden.py).  CPU only.  Tolerances: float outputs 1e-6 abs + 1e-6 rel (both sides fp32,
different libm paths for atan2/sqrt); boolean / integer outputs bit-exact."""
import os

import numpy as np
import pytest

from oracle import drift_mdp as M
from oracle import drift_reset as R
from oracle import mathlib as ml
from oracle import params as P

TOL = dict(rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("tag", ["n256", "edges"])
def test_drift_terms_match_reference(golden, tag):
    g = golden(f"drift_mdp_{tag}")
    p = P.drift_params()
    steer = g["joint_pos"][:, 0:2]
    np.testing.assert_allclose(M.side_slip(g["lin_vel_b"], p.slip_min, p.slip_max, p.slip_min_vx), g["side_slip"], **TOL)
    np.testing.assert_allclose(M.vel_dist(g["lin_vel_b"], p.speed_target, p.speed_offset), g["vel_dist"], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(M.track_progress_rate(g["ang_vel_w"]), g["track_progress_rate"])
    np.testing.assert_allclose(M.turn_left_go_right(steer, g["ang_vel_b"], p.tlgr_thresh), g["turn_left_go_right"], **TOL)
    np.testing.assert_allclose(M.energy_through_turn(g["pos"], g["lin_vel_b"], p.straight), g["energy_through_turn"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(M.cross_track_dist(g["pos"], p.straight, p.r_line, p.ctd_offset, p.ctd_p), g["cross_track_dist"], rtol=1e-5, atol=2e-6)
    np.testing.assert_array_equal(M.in_range(g["pos"], p.straight, p.r_in), g["in_range"])
    np.testing.assert_array_equal(M.off_track(g["pos"], p.straight, p.r_out), g["off_track"])
    np.testing.assert_array_equal(M.cart_off_track(g["pos"], p.straight, p.r_in, p.r_out), g["cart_off_track"])
    np.testing.assert_array_equal(np.asarray(p.weight[:7], np.float32), g["weights"])


def test_action_terms_match_reference(golden):
    g = golden("actions")
    for tag, ap in (("rwd", P.mushr_action(0)), ("4wd", P.mushr_action(1)),
                    ("f1tenth", P.mushr_action(1, base_length=0.365, base_width=0.284))):
        assert np.allclose(g[f"{tag}_geom"], [ap.base_length, ap.base_width, ap.wheel_radius, *ap.scale])
        # the wrapper clip is not part of the action term: golden raw == input
        np.testing.assert_array_equal(g[f"{tag}_raw"], g["actions"])
        proc = M.process_actions(g["actions"], ap)
        np.testing.assert_allclose(proc, g[f"{tag}_processed"], **TOL)
        fn = M.rwd_targets if tag == "rwd" else M.fwd_targets
        steer, wheel = fn(proc[:, 0], proc[:, 1], ap)
        np.testing.assert_allclose(steer, g[f"{tag}_steer_pos_target"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(wheel, g[f"{tag}_wheel_vel_target"], rtol=1e-5, atol=1e-4)
    ap = P.mushr_action(1)
    proc = M.process_actions(g["4wd_special_actions"], ap)
    steer, wheel = M.fwd_targets(proc[:, 0], proc[:, 1], ap)
    np.testing.assert_allclose(wheel, g["4wd_special_wheel_vel_target"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(steer, g["4wd_special_steer_pos_target"], rtol=1e-6, atol=1e-6)
    # base class (tanh bounding, reverse allowed, true Ackermann angles)
    ap = P.mushr_action(1)
    ap.bounding, ap.no_reverse = 2, 0
    proc = M.process_actions(g["actions"], ap)
    np.testing.assert_allclose(proc, g["base_processed"], **TOL)
    steer, wheel = M.ackermann_base_targets(proc[:, 0], proc[:, 1], ap)
    np.testing.assert_allclose(steer, g["base_steer_pos_target"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(wheel, g["base_wheel_vel_target"], rtol=1e-5, atol=1e-4)


def test_reset_along_track_matches_reference(golden):
    g = golden("reset_track")
    ref = R.reference_poses(g["u_dists"])
    np.testing.assert_allclose(ref, g["reference_poses"], rtol=1e-6, atol=2e-5)  # degrees: 2e-5 abs on ~300
    pose, vel = R.reset_pose(g["reference_poses"], g["idx"], g["u_xy"], g["u_yaw"], 0.5, 1.0)
    np.testing.assert_allclose(pose, g["pose"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(vel, g["vel"])
    # the in-kernel table is the same data in radians
    t = R.ref_pose_table(g["reference_poses"])
    assert t.shape == (3, 32)
    np.testing.assert_allclose(t[2, :20], np.deg2rad(g["reference_poses"][:, 1, 2]), rtol=1e-6)


def test_curriculum_matches_reference(golden):
    g = golden("curriculum")
    w = [10.0, 0.0, -5000.0]
    got = []
    for step in g["steps"]:
        for k in range(3):
            inc, epi, mx = g["params"][k]
            w[k] = R.increase_reward_weight_over_time(int(step), 250, w[k], inc, int(epi), mx)
        got.append(list(w))
    np.testing.assert_array_equal(np.array(got), g["weights"])
    assert got[-1] == [230.0, 60.0, -11000.0]  # max_increases + 1 increments


def test_math_self_consistency():
    """unpinned helpers: round trips and known answers"""
    rng = np.random.RandomState(0)
    rpy = np.stack([rng.uniform(-3, 3, 500), rng.uniform(-1.5, 1.5, 500), rng.uniform(-3, 3, 500)], -1).astype(np.float32)
    q = ml.quat_from_euler_xyz(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    np.testing.assert_allclose(np.linalg.norm(q, axis=-1), 1.0, atol=1e-6)
    r, p, y = ml.euler_xyz_from_quat(q)
    wrap = lambda a: np.mod(a, 2 * np.pi)
    d = lambda a, b: np.abs(np.mod(a - b + np.pi, 2 * np.pi) - np.pi)
    assert d(r, wrap(rpy[:, 0])).max() < 2e-4 and d(p, wrap(rpy[:, 1])).max() < 2e-4 and d(y, wrap(rpy[:, 2])).max() < 2e-4
    assert (r >= 0).all() and (r < 2 * np.pi + 1e-6).all()
    Rm = ml.matrix_from_quat(q)
    np.testing.assert_allclose(np.einsum("nij,nkj->nik", Rm, Rm), np.broadcast_to(np.eye(3), Rm.shape), atol=2e-6)
    v = rng.normal(size=(500, 3)).astype(np.float32)
    np.testing.assert_allclose(ml.rotate(q, ml.rotate_inverse(q, v)), v, atol=2e-6)
    # yaw 90 deg maps body x to world y
    q90 = ml.quat_from_euler_xyz(0.0, 0.0, np.pi / 2)
    np.testing.assert_allclose(ml.rotate(q90[None], np.array([[1.0, 0, 0]])), [[0, 1, 0]], atol=1e-6)
    # identity -> zero angles; small negative yaw wraps to just below 2pi (reference quirk)
    _, _, yw = ml.euler_xyz_from_quat(ml.quat_from_euler_xyz(0.0, 0.0, -0.01)[None])
    assert abs(yw[0] - (2 * np.pi - 0.01)) < 1e-5


def test_philox_known_answers():
    """Philox4x32 (the keyed generator of every random draw; the HIP kernels are bit-exact to this oracle,
    tests/test_gpu_drift_parity.py::test_philox_bit_exact): the round function and the key schedule at TEN rounds against the three
    known-answer vectors of the Random123 distribution (kat_vectors: counter / key all zero, all ones, and the digits of pi).  The
    draws use the first ROUNDS = 7 of the same rounds (the paper's smallest Crush-resistant count for this width), pinned by the
    distribution's 7-round vectors."""
    from oracle import philox as PH
    assert PH.ROUNDS == 7

    def run(c, k):   # counter = (env, step low, step high, stream), key = (seed low, seed high)
        out = PH.philox4x32(np.array([c[0]]), c[1] | (c[2] << 32), c[3], k[0] | (k[1] << 32), rounds=10)
        return [int(v) for v in out[:, 0]]
    def run7(c, k):
        out = PH.philox4x32(np.array([c[0]]), c[1] | (c[2] << 32), c[3], k[0] | (k[1] << 32))
        return [int(v) for v in out[:, 0]]
    # the seven-round generator the draws use: kat_vectors' `philox4x32 7` lines for the all-zero and the all-ones counter / key
    assert run7((0, 0, 0, 0), (0, 0)) == [0x5F6FB709, 0x0D893F64, 0x4F121F81, 0x4F730A48]
    assert run7((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == [0x5207DDC2, 0x45165E59, 0x4D8EE751, 0x8C52F662]
    assert run((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert run((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert run((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == [0xD16CFE09, 0x94FDCCEB, 0x5001E420,
                                                                                             0x24126EA1]


def test_noise_draws_are_uniform_and_standard_normal():
    """what the observation-noise model (IsaacLab GaussianNoiseCfg, un-vendored: N(0, std)) needs from the generator:
    u01 of the Philox words is U[0, 1) and the Box-Muller pairs are N(0, 1) -- Kolmogorov-Smirnov against scipy's
    distributions on 2 x 10^5 draws, independent across envs / steps / streams (lag correlations)"""
    from scipy import stats

    from oracle import philox as PH
    u = PH.uniform4(np.arange(50000), 5, 0, 42).reshape(-1)
    assert u.min() >= 0.0 and u.max() < 1.0 and stats.kstest(u, "uniform").pvalue > 1e-3
    u8 = PH.uniform8(np.arange(25000), 5, 0, 42)                   # the 16-bit draws of the drift step: open interval, halves independent
    assert u8.min() >= 2.0 ** -17 and u8.max() <= 1.0 - 2.0 ** -17 and stats.kstest(u8.reshape(-1), "uniform").pvalue > 1e-3
    assert np.abs(np.corrcoef(u8) - np.eye(8)).max() < 0.03
    z = PH.normal12(np.arange(20000), 7, 42)                       # [12, n]: the 12 observation-noise normals per env
    assert stats.kstest(z.reshape(-1), "norm").pvalue > 1e-3
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    c = np.corrcoef(z)                                             # the 12 values of an env are mutually uncorrelated ...
    assert np.abs(c - np.eye(12)).max() < 0.03
    z2 = PH.normal12(np.arange(20000), 8, 42)                      # ... and so are consecutive steps and neighbouring envs
    assert abs(np.corrcoef(z[0], z2[0])[0, 1]) < 0.03 and abs(np.corrcoef(z[0, :-1], z[0, 1:])[0, 1]) < 0.03


def test_math_helpers_match_scipy_rotation():
    """the IsaacLab helpers the reference relies on (quat_from_euler_xyz, matrix_from_quat, euler_xyz_from_quat,
    quat_rotate / quat_rotate_inverse -- IsaacLab v2.0.2 is not vendored, so the oracle restates their published
    definitions) against an independent implementation of the same convention: scipy's Rotation, extrinsic x-y-z
    Euler angles, quaternion (w, x, y, z); fp32 rounding (2e-6), angles on the circle"""
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(0)
    rpy = np.stack([rng.uniform(-1.4, 1.4, 1000), rng.uniform(-1.4, 1.4, 1000), rng.uniform(-3.1, 3.1, 1000)], -1)
    ref = Rotation.from_euler("xyz", rpy)
    q = ml.quat_from_euler_xyz(rpy[:, 0], rpy[:, 1], rpy[:, 2])
    sq = ref.as_quat()                                   # scipy: (x, y, z, w)
    sq = np.concatenate([sq[:, 3:4], sq[:, :3]], 1)
    sign = np.sign((q * sq).sum(1))[:, None]             # q and -q are the same rotation
    np.testing.assert_allclose(q, sign * sq, atol=2e-6)
    np.testing.assert_allclose(ml.matrix_from_quat(q), ref.as_matrix(), atol=2e-6)
    r, p, y = ml.euler_xyz_from_quat(q)
    want = np.mod(Rotation.from_quat(np.concatenate([q[:, 1:], q[:, :1]], 1)).as_euler("xyz"), 2 * np.pi)
    d = np.abs(np.stack([r, p, y], 1) - want)
    assert np.minimum(d, 2 * np.pi - d).max() < 5e-6
    v = rng.normal(size=(1000, 3))
    np.testing.assert_allclose(ml.rotate(q, v), ref.apply(v), atol=5e-6)
    np.testing.assert_allclose(ml.rotate_inverse(q, v), ref.inv().apply(v), atol=5e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/source"), reason="the reference tree exists only in the authoring container")
def test_committed_vectors_regenerate_from_the_reference(tmp_path):
    """provenance of tests/golden/*.npz: running gen_golden.py (which imports the reference's own functions) again
    reproduces every committed array exactly"""
    import subprocess
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    env = dict(os.environ, WL_GOLDEN_OUT=str(tmp_path), PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, os.path.join(here, "gen_golden.py")], check=True, env=env, capture_output=True, timeout=600)
    names = sorted(f for f in os.listdir(here) if f.endswith(".npz"))
    assert names == sorted(f for f in os.listdir(tmp_path) if f.endswith(".npz")) and len(names) == 10
    for f in names:
        a, b = np.load(os.path.join(here, f)), np.load(tmp_path / f)
        assert set(a.files) == set(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and np.array_equal(a[k], b[k], equal_nan=True), (f, k)


@pytest.mark.skipif(not os.path.isdir("/root/reference/source"), reason="the reference tree exists only in the authoring container")
@pytest.mark.parametrize("offset", [1000, 2024])
def test_oracle_matches_the_reference_on_fresh_seeds(tmp_path, offset):
    """beyond the committed vectors: new states / actions / query points (every input seed shifted), pushed through the
    reference's own functions by gen_golden.py, and the whole oracle-vs-golden suite run against those files"""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, WL_GOLDEN_OUT=str(tmp_path), WL_GOLDEN_SEED_OFFSET=str(offset), PYTHONDONTWRITEBYTECODE="1")
    subprocess.run([sys.executable, os.path.join(here, "golden", "gen_golden.py")], check=True, env=env, capture_output=True, timeout=600)
    env = dict(os.environ, WL_GOLDEN_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(here, "test_oracle_golden_drift.py"),
                        os.path.join(here, "test_oracle_golden_elev_visual.py"), os.path.join(here, "test_plugin_terms_cpu.py"),
                        "-k", "not regenerate and not fresh_seeds"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
