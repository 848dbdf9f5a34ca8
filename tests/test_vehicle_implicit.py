"""The linearly implicit integrator of the DESIGNED vehicle model (oracle/vehicle.py, vp.implicit = 1; the HIP kernels equal it to
fp32 tolerance): the elevation and visual tasks step it ONCE per sim.dt -- h = 10 ms and 20 ms, the reference's own physics rate
(mushr_elevation_env_cfg.py:461-462, mushr_visual_env_cfg.py:435-436) -- where the explicit scheme needed h <= 5 ms and a cap on
the tyre stiffness.  There is no dynamics oracle to be equal to (PhysX is closed), so the scheme is held to what a one-step
method must deliver: the force laws' steady states whatever h is, first-order convergence to the fine-step solution of the
same force laws, stability at mu = 2 near rest, and the physics bounds of tests/test_vehicle_behaviour.py."""
import numpy as np
import pytest

from oracle import drift_mdp as M
from oracle import heightfield as H
from oracle import params as P
from oracle import vehicle as V
from oracle.elev_step import elev_params, ground_fn
from oracle.mathlib import matrix_from_quat
from oracle.visual_step import visual_params

G = 9.81
F = np.float32


def _vehicle(h_ms, implicit=1, mu_ground=2.0):
    vp = P.mushr_vehicle(drive=1, motor_limit=0.25, substeps=1, ground_mu=(mu_ground, mu_ground), implicit=implicit)
    return vp, F(h_ms * 1e-3)


def _rest(n, vp, z=None):
    x = np.zeros((n, 3), F)
    x[:, 2] = vp.cg_z if z is None else z
    q = np.zeros((n, 4), F)
    q[:, 0] = 1
    return [x, q, np.zeros((n, 3), F), np.zeros((n, 3), F), np.zeros((n, 4), F), np.zeros(n, F), np.zeros(n, F)]


def _run(vp, h, T, action, st=None, n=4, mass=3.35, mu_w=(1.0, 1.0), damp=1000.0, ground=V.flat_ground, every=None):
    """integrate T seconds under a constant action (or a function of time); -> history [K, 17] of env 0 sampled every `every`
    seconds (default: every sub-step) and the final state"""
    st = [a.copy() for a in (st if st is not None else _rest(n, vp))]
    n = st[0].shape[0]
    m, ms, md, dm = (np.full(n, v, F) for v in (mass, mu_w[0], mu_w[1], damp))
    ap = P.mushr_action(1)
    steps = int(round(T / float(h)))
    per = 1 if every is None else int(round(every / float(h)))
    hist = []
    for k in range(steps):
        a = action(k * float(h)) if callable(action) else action
        a = np.tile(np.asarray(a, F), (n, 1))
        proc = M.process_actions(M.clip_action(a), ap)
        steer2, wt = M.fwd_targets(proc[:, 0], proc[:, 1], ap)
        st = list(V.substep(*st, steer2[:, 0].astype(F), wt.astype(F), m, ms, md, dm, vp, h, ground))
        if (k + 1) % per == 0:
            hist.append(np.concatenate([st[0][0], st[1][0], st[2][0], st[3][0], st[4][0]]))
    return np.array(hist), st


def _body(hist):
    R = matrix_from_quat(hist[:, 3:7])
    return np.einsum("nji,nj->ni", R, hist[:, 7:10])


@pytest.mark.parametrize("h_ms", [10, 20])
def test_rest_is_an_equilibrium_at_mu_2(h_ms):
    vp, h = _vehicle(h_ms)
    hist, _ = _run(vp, h, 4.0, [0.0, 0.0], mass=3.4)
    assert abs(hist[-1, 2] - vp.cg_z) < 1e-4                        # the root origin rests on the ground at nominal load
    assert np.abs(hist[-1, 7:13]).max() < 1e-4 and np.abs(hist[-1, 13:17]).max() < 1e-3
    heavy, _ = _run(vp, h, 4.0, [0.0, 0.0], mass=4.4)               # +1 kg settles k dz = dm g / 4 lower, whatever h is
    assert heavy[-1, 2] - vp.cg_z == pytest.approx(-1.0 * G / (4 * 3000.0), abs=2e-4)
    assert np.abs(heavy[-1, 7:13]).max() < 1e-3
    st = _rest(4, vp)                                               # a shove near rest dies out: no limit cycle at mu = 2
    st[2][:, 1], st[3][:, 2] = 0.3, 1.0
    pert, _ = _run(vp, h, 2.0, [0.0, 0.0], st=st)
    assert np.abs(pert[-1, 7:13]).max() < 1e-4 and np.isfinite(pert).all()
    assert np.abs(pert[:, 7:13]).max() <= 1.0 + 1e-3                # and never grows on the way


def test_steady_cornering_does_not_depend_on_the_step():
    """a steady state of the force laws in the body frame is a fixed point of the scheme: (v_x, v_y, w_z) of a held turn agree
    between h = 20, 10 and 2.5 ms, and with the explicit scheme at a step where its stiffness cap never binds"""
    ss = {}
    for h_ms, implicit in ((20, 1), (10, 1), (2.5, 1), (0.5, 0)):
        vp, h = _vehicle(h_ms, implicit)
        hist, _ = _run(vp, h, 4.0, [0.6, 0.6], every=0.02)
        vb = _body(hist[-20:])
        assert np.ptp(hist[-20:, 12]) < 2e-3                         # it IS steady
        ss[(h_ms, implicit)] = np.array([vb[:, 0].mean(), vb[:, 1].mean(), hist[-20:, 12].mean()])
    ref = ss[(0.5, 0)]
    assert ref[0] > 1.5 and abs(ref[2]) > 1.0
    for k, v in ss.items():
        np.testing.assert_allclose(v, ref, atol=4e-3, err_msg=str(k))


def _tilted_plane(slope):
    """ground z = slope * x: -> height, unit normal"""
    nrm = np.array([-slope, 0.0, 1.0]) / np.sqrt(1 + slope * slope)

    def g(xy):
        return (F(slope) * xy[:, 0]).astype(F), np.tile(nrm.astype(F), (xy.shape[0], 1))
    return g


def test_creep_on_a_slope_does_not_depend_on_the_step():
    """parked on a 15 % slope the regularised-Coulomb tyres let the car creep at F / K: the same terminal speed at every h,
    implicit or explicit -- the scheme adds no damping of its own to a steady state"""
    slope = 0.15
    ground = _tilted_plane(slope)
    v_end = {}
    for h_ms, implicit in ((20, 1), (10, 1), (5, 1), (1, 0)):
        vp, h = _vehicle(h_ms, implicit)
        st = _rest(4, vp)
        hist, _ = _run(vp, h, 3.0, [0.0, 0.0], st=st, ground=ground, every=0.02)
        assert np.ptp(hist[-10:, 7]) < 1e-4
        v_end[(h_ms, implicit)] = hist[-1, 7:10].copy()
    ref = v_end[(1, 0)]
    assert 1e-3 < np.linalg.norm(ref) < 0.05                         # creeps downhill, slowly (the wheels are held by the servo)
    for k, v in v_end.items():
        np.testing.assert_allclose(v, ref, atol=1.5e-4, err_msg=str(k))


def test_first_order_convergence_to_the_fine_step_solution():
    """launch + a steering reversal (mu = 2): against the explicit scheme at h = 0.25 ms the position error of the implicit
    scheme shrinks about linearly with h, and at h = 10 ms it is the explicit 5 ms scheme's within a factor of three"""
    def act(t):
        return [0.8, 0.5 if t < 1.0 else -0.5]
    T = 2.0
    vp, h = _vehicle(0.25, 0)
    ref, _ = _run(vp, h, T, act, every=0.1)
    err = {}
    for h_ms, implicit in ((20, 1), (10, 1), (5, 1), (5, 0)):
        vp, h = _vehicle(h_ms, implicit)
        hist, _ = _run(vp, h, T, act, every=0.1)
        err[(h_ms, implicit)] = np.abs(hist[:, :2] - ref[:, :2]).max()
    path = np.linalg.norm(np.diff(ref[:, :2], axis=0), axis=1).sum()
    assert path > 2.5
    assert err[(20, 1)] < 0.03 * path, err                           # 20 ms: a few percent of the distance driven
    assert err[(10, 1)] < 0.62 * err[(20, 1)] and err[(5, 1)] < 0.62 * err[(10, 1)], err
    assert err[(10, 1)] < 3.0 * err[(5, 0)] + 2e-3, err


@pytest.mark.parametrize("h_ms", [10, 20])
def test_traction_cornering_and_kinematics_bounds(h_ms):
    vp, h = _vehicle(h_ms)
    mu = 2.0
    hist, _ = _run(vp, h, 3.0, [1.0, 0.0], every=0.02)               # launch: 4WD, motor-limited well below mu g
    v = np.hypot(hist[:, 7], hist[:, 8])
    acc = np.diff(v) / 0.02
    f_motor = 4 * 0.25 / 0.05                                         # four motors at their 0.25 N m limit on r = 0.05
    assert 0.7 * f_motor / 3.35 < acc.max() <= min(mu * G, f_motor / 3.35) * 1.05
    assert v[-1] == pytest.approx(3.0, abs=0.05) and np.all(hist[:, 13:17] <= 60.05)   # 60 rad/s x 0.05 m, no overshoot
    assert abs(hist[-1, 1]) < 1e-3 and abs(hist[-1, 12]) < 1e-3       # stays straight
    slow, _ = _run(vp, h, 6.0, [0.2, 0.4], every=0.02)                # 0.6 m/s, gentle steer: the kinematic radius
    vs, wz = np.hypot(slow[-1, 7], slow[-1, 8]), slow[-1, 12]
    r_kin = 0.325 / np.tan(0.4 * 0.488)
    assert 0.95 * r_kin < vs / wz < 1.35 * r_kin
    hard, _ = _run(vp, h, 4.0, [1.0, 0.6], every=0.02)                # cornering: the CoM's horizontal acceleration <= mu g
    a_h = np.linalg.norm(np.diff(hard[:, 7:9], axis=0), axis=1) / 0.02
    assert a_h[50:].max() <= mu * G * 1.05
    assert np.isfinite(hard).all() and np.abs(hard[:, 7:10]).max() < 3.5


def test_free_flight_is_the_explicit_scheme():
    """no wheel in contact: G^ = 0 and the implicit update IS the explicit one (gravity and the gyroscopic term only)"""
    rng = np.random.RandomState(1)
    n = 8
    out = []
    for implicit in (0, 1):
        vp, h = _vehicle(10, implicit)
        st = _rest(n, vp, z=5.0)
        q = rng.normal(size=(n, 4)).astype(F) if implicit == 0 else q0
        q0 = q
        st[1] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)
        st[2] = np.tile(np.array([1.0, -2.0, 0.5], F), (n, 1))
        st[3] = np.tile(np.array([2.0, -1.0, 3.0], F), (n, 1))
        _, s = _run(vp, h, 0.2, [0.0, 0.0], st=st)
        out.append(s)
    for a, b in zip(*out):
        np.testing.assert_allclose(a, b, atol=2e-5)


def test_policy_like_driving_on_the_terrain_stays_bounded():
    """elevation task parameters on the synthetic terrain, smooth random actions, 6 s at h = 10 ms and 20 ms: finite, the speed
    stays near the 3 m/s wheel-speed target, the body rates bounded"""
    hf = H.make_terrain()
    ground = ground_fn(hf)
    p = elev_params()
    n = 64
    rng = np.random.RandomState(5)
    xy = rng.uniform(-8, 8, (n, 2)).astype(F)
    yaw = rng.uniform(-3.14, 3.14, n).astype(F)
    zt, _ = ground(xy)
    for h_ms in (10, 20):
        vp, h = _vehicle(h_ms, 1, mu_ground=1.0)
        st = _rest(n, vp)
        st[0] = np.concatenate([xy, (np.maximum(0.25, zt + 0.06) + vp.cg_z)[:, None]], -1).astype(F)
        st[1] = np.stack([np.cos(yaw / 2), 0 * yaw, 0 * yaw, np.sin(yaw / 2)], -1).astype(F)
        m, ms, md, dm = (np.full(n, v, F) for v in (3.35, 2.0, 1.0, 1000.0))
        x = rng.uniform(-1, 1, (n, 2))
        vmax = wmax = 0.0
        for k in range(int(6.0 / float(h))):
            if k % int(round(0.1 / float(h))) == 0:
                x = 0.8 * x + 0.35 * rng.randn(n, 2)
                a = np.clip(x, -1, 1).astype(F)
                a[:, 0] = np.abs(a[:, 0])
                proc = M.process_actions(M.clip_action(a), p.action)
                steer2, wt = M.fwd_targets(proc[:, 0], proc[:, 1], p.action)
            st = list(V.substep(*st, steer2[:, 0].astype(F), wt.astype(F), m, ms, md, dm, vp, h, ground))
            vmax = max(vmax, float(np.linalg.norm(st[2], axis=1).max()))
            wmax = max(wmax, float(np.abs(st[3]).max()))
        assert all(np.isfinite(a).all() for a in st)
        assert vmax < 4.5 and wmax < 30.0, (h_ms, vmax, wmax)


def test_rolled_over_and_tumbling_cars_stay_bounded_at_20_ms():
    """The visual tasks have no rollover termination and the model no chassis collision: a car on its side or roof meets the ground with
    its wheel spheres only.  Two safeguards keep such states bounded at h = 20 ms (round 6: before them one car in ~10^5 env-steps spun
    itself up to 300 rad/s and out of fp32): the normal spring's force is capped at susp_fmax (24 x the static wheel load) and the implicit
    update of a car tilted by more than ~40 degrees uses an isotropic over-estimate of the damping matrix.  Here: cars thrown onto the
    synthetic terrain in every orientation, spinning, driven with N(0, 1) actions at mu = 2 for 4 s."""
    hf = H.make_terrain()
    ground = ground_fn(hf)
    vp, h = _vehicle(20)
    n = 256
    rng = np.random.RandomState(11)
    xy = rng.uniform(-10, 10, (n, 2)).astype(F)
    zt, _ = ground(xy)
    st = _rest(n, vp)
    st[0] = np.concatenate([xy, (zt + rng.uniform(0.05, 0.4, n))[:, None]], -1).astype(F)
    q = rng.normal(size=(n, 4))
    st[1] = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(F)           # every orientation
    st[2] = rng.uniform(-2, 2, (n, 3)).astype(F)
    st[3] = rng.uniform(-8, 8, (n, 3)).astype(F)
    m, ms, md, dm = (np.full(n, v, F) for v in (3.4, 1.0, 1.0, 1000.0))
    ap = P.mushr_action(1)
    vmax = wmax = 0.0
    for k in range(200):
        if k % 10 == 0:
            a = np.clip(rng.randn(n, 2), -1.5, 1.5).astype(F)
            proc = M.process_actions(M.clip_action(a), ap)
            steer2, wt = M.fwd_targets(proc[:, 0], proc[:, 1], ap)
        st = list(V.substep(*st, steer2[:, 0].astype(F), wt.astype(F), m, ms, md, dm, vp, h, ground))
        assert all(np.isfinite(a_).all() for a_ in st), k
        if k >= 25:                                      # after the throw-in has landed
            vmax, wmax = max(vmax, float(np.abs(st[2]).max())), max(wmax, float(np.abs(st[3]).max()))
    assert vmax < 12.0 and wmax < 60.0, (vmax, wmax)


def test_the_spring_cap_is_out_of_reach_of_driving_and_of_the_spawn_drop():
    """susp_fmax (200 N = 6.7 cm of penetration, more than a wheel radius) against what the wheels' springs carry: 8.3 N at rest, and the
    0.25 m spawn drop of the elevation task compresses them to 2.4 cm"""
    vp, h = _vehicle(10)
    assert vp.susp_fmax == pytest.approx(24 * 3.4 * G / 4)
    zrel = vp.wheel_z - vp.cg_z
    st = _rest(4, vp, z=vp.cg_z + 0.25)
    m, ms, md, dm = (np.full(4, v, F) for v in (3.4, 1.0, 1.0, 1000.0))
    z = np.zeros(4, F)
    pen = 0.0
    for k in range(80):
        st = list(V.substep(*st, z, np.zeros((4, 4), F), m, ms, md, dm, vp, h))
        pen = max(pen, float(vp.wheel_radius - (st[0][:, 2] + zrel).min()))
    assert 0.006 < pen and 3000.0 * pen < 0.5 * vp.susp_fmax, pen            # a hard landing: 9 x the static 2.8 mm, far from 6.7 cm
    assert abs(float(st[0][0, 2]) - vp.cg_z) < 1e-4 and np.abs(st[2]).max() < 1e-3      # and it comes to rest where it should
