"""Behavioural validation of the DESIGNED vehicle model (oracle/vehicle.py == the HIP kernel to fp32 tolerance).
There is no dynamics oracle to be equal to (PhysX is closed, USDs are missing -- DESIGN.md section 4), so the model is
checked against physics it must obey: static equilibrium, traction and cornering limits set by mu*g, kinematic
turning radius at low speed, loss of rear grip under wheel-spin."""
import numpy as np
import pytest

from oracle import drift_step as S
from oracle import params as P
from oracle.mathlib import matrix_from_quat

G = 9.81


def _params():
    p = P.drift_params()
    p.enable_pushes = p.enable_corruption = 0
    p.max_episode_length = 10 ** 9
    p.r_out, p.r_in = 1e18, 0.0  # no terminations: an open plane
    return p


def _run(action, steps, mu=0.4, mass=3.4, damp=30.0, state0=None):
    p = _params()
    st = np.zeros((35, 64), np.float32)
    st[3] = 1
    if state0 is not None:
        st[:19] = state0[:, None]
    st[23], st[24], st[25], st[26] = mu, mu, damp, mass
    ep = np.zeros(64, np.int32)
    ref = np.zeros((3, 32), np.float32)
    hist = []
    for k in range(steps):
        S.step(p, st, ep, ref, np.tile(np.asarray(action, np.float32), (64, 1)), 0, k)
        hist.append(st[:19, 0].copy())
    return np.array(hist)


def _body_vel(h):
    R = matrix_from_quat(h[:, 3:7])
    return np.einsum("nji,nj->ni", R, h[:, 7:10])


def test_rest_is_an_equilibrium():
    h = _run([0.0, 0.0], 100)
    assert abs(h[-1, 2]) < 1e-4                      # root origin rests on the ground at nominal load
    assert np.abs(h[-1, 7:13]).max() < 1e-4 and np.abs(h[-1, 13:17]).max() < 1e-3
    heavy = _run([0.0, 0.0], 100, mass=4.4)          # +1 kg settles k*dz = dm*g/4 lower and stays there
    assert heavy[-1, 2] == pytest.approx(-1.0 * G / (4 * 3000.0), abs=2e-4)
    assert np.abs(heavy[-1, 7:13]).max() < 1e-3


def test_straight_line_traction_limit_and_top_speed():
    mu = 0.4
    h = _run([1.0, 0.0], 200, mu=mu)
    v = np.hypot(h[:, 7], h[:, 8])
    acc = np.diff(v) / 0.02
    mu_eff = mu * 1.1                                 # wheel x ground, "multiply" combine
    # rear-wheel drive: a <= mu*g*(rear static share + load transfer a*h/(g*L))
    bound = mu_eff * G * (0.5 + acc.max() * 0.06 / (G * 0.325))
    assert 0.8 * mu_eff * G * 0.5 < acc.max() <= bound * 1.03
    assert v[-1] == pytest.approx(3.0, abs=0.05)      # 60 rad/s * 0.05 m, no reverse, no overshoot
    assert abs(h[-1, 1]) < 1e-3 and abs(h[-1, 12]) < 1e-3   # stays straight
    assert np.all(h[:, 13:15] <= 60.01)               # driven wheels never exceed their velocity target


def test_low_speed_turn_follows_ackermann_kinematics():
    h = _run([0.2, 0.4], 500)                         # 0.6 m/s target, gentle steer
    v, wz = np.hypot(h[-1, 7], h[-1, 8]), h[-1, 12]
    steer = h[-1, 17]
    assert steer == pytest.approx(np.tan(0.4 * 0.488), rel=1e-3)   # the joint target is tan(delta) (reference quirk)
    r_kin = 0.325 / np.tan(steer)
    # both rear wheels are driven to the SAME speed (rc_car_actions.py:24-27: no differential), which resists yaw:
    # the car must understeer relative to the kinematic radius, but only moderately at 0.6 m/s
    assert r_kin < v / wz < 1.35 * r_kin
    assert v * wz < 0.44 * G


def test_cornering_is_friction_limited_and_wheelspin_steps_the_rear_out():
    mu_eff = 0.4 * 1.1
    gentle = _run([0.4, 0.5], 400)
    hard = _run([1.0, 0.5], 400)
    for h in (gentle, hard):
        a_h = np.linalg.norm(np.diff(h[:, 7:9], axis=0), axis=1) / 0.02
        assert a_h[100:].max() <= mu_eff * G * 1.05               # horizontal CoM acceleration never beats mu*g
    beta = lambda h: np.arctan2(_body_vel(h)[:, 1], _body_vel(h)[:, 0])[-100:].mean()
    assert abs(beta(gentle)) < 0.2                                # grips
    assert beta(hard) < -0.4                                      # rear slides outwards in a left turn: a drift
    assert hard[-100:, 12].mean() > 2.0 * gentle[-100:, 12].mean()   # and yaws much faster


def test_braking_is_friction_limited():
    h = _run([1.0, 0.0], 200)
    h2 = _run([0.0, 0.0], 100, state0=h[-1])          # throttle to zero: the motor holds the rear wheels, car brakes
    v = np.hypot(h2[:, 7], h2[:, 8])
    dec = -np.diff(v) / 0.02
    assert dec.max() <= 0.44 * G * 0.62               # only the rear axle brakes, minus forward load transfer
    assert v[-1] < v[0] - 1.0


def test_more_grip_more_acceleration():
    a = lambda mu: np.diff(np.hypot(*_run([1.0, 0.0], 40, mu=mu)[:, 7:9].T)).max() / 0.02
    assert a(0.5) > a(0.4) > a(0.3)


def test_free_flight_converges_to_the_rigid_body_equations():
    """airborne (no contact), the integrator must solve Euler's rigid-body equations + quaternion kinematics + gravity:
    against scipy's solve_ivp (RK45, rtol 1e-10) of the same equations the state after 0.1 s agrees to the integrator's
    first-order error, and that error halves when the step is halved"""
    from scipy.integrate import solve_ivp

    from oracle import vehicle as V
    vp = P.drift_params().vehicle
    n, T = 8, 0.1
    rng = np.random.RandomState(1)
    mass = np.full(n, 3.4, np.float32)
    x0 = np.tile(np.array([0.0, 0.0, 5.0], np.float32), (n, 1))
    q0 = rng.normal(size=(n, 4)).astype(np.float32)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    v0 = rng.uniform(-2, 2, (n, 3)).astype(np.float32)
    w0 = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    I = 3.4 * np.array([vp.gyr_x ** 2, vp.gyr_y ** 2, vp.gyr_z ** 2])

    def rhs(t, y):
        q, wb = y[6:10], y[10:13]
        qw, qx, qy, qz = q
        R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                      [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                      [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        ww = R @ wb
        dq = 0.5 * np.array([-ww[0] * qx - ww[1] * qy - ww[2] * qz, ww[0] * qw + ww[1] * qz - ww[2] * qy,
                             -ww[0] * qz + ww[1] * qw + ww[2] * qx, ww[0] * qy - ww[1] * qx + ww[2] * qw])
        return np.concatenate([y[3:6], [0.0, 0.0, -vp.gravity], dq, -np.cross(wb, I * wb) / I])

    exact = np.array([solve_ivp(rhs, (0, T), np.concatenate([x0[i], v0[i], q0[i], w0[i]]).astype(np.float64), rtol=1e-10,
                                atol=1e-12).y[:, -1] for i in range(n)])

    def integrate(h):
        x, q, v, wb = x0.copy(), q0.copy(), v0.copy(), w0.copy()
        wheel, th, om = np.zeros((n, 4), np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32)
        z = np.zeros(n, np.float32)
        for _ in range(int(round(T / h))):
            x, q, v, wb, wheel, th, om = V.substep(x, q, v, wb, wheel, th, om, z, np.zeros((n, 4), np.float32), mass,
                                                   np.full(n, 0.4, np.float32), np.full(n, 0.4, np.float32), z + 30, vp, h)
        sign = np.sign((q * exact[:, 6:10]).sum(1))[:, None]
        return (np.abs(x - exact[:, 0:3]).max(), np.abs(v - exact[:, 3:6]).max(), np.abs(sign * q - exact[:, 6:10]).max(),
                np.abs(wb - exact[:, 10:13]).max())

    e1, e2 = integrate(0.005), integrate(0.0025)
    assert max(e1) < 2e-2, e1                                  # first-order error of the 5 ms step over 0.1 s
    assert e1[3] > 1e-4 and all(b < 0.62 * a + 2e-5 for a, b in zip(e1, e2)), (e1, e2)   # ~ halves with the step


def test_steering_drive_converges_to_the_pd_joint_equation():
    """the implicit PD steering drive (J theta'' = kp (target - theta) - kd theta', hound.py:5-12; very stiff: kd / J =
    5e4 1/s) in its linear regime against scipy's stiff solver: a small step response agrees to first order in the step
    length and the error halves with it; a step far beyond the steering range rides the 10 rad/s rate limit"""
    from scipy.integrate import solve_ivp

    from oracle import vehicle as V
    vp = P.drift_params().vehicle
    J, kp, kd = vp.steer_inertia, vp.steer_kp, vp.steer_kd
    target, T = 0.02, 0.1
    exact = solve_ivp(lambda t, y: [y[1], (kp * (target - y[0]) - kd * y[1]) / J], (0, T), [0.0, 0.0], method="Radau", rtol=1e-10,
                      atol=1e-13).y[0, -1]

    def run(h, tgt, t_end):
        th, om = np.zeros(1, np.float32), np.zeros(1, np.float32)
        for _ in range(int(round(t_end / h))):
            th, om = V.steer_update(th, om, np.float32(tgt), vp, np.float32(h))
        return float(th[0]), float(om[0])

    e1, e2 = abs(run(0.005, target, T)[0] - exact), abs(run(0.0025, target, T)[0] - exact)
    assert abs(exact - target * (1 - np.exp(-kp / kd * T))) < 2e-4 * target          # the slow pole kp / kd dominates
    assert e1 < 0.03 * target and 0.4 * e1 < e2 < 0.62 * e1, (e1, e2, exact)
    th, om = run(0.005, 2.0, 0.02)                       # a step far beyond the steering range: rides the rate limit
    assert abs(om - vp.steer_vel_limit) < 1e-4 and abs(th - 0.02 * vp.steer_vel_limit) < 0.051
