"""Rigid-body + 4-tyre vehicle integrator, numpy float32.  DESIGNED MODEL (parity unpinned): replaces the PhysX
articulation the reference steps at mushr_drift_env_cfg.py:393-394 (dt 0.005 x decimation 4).  PhysX is closed
source and the robot USDs are missing (.MISSING_LARGE_BLOBS), so this file is the executable specification that
the HIP kernel (wheeledlab_amd/csrc/wl_vehicle.h) must match to fp32 tolerance; DESIGN.md section 4 derives it.

Reference-owned constants used here: actuator gains / limits (wheeledlab_assets/wheeledlab_assets/hound.py:4-52),
geometry (wheeledlab_tasks/common/actions.py:17-20), friction + "multiply" combine (mushr_drift_env_cfg.py:45-50,
98-109).

State per env: x (CoM, world), q (wxyz), v (CoM lin vel, world), w_b (ang vel, body), wheel spin [4] in order
bl, br, fl, fr, steer angle + rate.

Two integrators of the SAME force laws (vp.implicit):
  0  explicit: semi-implicit Euler of the body under the wheel forces evaluated at the current state (wheel spin and steering
     joint implicit).  Stable for K h / m <~ 1: h <= 5 ms and a cap on the tyre's secant stiffness.  The drift tasks
     (sim.dt = 5 ms, mushr_drift_env_cfg.py:393) step with it.
  1  linearly implicit (round 6): the body's velocity increment solves  (M + h G^) du = h r(u_n)  with r the body-frame
     residual (forces + gravity - frame rotation, torques - gyroscopic term) and G^ a structured approximation of the contact
     forces' damping matrix -d r / d u: the exact 3 x 3 block of the in-plane motion (v_x, v_y, w_z) for the nominal wheel
     positions, and the diagonal for heave / roll / pitch.  A Rosenbrock-W step: first-order consistent for ANY G^, every
     steady state of the force laws (rest, steady cornering, constant creep on a slope) is a fixed point whatever h is, and
     stable without a stiffness cap at the reference's own physics rate -- h = sim.dt = 10 ms (elevation,
     mushr_elevation_env_cfg.py:461) and 20 ms (visual, mushr_visual_env_cfg.py:435), where PhysX runs its implicit TGS solver
     (wheeledlab_assets/mushr.py:22-36).
"""
import numpy as np

from .mathlib import F, f32, matrix_from_quat

WHEELS = ("bl", "br", "fl", "fr")


def dc_motor_limits(w, vp):
    """IsaacLab DCMotor clip (unpinned): tau_max = clip(sat*(1 - w/w_lim), 0, lim), ..."""
    sat, lim, wl = F(vp.motor_sat), F(vp.motor_limit), F(vp.motor_vel_limit)
    hi = np.clip(sat * (F(1) - w / wl), F(0), lim)
    lo = np.clip(sat * (F(-1) - w / wl), -lim, F(0))
    return lo.astype(F), hi.astype(F)


def steer_update(th, om, target, vp, h):
    """implicit (backward-Euler) PD drive, effort- and rate-limited (hound.py:5-12)"""
    J, kp, kd = F(vp.steer_inertia), F(vp.steer_kp), F(vp.steer_kd)
    e = target - th
    om_n = (om + h * kp * e / J) / (F(1) + h * kd / J + h * h * kp / J)
    tau = np.clip(J * (om_n - om) / h, -F(vp.steer_effort), F(vp.steer_effort))
    om_n = np.clip(om + h * tau / J, -F(vp.steer_vel_limit), F(vp.steer_vel_limit))
    return (th + h * om_n).astype(F), om_n.astype(F)


def flat_ground(xy):
    n = xy.shape[0]
    return np.zeros(n, F), np.tile(f32([0, 0, 1]), (n, 1))


def solve_spd3(a11, a12, a13, a22, a23, a33, b1, b2, b3):
    """LDL^T solve of a symmetric positive definite 3 x 3 system, elementwise over envs (no pivoting: the matrices here are
    identity + positive semi-definite)"""
    i1 = F(1) / a11
    l21, l31 = a12 * i1, a13 * i1
    d2 = a22 - l21 * a12
    t32 = a23 - l31 * a12
    i2 = F(1) / d2
    l32 = t32 * i2
    d3 = a33 - l31 * a13 - l32 * t32
    y2 = b2 - l21 * b1
    y3 = b3 - l31 * b1 - l32 * y2
    x3 = y3 / d3
    x2 = y2 * i2 - l32 * x3
    x1 = b1 * i1 - l21 * x2 - l31 * x3
    return x1.astype(F), x2.astype(F), x3.astype(F)


def implicit_body_update(v, wb, R, Fw, Tb, Ib, mass, S, vp, h):
    """(M + h G^) du = h r: the linearly implicit velocity update of the module docstring.  Fw: total force incl. gravity
    (world), Tb: total contact torque about the CoM (body), S: the sums of the wheels' Jacobian pieces.
    Unknowns in the body frame; the in-plane block is solved in (dv_x, dv_y, gyr_z dw_z), which makes it symmetric with the
    mass divided out:  [1 + q Sxx, q Sxy, q Sxw / gz; ., 1 + q Syy, q Syw / gz; ., ., 1 + q Sww / gz^2],  q = h / m."""
    q = (h / mass).astype(F)
    gx, gy, gz = F(vp.gyr_x), F(vp.gyr_y), F(vp.gyr_z)
    vb = np.einsum("nji,nj->ni", R, v).astype(F)
    rot = np.cross(wb, vb).astype(F)                       # d/dt of a world-constant vector seen from the body frame is -w x v
    ab = (np.einsum("nji,nj->ni", R, Fw) / mass[:, None] - rot).astype(F)
    alpha = ((Tb - np.cross(wb, Ib * wb)) / Ib).astype(F)
    igz = F(1) / gz
    # heave / roll / pitch: diagonal.  Normal direction: the spring-damper's damping with the position update folded in,
    # c + h k, for every wheel in contact at its nominal lever (half track / half wheelbase); the tyres act on roll and pitch
    # through the height of the CoM above the contact patches.
    Dn = F(vp.susp_c) + h * F(vp.susp_k)
    az = (F(vp.wheel_z) - F(vp.cg_z)) - F(vp.wheel_radius)
    by2 = F(vp.half_track) * F(vp.half_track)
    bx2 = F(0.5) * (F(vp.half_wheelbase_f) * F(vp.half_wheelbase_f) + F(vp.half_wheelbase_r) * F(vp.half_wheelbase_r))
    nD = S["nc"] * Dn
    # The nominal geometry is that of a car standing on its wheels.  Tilted by more than ~40 degrees (world up in the body frame,
    # R[2, 2] < 0.75: on its side, on its roof, tumbling -- the visual task has no rollover termination) the contact normals and
    # levers are anywhere, and G^ falls back to an isotropic bound that over-estimates the damping matrix whatever the geometry
    # (J^T D J <= 2 d_max blockdiag(I, |arm|^2 I) per wheel): g = 2 (sum of the wheels' kx + ky + contact damping) on every
    # translation, g rho^2 on every rotation, rho the longest lever.  Over-estimating G^ only slows stiff modes down; an
    # under-estimate at h = 20 ms let a car that had landed on its roof spin itself up to 300 rad/s and beyond fp32.
    tilted = R[:, 2, 2] < F(0.75)
    g_iso = F(2) * (S["xx"] + S["yy"] + nD)
    rho2 = bx2 + by2 + az * az
    one = np.ones_like(q)
    a11 = np.where(tilted, one + q * g_iso, one + q * S["xx"])
    a22 = np.where(tilted, one + q * g_iso, one + q * S["yy"])
    a33 = np.where(tilted, one + q * g_iso * rho2 * igz * igz, one + q * S["ww"] * igz * igz)
    off = np.where(tilted, F(0), F(1)).astype(F)
    dvx, dvy, dwz_g = solve_spd3(a11, off * q * S["xy"], off * q * S["xw"] * igz, a22, off * q * S["yw"] * igz, a33,
                                 h * ab[:, 0], h * ab[:, 1], h * alpha[:, 2] * gz)
    dvz = h * ab[:, 2] / np.where(tilted, one + q * g_iso, one + q * nD)
    dwx = h * alpha[:, 0] / np.where(tilted, one + q * g_iso * rho2 / (gx * gx), one + q * (az * az * S["yy"] + by2 * nD) / (gx * gx))
    dwy = h * alpha[:, 1] / np.where(tilted, one + q * g_iso * rho2 / (gy * gy), one + q * (az * az * S["xx"] + bx2 * nD) / (gy * gy))
    dvb = np.stack([dvx, dvy, dvz], -1).astype(F) + h * rot
    v = (v + np.einsum("nij,nj->ni", R, dvb)).astype(F)
    wb = (wb + np.stack([dwx, dwy, dwz_g * igz], -1)).astype(F)
    return v, wb


def substep(x, q, v, wb, wheel, th, om, steer_target, wheel_target, mass, mu_s_w, mu_d_w, damp, vp, h, ground=flat_ground, probe=None):
    """one integrator sub-step of length h.  Arrays are [N,...] float32.  Returns the new state tuple.
    probe: a dict the parity tests pass to learn which envs sit on one of the model's discontinuities during this sub-step (the
    causes a HIP-vs-oracle difference beyond rounding can have): per env, `contact` -- the in-contact pattern of the four wheels as
    a bit mask, appended per sub-step -- and `fz_margin` -- the smallest distance of a wheel from making or breaking contact, in
    newtons of normal force (|k pen - c v_n| in contact, k |pen| above the ground), minimised over the sub-steps"""
    h = F(h)
    r = F(vp.wheel_radius)
    th, om = steer_update(th, om, steer_target, vp, h)
    R = matrix_from_quat(q)
    ww = np.einsum("nij,nj->ni", R, wb).astype(F)
    mu_s = (mu_s_w * F(vp.ground_mu_s)).astype(F)
    mu_d = np.minimum(mu_d_w * F(vp.ground_mu_d), mu_s).astype(F)
    cs, sn = np.cos(th), np.sin(th)
    Ftot = np.zeros_like(x)
    Ttot = np.zeros_like(x)
    new_wheel = np.empty_like(wheel)
    Iw, bw = F(vp.wheel_inertia), F(vp.wheel_damping)
    zrel = F(vp.wheel_z) - F(vp.cg_z)
    implicit = bool(getattr(vp, "implicit", 0))
    S = {k: np.zeros(x.shape[0], F) for k in ("xx", "yy", "xy", "xw", "yw", "ww", "nc")}
    for i, name in enumerate(WHEELS):
        front = name[0] == "f"
        px = F(vp.half_wheelbase_f) if front else -F(vp.half_wheelbase_r)
        py = F(vp.half_track) if name[1] == "l" else -F(vp.half_track)
        pb = f32([px, py, zrel])
        arm_c = (R @ pb).astype(F)                                  # CoM -> wheel centre, world
        cw = x + arm_c
        zg, nrm = ground(cw[:, :2])
        pen = r - (cw[:, 2] - zg) * nrm[:, 2]
        arm = arm_c - r * nrm                                       # CoM -> contact point
        vcp = v + np.cross(ww, arm)
        vn = (vcp * nrm).sum(-1)
        # spring (its force capped at susp_fmax: a penalty spring is meaningless centimetres deep, where only tumbling cars get) + damper,
        # no adhesion
        Fz = np.where(pen > 0, np.maximum(np.minimum(F(vp.susp_k) * pen, F(getattr(vp, 'susp_fmax', 1e30))) - F(vp.susp_c) * vn, F(0)), F(0)).astype(F)
        if probe is not None:
            m_i = np.where(pen > 0, np.abs(np.minimum(F(vp.susp_k) * pen, F(getattr(vp, 'susp_fmax', 1e30))) - F(vp.susp_c) * vn), F(vp.susp_k) * np.abs(pen))
            probe["_margin"] = np.minimum(probe.get("_margin", np.inf), m_i)
            probe["_mask"] = probe.get("_mask", 0) + (Fz > 0).astype(np.int32) * (1 << i)
        if front:
            hb = np.stack([cs, sn, np.zeros_like(cs)], -1)
        else:
            hb = np.tile(f32([1, 0, 0]), (x.shape[0], 1))
        hw = np.einsum("nij,nj->ni", R, hb).astype(F)
        t = hw - (hw * nrm).sum(-1, keepdims=True) * nrm
        tx = t / np.sqrt(np.maximum((t * t).sum(-1, keepdims=True), F(1e-6)))      # floored: a heading parallel to the ground normal has no tangent
        ty = np.cross(nrm, tx)
        vcx = (vcp * tx).sum(-1)
        vcy = (vcp * ty).sum(-1)
        w_i = wheel[:, i]
        vden = np.maximum(F(vp.v_min), F(vp.slip_peak) * np.maximum(np.abs(vcx), np.abs(w_i * r)))
        sx = (w_i * r - vcx) / vden
        sy = -vcy / vden
        sig = np.sqrt(sx * sx + sy * sy)
        sig_safe = np.maximum(sig, F(1))
        gq = np.where(sig <= 1, mu_s * (F(2) - sig), (mu_d + (mu_s - mu_d) / sig_safe) / sig_safe)
        K = Fz * gq / vden
        # explicit-stepping stability cap: the tyre may not take out more than half of this wheel's share of the
        # body's momentum per sub-step (binds only for high-friction tasks at low speed; never in the drift task)
        if not implicit:
            K = np.minimum(K, F(0.125) * mass / h).astype(F)
        driven = (vp.drive == 1) or (not front)
        d = damp if driven else np.zeros_like(damp)
        wt = wheel_target[:, i]
        lo, hi = dc_motor_limits(w_i, vp)
        A = Iw / h + bw + K * r * r
        rhs0 = Iw * w_i / h + r * K * vcx
        # tau = clip(d (wt - w'), lo, hi) is non-increasing in w': the solution is the unclipped root w_u clamped to the roots of the
        # two constant-torque equations -- a median, never the difference d (wt - w_u) itself, whose rounding the wheel equation
        # amplifies by d / A (3e3 in the drift task, 1e5 with the elevation task's servo damping of 1000)
        w_u = (rhs0 + d * wt) / (A + d)
        w_n = np.clip(w_u, (rhs0 + lo) / A, (rhs0 + hi) / A)
        Fx = K * (w_n * r - vcx)
        Fy = -K * vcy
        Fmax = mu_s * Fz
        mag2 = Fx * Fx + Fy * Fy
        over = mag2 > Fmax * Fmax
        scale = np.where(over, Fmax / np.sqrt(np.maximum(mag2, F(1e-30))), F(1)).astype(F)
        Fx, Fy = Fx * scale, Fy * scale
        # saturated branch: redo the wheel with the force that is actually applied (Newton's third law)
        A2 = Iw / h + bw
        rhs2 = Iw * w_i / h - r * Fx
        w_u2 = (rhs2 + d * wt) / (A2 + d)
        w_n = np.where(over, np.clip(w_u2, (rhs2 + lo) / A2, (rhs2 + hi) / A2), w_n)
        new_wheel[:, i] = w_n
        if implicit:
            # damping of this wheel's force against the contact-point velocity (secant): K_s = K * scale (on the friction
            # circle the force no longer grows with the slip) laterally; longitudinally K_s with the wheel spin eliminated:
            # K_s A0 / (A0 + K r^2) for a free-spinning wheel (it follows the ground), K_s (A0 + d) / (A0 + d + K r^2) for a
            # velocity-servoed one (it resists).  Between the two by s: 1 while the torque the unclipped root asks for,
            # t_eq = A w_u - rhs0, is inside the DC-motor window, fading to 0 over the last tenth of motor_limit before the
            # window clips it.  The RATIO kx / ky is blended, not the damping d: with d ~ 1000 >> A a blend of d is a 0 / 1
            # switch in all but name, and two arithmetics that disagree in the last bit at the window's edge would take
            # different Jacobians.  Rotated into the body axes by the wheel's heading (hc, hs) and moved to the CoM by the
            # wheel's NOMINAL position (px, py): the in-plane Jacobian of a car standing on its wheels.
            t_eq = A * w_u - rhs0
            s_m = np.clip(np.minimum(hi - t_eq, t_eq - lo) * (F(10) / F(vp.motor_limit)), F(0), F(1))
            ky = (K * scale).astype(F)
            Kr2 = K * r * r
            kx = (ky * (F(1) - Kr2 * (F(1) / A + s_m * (F(1) / (A + d) - F(1) / A)))).astype(F)
            hc, hs = hb[:, 0], hb[:, 1]
            kxb = kx * hc * hc + ky * hs * hs
            kyb = kx * hs * hs + ky * hc * hc
            kxy = (kx - ky) * hc * hs
            gxw = -py * kxb + px * kxy
            gyw = px * kyb - py * kxy
            S["xx"] += kxb
            S["yy"] += kyb
            S["xy"] += kxy
            S["xw"] += gxw
            S["yw"] += gyw
            S["ww"] += -py * gxw + px * gyw
            S["nc"] += (Fz > 0).astype(F)
        Fi = Fz[:, None] * nrm + Fx[:, None] * tx + Fy[:, None] * ty
        Ftot += Fi
        Ttot += np.cross(arm, Fi)
    Ftot[:, 2] -= mass * F(vp.gravity)
    Ib = mass[:, None] * f32([vp.gyr_x ** 2, vp.gyr_y ** 2, vp.gyr_z ** 2])[None, :]
    Tb = np.einsum("nji,nj->ni", R, Ttot).astype(F)
    if implicit:
        v, wb = implicit_body_update(v, wb, R, Ftot, Tb, Ib, mass, S, vp, h)
    else:
        v = (v + h * Ftot / mass[:, None]).astype(F)
        wb = (wb + h * (Tb - np.cross(wb, Ib * wb)) / Ib).astype(F)
    ww = np.einsum("nij,nj->ni", R, wb).astype(F)
    x = (x + h * v).astype(F)
    qw, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    hh = F(0.5) * h
    dq = np.stack([
        -ww[:, 0] * qx - ww[:, 1] * qy - ww[:, 2] * qz,
        ww[:, 0] * qw + ww[:, 1] * qz - ww[:, 2] * qy,
        -ww[:, 0] * qz + ww[:, 1] * qw + ww[:, 2] * qx,
        ww[:, 0] * qy - ww[:, 1] * qx + ww[:, 2] * qw], -1)
    q = q + hh * dq
    q = (q / np.sqrt((q * q).sum(-1, keepdims=True))).astype(F)
    if probe is not None:
        probe.setdefault("contact", []).append(probe.pop("_mask"))
        probe["fz_margin"] = np.minimum(probe.get("fz_margin", np.inf), probe.pop("_margin"))
    return x, q, v, wb, new_wheel.astype(F), th, om
