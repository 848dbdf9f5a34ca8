"""Rigid-body + 4-tyre vehicle integrator, numpy float32.  DESIGNED MODEL (parity unpinned): replaces the PhysX
articulation the reference steps at mushr_drift_env_cfg.py:393-394 (dt 0.005 x decimation 4).  PhysX is closed
source and the robot USDs are missing (.MISSING_LARGE_BLOBS), so this file is the executable specification that
the HIP kernel (wheeledlab_amd/csrc/wl_vehicle.h) must match to fp32 tolerance; DESIGN.md section 4 derives it.

Reference-owned constants used here: actuator gains / limits (wheeledlab_assets/wheeledlab_assets/hound.py:4-52),
geometry (wheeledlab_tasks/common/actions.py:17-20), friction + "multiply" combine (mushr_drift_env_cfg.py:45-50,
98-109).

State per env: x (CoM, world), q (wxyz), v (CoM lin vel, world), w_b (ang vel, body), wheel spin [4] in order
bl, br, fl, fr, steer angle + rate.
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


def substep(x, q, v, wb, wheel, th, om, steer_target, wheel_target, mass, mu_s_w, mu_d_w, damp, vp, h, ground=flat_ground):
    """one integrator sub-step of length h.  Arrays are [N,...] float32.  Returns the new state tuple."""
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
        Fz = np.where(pen > 0, np.maximum(F(vp.susp_k) * pen - F(vp.susp_c) * vn, F(0)), F(0)).astype(F)
        if front:
            hb = np.stack([cs, sn, np.zeros_like(cs)], -1)
        else:
            hb = np.tile(f32([1, 0, 0]), (x.shape[0], 1))
        hw = np.einsum("nij,nj->ni", R, hb).astype(F)
        t = hw - (hw * nrm).sum(-1, keepdims=True) * nrm
        tx = t / np.sqrt((t * t).sum(-1, keepdims=True))
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
        K = np.minimum(K, F(0.125) * mass / h).astype(F)
        driven = (vp.drive == 1) or (not front)
        d = damp if driven else np.zeros_like(damp)
        wt = wheel_target[:, i]
        lo, hi = dc_motor_limits(w_i, vp)
        A = Iw / h + bw + K * r * r
        rhs0 = Iw * w_i / h + r * K * vcx
        w_u = (rhs0 + d * wt) / (A + d)
        # unclipped motor: take w_u itself (tau = d (wt - w_u) cancels catastrophically near the target)
        tau_u = d * (wt - w_u)
        tau = np.clip(tau_u, lo, hi)
        w_n = np.where(tau == tau_u, w_u, (rhs0 + tau) / A)
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
        tau_u2 = d * (wt - w_u2)
        tau2 = np.clip(tau_u2, lo, hi)
        w_n = np.where(over, np.where(tau2 == tau_u2, w_u2, (rhs2 + tau2) / A2), w_n)
        new_wheel[:, i] = w_n
        Fi = Fz[:, None] * nrm + Fx[:, None] * tx + Fy[:, None] * ty
        Ftot += Fi
        Ttot += np.cross(arm, Fi)
    Ftot[:, 2] -= mass * F(vp.gravity)
    v = (v + h * Ftot / mass[:, None]).astype(F)
    Ib = mass[:, None] * f32([vp.gyr_x ** 2, vp.gyr_y ** 2, vp.gyr_z ** 2])[None, :]
    Tb = np.einsum("nji,nj->ni", R, Ttot).astype(F)
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
    return x, q, v, wb, new_wheel.astype(F), th, om
