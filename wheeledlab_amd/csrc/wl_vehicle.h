// wl_vehicle.h -- rigid body + 4 tyre contacts, one integrator sub-step, all state in registers.
// Replaces the PhysX articulation step of the reference (mushr_drift_env_cfg.py:393-394); model derivation in
// DESIGN.md section 4, executable spec in oracle/vehicle.py.  Actuator constants: wheeledlab_assets/hound.py:4-52.
//
// Two integrators of the same force laws (template parameter IMPL, WlVehicleParams.implicit):
//   explicit  -- semi-implicit Euler of the body under the wheel forces at the current state; h <= 5 ms and a cap on the tyre's
//                secant stiffness.  The drift tasks (sim.dt = 5 ms, mushr_drift_env_cfg.py:393-394).
//   implicit  -- round 6: linearly implicit (Rosenbrock-W).  The body's velocity increment solves (M + h G^) du = h r(u_n):
//                r the body-frame residual, G^ the contact forces' damping matrix approximated by its exact in-plane 3 x 3 block
//                (v_x, v_y, w_z) for the nominal wheel positions and its diagonal for heave / roll / pitch.  First-order
//                consistent for any G^, every steady state of the force laws is a fixed point whatever h is, no stiffness cap:
//                ONE sub-step per sim.dt at the reference's own physics rate -- 10 ms elevation, 20 ms visual
//                (mushr_elevation_env_cfg.py:461-462, mushr_visual_env_cfg.py:435-436; PhysX: implicit TGS, mushr.py:22-36).
#pragma once
#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"

struct VehState {
    V3 x;        // CoM position, world
    Quat q;      // body -> world
    V3 v;        // CoM linear velocity, world
    V3 wb;       // angular velocity, body frame
    float wheel[4];  // spin, order bl, br, fl, fr
    float th, om;    // steer angle / rate
};

struct EnvConst {   // per-env constants hoisted out of the sub-step loop (VGPRs that live through it: keep them few)
    float h_inv_mass;          // h / m   (inertia = m * gyr^2: everything else about the mass is uniform, see VehDerived)
    float K_cap;               // 0.125 m / h
    float mu_s, mu_d, mu_sd;   // combined (wheel x ground) friction: static, dynamic, static - dynamic
    float damp;                // throttle damping of driven wheels
    float inv_A0_damp;         // 1 / (Iw/h + bearing damping + damp): the saturated-branch wheel solve of a driven wheel
    float steer_target;
    float wheel_target[4];
    // quad form: THIS lane's wheel (picked once per env-step, not per sub-step): velocity target, body position,
    // throttle damping (0: undriven), 1 / (A0 + that damping)
    float wt_lane, bx_lane, by_lane, d_lane, inv_A0d_lane;
    float lx_lane, ly_lane;    // (bx_lane, by_lane) / gyr_z
};


// Uniform constants derived from WlVehicleParams + dt.  gfx950 has no scalar float ALU: anything computed from
// uniform parameters inside the kernel lands in VGPRs and, being loop-invariant, stays there through every sub-step.
// The C-ABI wrappers compute these once on the host instead, so in-kernel they are SGPR operands.
struct VehDerived {
    float h, inv_h, half_h;                 // sub-step length
    float steer_a, steer_b;                 // h kp / J ;  1 / (1 + h kd / J + h^2 kp / J)
    float steer_J_h, steer_h_J;             // J / h ; h / J
    float zrel;                             // wheel centre z relative to the CoM (body frame)
    float Iw_h, A0, inv_A0;                 // wheel inertia / h ; Iw_h + bearing damping ; 1 / that
    float hg;                               // h * gravity
    float inv_g2x, inv_g2y, inv_g2z;        // 1 / gyr^2: h / I = (h / m) * inv_g2
    float cgx, cgy, cgz;                    // gyroscopic coefficients h (Iz - Iy) / Ix, ... : the mass cancels
    float inv_wlim, mot_b;                  // 1 / motor_vel_limit ; motor_sat / motor_vel_limit
    float mot_g;                            // 10 / motor_limit: the implicit integrator's fade of the motor damping at the window's edge
    float r2;                               // wheel radius squared
    // implicit integrator (oracle/vehicle.py::implicit_body_update)
    float Dn;                               // c + h k: the normal spring-damper's damping with the position update folded in
    float az2, bx2, by2;                    // squared nominal levers: CoM height above the contact patches, half wheelbase, half track
    float gz, inv_gz;                       // gyr_z: the in-plane block is solved in (dv_x, dv_y, gz dw_z) -- symmetric, mass-free
    float lxf, lxr, ly;                     // nominal wheel positions / gz: front x, rear x (negative), left y
    float iso_x, iso_y, iso_z;              // rho^2 / gyr^2 per axis, rho the longest lever: the tilted car's isotropic bound on the rotations
    int32_t n_sub;                          // decimation * substeps
};

inline VehDerived derive_vehicle(const WlVehicleParams& vp, float sim_dt, int decimation) {
    VehDerived d;
    d.h = sim_dt / (float)vp.substeps;
    d.inv_h = 1.f / d.h;
    d.half_h = 0.5f * d.h;
    const float invJ = 1.f / vp.steer_inertia;
    d.steer_a = d.h * vp.steer_kp * invJ;
    d.steer_b = 1.f / (1.f + d.h * vp.steer_kd * invJ + d.h * d.h * vp.steer_kp * invJ);
    d.steer_J_h = vp.steer_inertia * d.inv_h;
    d.steer_h_J = d.h * invJ;
    d.zrel = vp.wheel_z - vp.cg_z;
    d.Iw_h = vp.wheel_inertia * d.inv_h;
    d.A0 = d.Iw_h + vp.wheel_damping;
    d.inv_A0 = 1.f / d.A0;
    d.hg = d.h * vp.gravity;
    const float gx2 = vp.gyr_x * vp.gyr_x, gy2 = vp.gyr_y * vp.gyr_y, gz2 = vp.gyr_z * vp.gyr_z;
    d.inv_g2x = 1.f / gx2;
    d.inv_g2y = 1.f / gy2;
    d.inv_g2z = 1.f / gz2;
    d.cgx = d.h * (gz2 - gy2) / gx2;
    d.cgy = d.h * (gx2 - gz2) / gy2;
    d.cgz = d.h * (gy2 - gx2) / gz2;
    d.inv_wlim = 1.f / vp.motor_vel_limit;
    d.mot_b = vp.motor_sat / vp.motor_vel_limit;
    d.mot_g = 10.f / vp.motor_limit;
    d.r2 = vp.wheel_radius * vp.wheel_radius;
    d.Dn = vp.susp_c + d.h * vp.susp_k;
    const float az = d.zrel - vp.wheel_radius;
    d.az2 = az * az;
    d.bx2 = 0.5f * (vp.half_wheelbase_f * vp.half_wheelbase_f + vp.half_wheelbase_r * vp.half_wheelbase_r);
    d.by2 = vp.half_track * vp.half_track;
    d.gz = vp.gyr_z;
    d.inv_gz = 1.f / vp.gyr_z;
    d.lxf = vp.half_wheelbase_f * d.inv_gz;
    d.lxr = -vp.half_wheelbase_r * d.inv_gz;
    d.ly = vp.half_track * d.inv_gz;
    const float rho2 = d.bx2 + d.by2 + d.az2;
    d.iso_x = rho2 * d.inv_g2x, d.iso_y = rho2 * d.inv_g2y, d.iso_z = rho2 * d.inv_g2z;
    d.n_sub = decimation * vp.substeps;
    return d;
}

// per-env constants from the env's randomisation rows (wheel friction, throttle damping, mass)
WL_DEV void env_const_rows(EnvConst& ec, const WlVehicleParams& vp, const VehDerived& vd, float mass, float mu_s_wheel,
                           float mu_d_wheel, float damp) {
    ec.h_inv_mass = vd.h * rcp(mass);
    ec.K_cap = 0.125f * mass * vd.inv_h;
    ec.mu_s = mu_s_wheel * vp.ground_mu_s;
    ec.mu_d = fminf(mu_d_wheel * vp.ground_mu_d, ec.mu_s);
    ec.mu_sd = ec.mu_s - ec.mu_d;
    ec.damp = damp;
    ec.inv_A0_damp = rcp(vd.A0 + damp);
}

struct FlatGround {
    static constexpr bool kFlat = true;   // n == (0, 0, 1), zg == 0 everywhere
    WL_DEV void sample(float, float, float& zg, V3& n) const {
        zg = 0.f;
        n = v3(0.f, 0.f, 1.f);
    }
    template <int W>     // W: the wheel slot of a lane (samplers that keep per-wheel state, wl_heightfield.h)
    WL_DEV void sample_wheel(float x, float y, float& zg, V3& n) const { sample(x, y, zg, n); }
};

// by-value pick of this lane's element: the operands are SSA values, so the selection is three v_cndmask.  (Written as
// a ternary chain over array elements the compiler folds it into a wid-indexed load of the array, which then lives in
// scratch memory -- a store -> load round trip on the critical tail of the step.)
WL_DEV float quad_pick(int wid, float a, float b, float c, float d) { return wid == 0 ? a : wid == 1 ? b : wid == 2 ? c : d; }

// per-lane constants of the quad form (lane `wid` owns wheel `wid`): picked once per env-step, not per sub-step
WL_DEV void env_const_lane(EnvConst& ec, const WlVehicleParams& vp, const VehDerived& vd, int wid) {
    const bool front = wid >= 2, left = (wid & 1) == 0;
    const bool driven = (vp.drive == 1) || !front;
    ec.wt_lane = quad_pick(wid, ec.wheel_target[0], ec.wheel_target[1], ec.wheel_target[2], ec.wheel_target[3]);
    ec.bx_lane = front ? vp.half_wheelbase_f : -vp.half_wheelbase_r;
    ec.by_lane = left ? vp.half_track : -vp.half_track;
    ec.d_lane = driven ? ec.damp : 0.f;
    ec.inv_A0d_lane = driven ? ec.inv_A0_damp : vd.inv_A0;
    ec.lx_lane = front ? vd.lxf : vd.lxr;
    ec.ly_lane = left ? vd.ly : -vd.ly;
}

// ---- one wheel: contact + tyre + wheel-spin solve, everything in the BODY frame ------------------------------------------
// Round 1 worked in the world frame: per wheel and sub-step R pb, ww x arm, a tangent frame rebuilt from R h, arm x F and
// finally R^T T -- 3 559 VALU instructions per env-step, which (not HBM) bound the lane form (profiles/r01_pmc_sq).  In
// the body frame the arm of wheel i is pb_i - r n with n the ground normal in body coordinates (on flat ground: the
// third ROW of R, shared by all wheels), the heading of an axle is h = (cos th, sin th, 0), and
//     tx = (h - g n) / |h - g n|,  ty = n x tx = (n x h) / |h - g n|,   g = n . h,  |h - g n|^2 = 1 - g^2
// so that the slip velocities need no frame at all:
//     v_cx = (h . vc - g vn) it,   v_cy = ((n x h) . vc) it,   it = rsq(1 - g^2),   vn = n . vc
// and the force comes back as F = fx h + fy (n x h) + (Fz - fx g) n with fx = Fx it, fy = Fy it.  Torques are summed in
// the body frame (no R^T T), the quaternion is advanced with the body rate (q (0, w_b) == (0, R w_b) q: no R w_b), and
// on flat ground the world z force is the plain sum of the normal loads.  Same model, same numbers to rounding
// (spec: oracle/vehicle.py::substep, which keeps the world-frame form), ~40 % fewer instructions per sub-step.
//   n: ground normal at the wheel (body frame); vc: contact-point velocity (body frame); pen: penetration along n_w
//   STEER: the wheel's heading in the body frame is (hc, hs, 0) = (cos th, sin th, 0); !STEER: (1, 0, 0) with every
//          product by the constant components spelled out as absent (the compiler must keep `x * 0`: x could be inf)
//   MOTOR: the wheel may be driven: d = throttle damping (0: undriven), inv_A0d = 1 / (A0 + d);
//          !MOTOR: known to be undriven (d == 0): tau == 0 and the DC-motor window drops out
struct TyreCoef {   // the contact force on the body (body frame) is F = fx h + fy (n x h) + kz n
    float fx, fy, kz;
    float Fz;       // normal load
    float kx, ky;   // IMPL: the force's damping against the contact-point velocity, along / across the wheel (secant)
    float dk;       // kx - ky
};
// F for coefficients c -- LINEAR in (fx, fy, kz): the axle form below maps the sum and the difference of its two wheels
template <bool STEER>
WL_DEV V3 tyre_force(float fx, float fy, float kz, V3 n, float hc, float hs) {
    const float fyz = fy * n.z;   // (n x h) = (-n.z hs, n.z hc, n.x hs - n.y hc)
    if constexpr (STEER) return v3(fmaf(kz, n.x, fmaf(fx, hc, -fyz * hs)), fmaf(kz, n.y, fmaf(fx, hs, fyz * hc)),
                                   fmaf(kz, n.z, fy * fmaf(n.x, hs, -n.y * hc)));
    else return v3(fmaf(kz, n.x, fx), fmaf(kz, n.y, fyz), fmaf(kz, n.z, -fy * n.y));
}
//   IMPL:  no stiffness cap; also returns the damping (kx, ky) of the force against the contact-point velocity: ky = K * scale
//          (scale: the friction circle's), kx = ky (1 - K r^2 cw) -- the wheel spin eliminated; cw the wheel's compliance,
//          1 / (A + d) while the motor servoes it, 1 / A while the DC-motor window clips the torque, blended over the last tenth
//          of motor_limit before the clip (continuous: a 0 / 1 switch would hand two arithmetics different Jacobians at the edge)
template <bool STEER, bool MOTOR, bool IMPL = false>
WL_DEV TyreCoef wheel_tyre(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, V3 n, V3 vc, float pen, float hc,
                           float hs, float d, float inv_A0d, float wt, float& w_spin) {
    const float r = vp.wheel_radius;
    const float vn = dot(n, vc);
    const float Fz = pen > 0.f ? fmaxf(fmaf(-vp.susp_c, vn, fminf(vp.susp_k * pen, vp.susp_fmax)), 0.f) : 0.f;   // (spring capped: WlVehicleParams.susp_fmax)
    // g = n . h ; m = (n x h).z ; lx = h . vc ; ly = (hc vc.y - hs vc.x)  [(n x h).xy = n.z (-hs, hc)]
    const float g = STEER ? fmaf(n.x, hc, n.y * hs) : n.x;
    const float m = STEER ? fmaf(n.x, hs, -n.y * hc) : -n.y;
    const float lx = STEER ? fmaf(hc, vc.x, hs * vc.y) : vc.x;
    const float ly = STEER ? fmaf(hc, vc.y, -hs * vc.x) : vc.y;
    // (floored: a wheel whose heading is parallel to the ground normal -- a car standing on its nose -- has 1 - g^2 = 0, or a rounding
    // below it, and rsq of that is inf / NaN: one of the ways a tumbling car left fp32)
    const float it = rsq(fmaxf(fmaf(-g, g, 1.f), 1e-6f));
    const float vcx = it * fmaf(-g, vn, lx);
    const float vcy = it * fmaf(n.z, ly, m * vc.z);
    const float w_i = w_spin;
    const float wr = w_i * r;
    const float vden = fmaxf(vp.v_min, vp.slip_peak * fmaxf(fabsf(vcx), fabsf(wr)));
    const float inv_vden = rcp(vden);
    const float sx = (wr - vcx) * inv_vden, sy = vcy * inv_vden;   // (sy's sign is immaterial: only sy^2 is used)
    // sig = |s| and 1 / max(sig, 1) from ONE rsq (sqrt + rcp were two quarter-rate instructions)
    const float s2 = fmaxf(fmaf(sx, sx, sy * sy), 1e-30f);
    const float rs = rsq(s2);
    const float sig = s2 * rs;
    // g(sig) / sig, branch-free: below the peak (sig <= 1, inv_sig == 1) the first term is mu_s and the second adds
    // mu_s (1 - sig) -> mu_s (2 - sig); above it the second term vanishes.  (As a ?: the two short arms become
    // divergent control flow: five exec-mask instructions around six arithmetic ones, every wheel, every sub-step.)
    const float inv_sig = fminf(rs, 1.f);
    const float gq = fmaf(ec.mu_s, fmaxf(1.f - sig, 0.f), fmaf(ec.mu_sd, inv_sig, ec.mu_d) * inv_sig);
    // explicit-stepping stability cap: at most half of this wheel's share of the body momentum per sub-step
    const float K = IMPL ? Fz * gq * inv_vden : fminf(Fz * gq * inv_vden, ec.K_cap);
    // implicit spin update with the tyre's secant stiffness (unconditionally stable): A w = rhs0 + tau(w) with the
    // DC-motor torque tau = clamp(d (wt - w), lo, hi), window [lo, hi] taken at the current spin (IsaacLab DCMotor,
    // hound.py:13-21).  tau is non-increasing in w, so the solution is the unclipped root w_u = (rhs0 + d wt) / (A + d)
    // clamped to the roots of the two constant-torque equations: ONE median instead of torque -> clamp -> compare ->
    // select (and never the cancelling difference d (wt - w_u), whose rounding d / A ~ 3e3 amplifies).
    const float A = fmaf(K, vd.r2, vd.A0);
    const float rK = r * K;
    const float rhs0 = fmaf(vd.Iw_h, w_i, rK * vcx);
    const float inv_A = rcp(A);
    float w_n, tau_lo = 0.f, tau_hi = 0.f;
    float cw = inv_A;   // IMPL: the wheel's compliance 1 / (A + d'): inv_A free-spinning, 1 / (A + d) servoed, blended by the window's fade
    if constexpr (MOTOR) {
        tau_hi = clampf(fmaf(-vd.mot_b, w_i, vp.motor_sat), 0.f, vp.motor_limit);     // sat (1 - w / w_lim)
        tau_lo = clampf(fmaf(-vd.mot_b, w_i, -vp.motor_sat), -vp.motor_limit, 0.f);   // sat (-1 - w / w_lim)
        const float inv_Ad = rcp(A + d);
        const float w_u = fmaf(d, wt, rhs0) * inv_Ad;
        w_n = clampf(w_u, (rhs0 + tau_lo) * inv_A, (rhs0 + tau_hi) * inv_A);
        if constexpr (IMPL) {   // s = clamp(min(hi - t_eq, t_eq - lo) * 10 / motor_limit, 0, 1), t_eq = A w_u - rhs0
            const float m_hi = fmaf(-A, w_u, rhs0 + tau_hi), m_lo = fmaf(A, w_u, -(rhs0 + tau_lo));
            cw = fmaf(clampf(fminf(m_hi, m_lo) * vd.mot_g, 0.f, 1.f), inv_Ad - inv_A, inv_A);
        }
    } else {
        w_n = rhs0 * inv_A;
    }
    float ks = K;
    float Fx = fmaf(w_n, rK, -K * vcx);
    float Fy = -K * vcy;
    const float Fmax = ec.mu_s * Fz;
    const float mag2 = fmaf(Fx, Fx, Fy * Fy);
    if (mag2 > Fmax * Fmax) {   // friction-circle saturation: re-solve the wheel against the force actually applied
        const float scale = Fmax * rsq(fmaxf(mag2, 1e-30f));
        Fx *= scale;
        Fy *= scale;
        if constexpr (IMPL) ks = K * scale;
        const float rhs2 = fmaf(vd.Iw_h, w_i, -r * Fx);
        if constexpr (MOTOR) {
            const float w_u2 = fmaf(d, wt, rhs2) * inv_A0d;   // 1 / (A0 + d): per-env constant
            w_n = clampf(w_u2, (rhs2 + tau_lo) * vd.inv_A0, (rhs2 + tau_hi) * vd.inv_A0);
        } else {
            w_n = rhs2 * vd.inv_A0;
        }
    }
    w_spin = w_n;
    // F = fx h + fy (n x h) + (Fz - fx g) n  with fx = Fx it, fy = Fy it
    const float fx = Fx * it;
    // kx / ky - 1 = -K r^2 x (the wheel's compliance)
    const float kxr1 = IMPL ? -(K * vd.r2) * cw : 0.f;
    return TyreCoef{fx, Fy * it, fmaf(-fx, g, Fz), Fz, fmaf(ks, kxr1, ks), ks, ks * kxr1};
}

// contact kinematics of one wheel at body position (bx, by, zrel)
struct Contact {
    V3 n;       // ground normal, body frame
    V3 arm;     // CoM -> contact point, body frame
    V3 vc;      // velocity of the contact point, body frame
    float pen;
};
// On flat ground the sub-expressions shared by the wheels of an axle / a side (arm components, the partial sums of
// vb + w_b x arm, the penetration) are written so that the compiler's CSE merges them across the inlined calls.
template <class Ground, int W = 0>
WL_DEV Contact wheel_contact(const WlVehicleParams& vp, const VehDerived& vd, const Ground& ground, const Mat3& R, const VehState& s,
                             V3 vb, float bx, float by) {
    const float r = vp.wheel_radius;
    Contact c;
    if constexpr (Ground::kFlat) {
        c.n = R.r2;   // world +z in body coordinates
        c.pen = fmaf(-c.n.y, by, fmaf(-c.n.x, bx, fmaf(-c.n.z, vd.zrel, r - s.x.z)));   // r - (x.z + r2 . pb)
    } else {
        const float cx = s.x.x + fmaf(R.r0.x, bx, fmaf(R.r0.y, by, R.r0.z * vd.zrel));
        const float cy = s.x.y + fmaf(R.r1.x, bx, fmaf(R.r1.y, by, R.r1.z * vd.zrel));
        const float cz = s.x.z + fmaf(R.r2.x, bx, fmaf(R.r2.y, by, R.r2.z * vd.zrel));
        float zg;
        V3 nw;
        ground.template sample_wheel<W>(cx, cy, zg, nw);
        c.pen = r - (cz - zg) * nw.z;
        c.n = mul_t(R, nw);
    }
    c.arm = v3(fmaf(-r, c.n.x, bx), fmaf(-r, c.n.y, by), fmaf(-r, c.n.z, vd.zrel));
    c.vc = v3(fmaf(-s.wb.z, c.arm.y, fmaf(s.wb.y, c.arm.z, vb.x)), fmaf(s.wb.z, c.arm.x, fmaf(-s.wb.x, c.arm.z, vb.y)),
              fmaf(-s.wb.y, c.arm.x, fmaf(s.wb.x, c.arm.y, vb.z)));
    return c;
}

// force / torque / load accumulators of one sub-step (body frame)
struct Wrench {
    V3 F, T;
    float Fz;
};
// IMPL: the in-plane damping matrix of the wheels' forces about the CoM, in (v_x, v_y, gz w_z) -- sums over the wheels of
// [kxb, kxy; kxy, kyb] (a wheel's (kx, ky) rotated into the body axes by its heading) moved to the CoM by the wheel's NOMINAL
// position (lx, ly) = (x, y) / gz -- and the number of wheels in contact (oracle/vehicle.py::substep, the S sums)
struct Jac {
    float xx, yy, xy, xw, yw, ww, nc;
};
template <bool STEER, bool FIRST>
WL_DEV void jac_add(Jac& J, const TyreCoef& o, float hc, float hs, float lx, float ly) {
    float kxb, kyb, kxy;
    if constexpr (STEER) {
        const float c2 = hc * hc, s2 = hs * hs;
        kxb = fmaf(o.kx, c2, o.ky * s2);
        kyb = fmaf(o.kx, s2, o.ky * c2);
        // (an fma, not a product: a bare product feeding quad_sum's first add is contracted into it at some inlining sites and not
        // at others -- measured as last-bit differences between the persistent and the stepping kernels)
        kxy = fmaf(o.dk, hc * hs, 0.f);
    } else {
        kxb = o.kx, kyb = o.ky, kxy = 0.f;
    }
    const float gxw = STEER ? fmaf(lx, kxy, -ly * kxb) : -ly * kxb;
    const float gyw = STEER ? fmaf(lx, kyb, -ly * kxy) : lx * kyb;
    const float gww = fmaf(lx, gyw, -ly * gxw);
    const float c = o.Fz > 0.f ? 1.f : 0.f;
    if constexpr (FIRST) {
        J = Jac{kxb, kyb, kxy, gxw, gyw, gww, c};
    } else {
        J.xx += kxb, J.yy += kyb, J.xy += kxy, J.xw += gxw, J.yw += gyw, J.ww += gww, J.nc += c;
    }
}
template <class Ground, bool STEER, bool MOTOR, bool FIRST, int W = 0, bool IMPL = false>
WL_DEV void wheel_step(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, const Ground& ground, const Mat3& R,
                       const VehState& s, V3 vb, float bx, float by, float hc, float hs, float d, float inv_A0d, float wt,
                       float& w_spin, Wrench& w, Jac* J = nullptr, float lx = 0.f, float ly = 0.f) {
    const Contact c = wheel_contact<Ground, W>(vp, vd, ground, R, s, vb, bx, by);
    const TyreCoef o = wheel_tyre<STEER, MOTOR, IMPL>(vp, vd, ec, c.n, c.vc, c.pen, hc, hs, d, inv_A0d, wt, w_spin);
    const V3 F = tyre_force<STEER>(o.fx, o.fy, o.kz, c.n, hc, hs);
    const V3 t = cross(c.arm, F);
    if constexpr (IMPL) jac_add<STEER, FIRST>(*J, o, hc, hs, lx, ly);
    if constexpr (FIRST) {   // plain assignment: `0 + x` cannot be folded (-0), it would cost an instruction per component
        w.F = F;
        w.T = t;
        w.Fz = o.Fz;
    } else {
        w.F = w.F + F;
        w.T = w.T + t;
        w.Fz += o.Fz;
    }
}

// The two wheels of an axle on FLAT ground (lane form): they share n, the heading and arm.x / arm.z, and F is linear
// in the tyre coefficients -- so the axle's force is the map of the coefficient SUMS, and all the torque needs beyond
// that is the map of their DIFFERENCES (x and z components): sum_i arm_i x F_i with arm.y = +-ht - r n.y.
struct AxleOut {
    V3 F;          // force of both wheels
    float dX, dZ;  // F_left - F_right, x and z components
    float ax;      // arm.x of the axle
    float Fz;      // load of both wheels
};
template <bool STEER, bool MOTOR, bool IMPL = false, bool FIRST = true>
WL_DEV AxleOut axle_step(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, const Mat3& R, const VehState& s, V3 vb,
                         float bx, float hc, float hs, float d, float inv_A0d, float wt_l, float wt_r, float& w_l, float& w_r,
                         Jac* J = nullptr, float lx = 0.f) {
    const FlatGround flat{};
    const Contact cl = wheel_contact(vp, vd, flat, R, s, vb, bx, vp.half_track);
    const Contact cr = wheel_contact(vp, vd, flat, R, s, vb, bx, -vp.half_track);
    const TyreCoef a = wheel_tyre<STEER, MOTOR, IMPL>(vp, vd, ec, cl.n, cl.vc, cl.pen, hc, hs, d, inv_A0d, wt_l, w_l);
    const TyreCoef b = wheel_tyre<STEER, MOTOR, IMPL>(vp, vd, ec, cr.n, cr.vc, cr.pen, hc, hs, d, inv_A0d, wt_r, w_r);
    if constexpr (IMPL) {
        jac_add<STEER, FIRST>(*J, a, hc, hs, lx, vd.ly);
        jac_add<STEER, false>(*J, b, hc, hs, lx, -vd.ly);
    }
    AxleOut o;
    o.F = tyre_force<STEER>(a.fx + b.fx, a.fy + b.fy, a.kz + b.kz, cl.n, hc, hs);
    const V3 D = tyre_force<STEER>(a.fx - b.fx, a.fy - b.fy, a.kz - b.kz, cl.n, hc, hs);   // (the y component is dead code)
    o.dX = D.x;
    o.dZ = D.z;
    o.ax = cl.arm.x;
    o.Fz = a.Fz + b.Fz;
    return o;
}

// steering: implicit PD drive, effort- and rate-limited (hound.py:5-12)
WL_DEV void steer_update(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, VehState& s) {
    const float e = ec.steer_target - s.th;
    float om_n = fmaf(vd.steer_a, e, s.om) * vd.steer_b;
    const float tau = clampf(vd.steer_J_h * (om_n - s.om), -vp.steer_effort, vp.steer_effort);
    om_n = clampf(fmaf(vd.steer_h_J, tau, s.om), -vp.steer_vel_limit, vp.steer_vel_limit);
    s.th = fmaf(vd.h, om_n, s.th);
    s.om = om_n;
}

// the pose half of a sub-step: x <- x + h v, q <- q + (h / 2) q (0, w_b) renormalised, both with the NEW velocities
WL_DEV void pose_integrate(const VehDerived& vd, VehState& s) {
    s.x = fma3(vd.h, s.v, s.x);
    // q <- q + (h / 2) q (0, w_b)   [== (h / 2) (0, R w_b) q, the world-rate form of the spec], then renormalise
    const V3 u = vd.half_h * s.wb;
    Quat q = s.q;
    const float nw = fmaf(-q.x, u.x, fmaf(-q.y, u.y, fmaf(-q.z, u.z, q.w)));
    const float nx = fmaf(q.w, u.x, fmaf(q.y, u.z, fmaf(-q.z, u.y, q.x)));
    const float ny = fmaf(q.w, u.y, fmaf(q.z, u.x, fmaf(-q.x, u.z, q.y)));
    const float nz = fmaf(q.w, u.z, fmaf(q.x, u.y, fmaf(-q.y, u.x, q.z)));
    const float inv_n = rsq(fmaf(nw, nw, fmaf(nx, nx, fmaf(ny, ny, nz * nz))));
    s.q = Quat{nw * inv_n, nx * inv_n, ny * inv_n, nz * inv_n};
}

// true for every lane if `p` holds in ANY lane of the wavefront (a scalar branch condition)
WL_DEV bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0; }

// semi-implicit Euler of the rigid body under the summed contact force Fb / torque Tb about the CoM (BODY frame);
// Fz_w: world z component of the contact force where the caller knows it without R (flat ground: the sum of the loads).
// The inertia tensor is m diag(gyr^2): h / I = (h / m) / gyr^2 and the gyroscopic term w x (I w) / I has the mass
// cancelled -- per-env h / m and uniform constants, no per-env inertia vectors living in registers through the loop.
template <bool FLAT>
WL_DEV void body_integrate(const VehDerived& vd, const EnvConst& ec, VehState& s, const Mat3& R, V3 Fb, V3 Tb, float Fz_w) {
    const float Fz = FLAT ? Fz_w : dot(R.r2, Fb);
    s.v = v3(fmaf(ec.h_inv_mass, dot(R.r0, Fb), s.v.x), fmaf(ec.h_inv_mass, dot(R.r1, Fb), s.v.y),
             fmaf(ec.h_inv_mass, Fz, s.v.z) - vd.hg);
    const V3 t = ec.h_inv_mass * Tb;
    const V3 w = s.wb;
    s.wb = v3(fmaf(vd.inv_g2x, t.x, fmaf(-vd.cgx, w.y * w.z, w.x)), fmaf(vd.inv_g2y, t.y, fmaf(-vd.cgy, w.z * w.x, w.y)),
              fmaf(vd.inv_g2z, t.z, fmaf(-vd.cgz, w.x * w.y, w.z)));
    pose_integrate(vd, s);
}

// The linearly implicit velocity update (oracle/vehicle.py::implicit_body_update): (M + h G^) du = h r in the body frame.
//   r: Fb + m g_b - m w x v_b  (g_b = -g R.r2: gravity in body coordinates; -w x v_b: a world-constant velocity seen from the
//      turning body, so that steady cornering is a fixed point), Tb - w x (I w);
//   G^: J (the in-plane 3 x 3 block, solved by LDL^T in (dv_x, dv_y, gz dw_z) with the mass divided out) and the diagonal for
//      heave / roll / pitch: Dn = c + h k per wheel in contact at its nominal lever, the tyres through the CoM height.
// The world velocity takes R (dv_b + h w x v_b): in free flight (J = 0) exactly the explicit update.
WL_DEV void body_integrate_implicit(const VehDerived& vd, const EnvConst& ec, VehState& s, const Mat3& R, V3 vb, V3 Fb, V3 Tb,
                                    const Jac& J) {
    // (every multiply-add spelled out: under -ffp-contract=fast the compiler picks which product of a sum it fuses per inlining
    // site, and the persistent kernels must reproduce the stepping kernels bit for bit)
    const float q = ec.h_inv_mass;
    const V3 w = s.wb;
    const V3 c = v3(fmaf(w.y, vb.z, -(w.z * vb.y)), fmaf(w.z, vb.x, -(w.x * vb.z)), fmaf(w.x, vb.y, -(w.y * vb.x)));   // w x v_b
    const V3 hrot = vd.h * c;
    // h a_b = q Fb - h g R.r2 - h w x v_b ;  h alpha = q Tb / gyr^2 - h (gyroscopic)
    const V3 ha = v3(fmaf(q, Fb.x, fmaf(-vd.hg, R.r2.x, -hrot.x)), fmaf(q, Fb.y, fmaf(-vd.hg, R.r2.y, -hrot.y)),
                     fmaf(q, Fb.z, fmaf(-vd.hg, R.r2.z, -hrot.z)));
    const V3 t = q * Tb;
    const V3 hal = v3(fmaf(vd.inv_g2x, t.x, -vd.cgx * (w.y * w.z)), fmaf(vd.inv_g2y, t.y, -vd.cgy * (w.z * w.x)),
                      fmaf(vd.inv_g2z, t.z, -vd.cgz * (w.x * w.y)));
    const float nD = J.nc * vd.Dn;
    // in-plane: LDL^T of [1 + q Jxx, q Jxy, q Jxw; ., 1 + q Jyy, q Jyw; ., ., 1 + q Jww] (identity + PSD: no pivoting); heave / roll /
    // pitch: the diagonal
    float a11 = fmaf(q, J.xx, 1.f), a12 = q * J.xy, a13 = q * J.xw, a22 = fmaf(q, J.yy, 1.f), a23 = q * J.yw, a33 = fmaf(q, J.ww, 1.f);
    float dz = fmaf(q, nD, 1.f), dx = fmaf(q * vd.inv_g2x, fmaf(vd.az2, J.yy, vd.by2 * nD), 1.f),
          dy = fmaf(q * vd.inv_g2y, fmaf(vd.az2, J.xx, vd.bx2 * nD), 1.f);
    // The nominal geometry is that of a car standing on its wheels.  Tilted by more than ~40 degrees (world up in the body frame,
    // R.r2.z < 0.75: on its side, on its roof, tumbling -- the visual task has no rollover termination) G^ falls back to an isotropic
    // bound that over-estimates the damping matrix whatever the geometry: g = 2 (sum of kx + ky + contact damping) on every
    // translation, g rho^2 on every rotation (oracle/vehicle.py::implicit_body_update).  Behind a wavefront-uniform branch: no car of
    // a wavefront is tilted in all but a few launches, and the 15 instructions of the fallback then cost one ballot.
    const bool tilted = R.r2.z < 0.75f;
    if (wave_any(tilted)) {
        const float qg = q * (2.f * (J.xx + J.yy + nD));
        const float iso = 1.f + qg;
        a11 = tilted ? iso : a11, a22 = tilted ? iso : a22, a33 = tilted ? fmaf(qg, vd.iso_z, 1.f) : a33;
        a12 = tilted ? 0.f : a12, a13 = tilted ? 0.f : a13, a23 = tilted ? 0.f : a23;
        dz = tilted ? iso : dz, dx = tilted ? fmaf(qg, vd.iso_x, 1.f) : dx, dy = tilted ? fmaf(qg, vd.iso_y, 1.f) : dy;
    }
    const float b1 = ha.x, b2 = ha.y, b3 = hal.z * vd.gz;
    const float i1 = rcp(a11);
    const float l21 = a12 * i1, l31 = a13 * i1;
    const float d2 = fmaf(-l21, a12, a22);
    const float t32 = fmaf(-l31, a12, a23);
    const float i2 = rcp(d2);
    const float l32 = t32 * i2;
    const float d3 = fmaf(-l32, t32, fmaf(-l31, a13, a33));
    const float y2 = fmaf(-l21, b1, b2);
    const float y3 = fmaf(-l32, y2, fmaf(-l31, b1, b3));
    const float x3 = y3 * rcp(d3);
    const float x2 = fmaf(y2, i2, -(l32 * x3));
    const float x1 = fmaf(b1, i1, fmaf(-l21, x2, -(l31 * x3)));
    // heave / roll / pitch: diagonal
    const float rz = rcp(dz), rx = rcp(dx), ry = rcp(dy);
    // world velocity += R (dv_b + h w x v_b)
    const V3 dv = v3(fmaf(vd.h, c.x, x1), fmaf(vd.h, c.y, x2), fmaf(ha.z, rz, hrot.z));
    s.v = v3(fmaf(R.r0.x, dv.x, fmaf(R.r0.y, dv.y, fmaf(R.r0.z, dv.z, s.v.x))), fmaf(R.r1.x, dv.x, fmaf(R.r1.y, dv.y, fmaf(R.r1.z, dv.z, s.v.y))),
             fmaf(R.r2.x, dv.x, fmaf(R.r2.y, dv.y, fmaf(R.r2.z, dv.z, s.v.z))));
    s.wb = v3(fmaf(hal.x, rx, w.x), fmaf(hal.y, ry, w.y), fmaf(x3, vd.inv_gz, w.z));
    pose_integrate(vd, s);
}

// sum over the 4 lanes of a quad with DPP quad_perm swaps (no LDS traffic); every lane gets the SAME bits
WL_DEV float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}
// broadcast lane K of each quad to its 4 lanes
template <int K>
WL_DEV float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xF, 0xF, true));
}
WL_DEV V3 quad_sum(V3 a) { return v3(quad_sum(a.x), quad_sum(a.y), quad_sum(a.z)); }

// One integrator sub-step.
//   LANES == 1: one lane owns the env and visits its 4 wheels (throughput form: no redundant work).  UNROLL: all four
//               wheels inlined (sub-expressions shared across axles and sides, more registers) or a rolled loop over
//               the two axles (fewer registers -> more wavefronts per SIMD).
//   LANES == 4: a quad of lanes owns the env, lane `wid` owns wheel `wid` (s.wheel[0] is ITS spin); the body state is
//               replicated, the wheel forces are summed across the quad with DPP.  Latency form for small env counts:
//               the critical path per sub-step drops from 4 wheels to 1.
//   DRIVE (lane form): 0 rear-wheel drive, 1 four-wheel drive: compiled in -- the other drive's front-axle code, its
//               branch and its live values (two wheel targets) leave the loop; -1: decided at run time (vp.drive)
template <int LANES, class Ground, bool UNROLL = true, int DRIVE = -1, bool IMPL = false>
WL_DEV void vehicle_substep(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, VehState& s,
                            const Ground& ground, int wid, float sn, float cs /* sin / cos of the steer angle in force */) {
    const Mat3 R = mat_from_quat(s.q);
    const V3 vb = mul_t(R, s.v);
    Wrench w;
    Jac J;
    if constexpr (LANES == 1 && Ground::kFlat) {
        const bool awd = DRIVE == 1 || (DRIVE < 0 && vp.drive == 1);
        // rear axle: always driven, never steered; front axle: steered, driven only with 4WD (the undriven wheel has no
        // motor arithmetic at all)
        const AxleOut ar = axle_step<false, true, IMPL, true>(vp, vd, ec, R, s, vb, -vp.half_wheelbase_r, 1.f, 0.f, ec.damp, ec.inv_A0_damp,
                                                              ec.wheel_target[0], ec.wheel_target[1], s.wheel[0], s.wheel[1], &J, vd.lxr);
        if constexpr (!UNROLL) __builtin_amdgcn_sched_barrier(0);
        AxleOut af;
        if (awd) af = axle_step<true, true, IMPL, false>(vp, vd, ec, R, s, vb, vp.half_wheelbase_f, cs, sn, ec.damp, ec.inv_A0_damp, ec.wheel_target[2],
                                                         ec.wheel_target[3], s.wheel[2], s.wheel[3], &J, vd.lxf);
        else af = axle_step<true, false, IMPL, false>(vp, vd, ec, R, s, vb, vp.half_wheelbase_f, cs, sn, 0.f, vd.inv_A0, 0.f, 0.f, s.wheel[2], s.wheel[3],
                                                      &J, vd.lxf);
        if constexpr (!UNROLL) __builtin_amdgcn_sched_barrier(0);
        const V3 n = R.r2;
        const float r = vp.wheel_radius, ht = vp.half_track;
        const float ayc = -r * n.y, az = fmaf(-r, n.z, vd.zrel);   // arm.y = +-ht + ayc ; arm.z
        w.F = ar.F + af.F;
        const float DX = ar.dX + af.dX, DZ = ar.dZ + af.dZ;
        w.T = v3(fmaf(ht, DZ, fmaf(ayc, w.F.z, -az * w.F.y)), fmaf(az, w.F.x, -fmaf(ar.ax, ar.F.z, af.ax * af.F.z)),
                 fmaf(ar.ax, ar.F.y, af.ax * af.F.y) - fmaf(ht, DX, ayc * w.F.x));
        w.Fz = ar.Fz + af.Fz;
    } else if constexpr (LANES == 1) {
        const float ht = vp.half_track, bxr = -vp.half_wheelbase_r, bxf = vp.half_wheelbase_f;
        wheel_step<Ground, false, true, true, 0, IMPL>(vp, vd, ec, ground, R, s, vb, bxr, ht, 1.f, 0.f, ec.damp, ec.inv_A0_damp,
                                                       ec.wheel_target[0], s.wheel[0], w, &J, vd.lxr, vd.ly);
        wheel_step<Ground, false, true, false, 1, IMPL>(vp, vd, ec, ground, R, s, vb, bxr, -ht, 1.f, 0.f, ec.damp, ec.inv_A0_damp,
                                                        ec.wheel_target[1], s.wheel[1], w, &J, vd.lxr, -vd.ly);
        if (DRIVE == 1 || (DRIVE < 0 && vp.drive == 1)) {
            wheel_step<Ground, true, true, false, 2, IMPL>(vp, vd, ec, ground, R, s, vb, bxf, ht, cs, sn, ec.damp, ec.inv_A0_damp,
                                                           ec.wheel_target[2], s.wheel[2], w, &J, vd.lxf, vd.ly);
            wheel_step<Ground, true, true, false, 3, IMPL>(vp, vd, ec, ground, R, s, vb, bxf, -ht, cs, sn, ec.damp, ec.inv_A0_damp,
                                                           ec.wheel_target[3], s.wheel[3], w, &J, vd.lxf, -vd.ly);
        } else {
            wheel_step<Ground, true, false, false, 2, IMPL>(vp, vd, ec, ground, R, s, vb, bxf, ht, cs, sn, 0.f, vd.inv_A0, 0.f, s.wheel[2], w,
                                                            &J, vd.lxf, vd.ly);
            wheel_step<Ground, true, false, false, 3, IMPL>(vp, vd, ec, ground, R, s, vb, bxf, -ht, cs, sn, 0.f, vd.inv_A0, 0.f, s.wheel[3], w,
                                                            &J, vd.lxf, -vd.ly);
        }
    } else {
        const bool front = wid >= 2;
        const float hc = front ? cs : 1.f, hs = front ? sn : 0.f;
        wheel_step<Ground, true, true, true, 0, IMPL>(vp, vd, ec, ground, R, s, vb, ec.bx_lane, ec.by_lane, hc, hs, ec.d_lane, ec.inv_A0d_lane,
                                                      ec.wt_lane, s.wheel[0], w, &J, ec.lx_lane, ec.ly_lane);
        w.F = quad_sum(w.F);
        w.T = quad_sum(w.T);
        if constexpr (Ground::kFlat && !IMPL) w.Fz = quad_sum(w.Fz);
        if constexpr (IMPL) {
            J.xx = quad_sum(J.xx), J.yy = quad_sum(J.yy), J.xy = quad_sum(J.xy), J.xw = quad_sum(J.xw), J.yw = quad_sum(J.yw);
            J.ww = quad_sum(J.ww), J.nc = quad_sum(J.nc);
        }
    }
    if constexpr (LANES == 1 && !UNROLL) __builtin_amdgcn_sched_barrier(0);
    if constexpr (IMPL) body_integrate_implicit(vd, ec, s, R, vb, w.F, w.T, J);
    else body_integrate<Ground::kFlat>(vd, ec, s, R, w.F, w.T, w.Fz);
}

// decimation x substeps integrator sub-steps (everything in registers).  A variant that software-pipelined the steering
// joint one sub-step ahead measured no gain: a wavefront alone on its SIMD pays ~3 ns per instruction whatever the
// chain looks like (tools/microbench/valu_issue.hip), so only fewer instructions on the critical lane help.
template <int LANES, class Ground, bool UNROLL = true, int DRIVE = -1, bool IMPL = false>
WL_DEV void vehicle_integrate(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, VehState& s,
                              const Ground& ground, int wid = 0) {
    for (int k = 0; k < vd.n_sub; ++k) {
        steer_update(vp, vd, ec, s);
        float sn, cs;
        sincos_fast(s.th, sn, cs);   // |th| <= tan(0.488) rad: hardware sin/cos, ~1e-6 abs
        vehicle_substep<LANES, Ground, UNROLL, DRIVE, IMPL>(vp, vd, ec, s, ground, wid, sn, cs);
    }
}
