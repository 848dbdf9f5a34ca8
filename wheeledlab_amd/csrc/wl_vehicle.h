// wl_vehicle.h -- rigid body + 4 tyre contacts, one integrator sub-step, all state in registers.
// Replaces the PhysX articulation step of the reference (mushr_drift_env_cfg.py:393-394); model derivation in
// DESIGN.md section 4, executable spec in oracle/vehicle.py.  Actuator constants: wheeledlab_assets/hound.py:4-52.
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

struct EnvConst {   // per-env constants hoisted out of the sub-step loop
    float weight, h_inv_mass;  // m g ; h / m
    float K_cap;               // 0.125 m / h
    float mu_s, mu_d;          // combined (wheel x ground) friction
    float damp;                // throttle damping of driven wheels
    float inv_A0, inv_A0_damp; // 1 / (Iw/h + bearing damping) and 1 / (that + damp): the saturated-branch wheel solve
    V3 Ib, h_inv_Ib;           // body inertia diag ; h / I
    float steer_target;
    float wheel_target[4];
    float wt_lane;             // quad form: the target of THIS lane's wheel (picked once per env-step, not per sub-step)
};


// Uniform constants derived from WlVehicleParams + dt.  gfx950 has no scalar float ALU: anything computed from
// uniform parameters inside the kernel lands in VGPRs and, being loop-invariant, stays there through every sub-step.
// The C-ABI wrappers compute these once on the host instead, so in-kernel they are SGPR operands.
struct VehDerived {
    float h, inv_h, half_h;                 // sub-step length
    float steer_a, steer_b;                 // h kp / J ;  1 / (1 + h kd / J + h^2 kp / J)
    float steer_J_h, steer_h_J;             // J / h ; h / J
    float zrel;                             // wheel centre z relative to the CoM (body frame)
    float Iw_h, A0;                         // wheel inertia / h ; Iw_h + bearing damping
    float inv_wlim;                         // 1 / motor_vel_limit
    float r2;                               // wheel radius squared
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
    d.inv_wlim = 1.f / vp.motor_vel_limit;
    d.r2 = vp.wheel_radius * vp.wheel_radius;
    d.n_sub = decimation * vp.substeps;
    return d;
}

WL_DEV void env_const_mass(EnvConst& ec, const WlVehicleParams& vp, const VehDerived& vd, float mass) {
    ec.weight = mass * vp.gravity;
    ec.h_inv_mass = vd.h * rcp(mass);
    ec.K_cap = 0.125f * mass * vd.inv_h;
    ec.Ib = v3(mass * (vp.gyr_x * vp.gyr_x), mass * (vp.gyr_y * vp.gyr_y), mass * (vp.gyr_z * vp.gyr_z));
    ec.h_inv_Ib = v3(vd.h * rcp(ec.Ib.x), vd.h * rcp(ec.Ib.y), vd.h * rcp(ec.Ib.z));
}

struct FlatGround {
    static constexpr bool kFlat = true;   // n == (0, 0, 1), zg == 0 everywhere: wheel_force drops the terms that vanish
    WL_DEV void sample(float, float, float& zg, V3& n) const {
        zg = 0.f;
        n = v3(0.f, 0.f, 1.f);
    }
};

// by-value pick of this lane's element: the operands are SSA values, so the selection is three v_cndmask.  (Written as
// a ternary chain over array elements the compiler folds it into a wid-indexed load of the array, which then lives in
// scratch memory -- a store -> load round trip on the critical tail of the step.)
WL_DEV float quad_pick(int wid, float a, float b, float c, float d) { return wid == 0 ? a : wid == 1 ? b : wid == 2 ? c : d; }

// Contact + tyre + wheel-spin solve of ONE wheel.  `front` / `left` are compile-time constants in the lane-per-env
// kernels (the call is inlined per wheel) and per-lane values in the quad kernels (one wheel per lane).
// Returns the contact force on the body (world) and its torque about the CoM; updates the wheel spin.
// FLAT (plane z = 0, n = +z): the products with the zero components of n are spelled out as absent -- the compiler must
// keep `x * 0` (x could be inf / NaN) and would issue a dozen of them per wheel and sub-step.
template <bool FLAT>
WL_DEV void wheel_force(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, const Mat3& R, V3 x, V3 v,
                        V3 ww, float cs, float sn, float zg, V3 n, bool front, bool left, float wt, float& w_spin,
                        V3& Fi, V3& Ti) {
    const float r = vp.wheel_radius;
    const V3 pb = v3(front ? vp.half_wheelbase_f : -vp.half_wheelbase_r, left ? vp.half_track : -vp.half_track, vd.zrel);
    const V3 arm_c = mul(R, pb);
    const float cz = x.z + arm_c.z;
    float pen, vn, vcx, vcy;
    V3 arm, vcp, tx, ty;
    if constexpr (FLAT) {
        pen = r - cz;
        arm = v3(arm_c.x, arm_c.y, arm_c.z - r);
        vcp = v + cross(ww, arm);
        vn = vcp.z;
    } else {
        pen = r - (cz - zg) * n.z;
        arm = fma3(-r, n, arm_c);
        vcp = v + cross(ww, arm);
        vn = dot(vcp, n);
    }
    const float Fz = pen > 0.f ? fmaxf(fmaf(vp.susp_k, pen, -vp.susp_c * vn), 0.f) : 0.f;
    // wheel heading projected into the contact plane (front wheels are rotated by the steer angle)
    const float hc = front ? cs : 1.f, hs = front ? sn : 0.f;
    const V3 hw = v3(fmaf(R.r0.x, hc, R.r0.y * hs), fmaf(R.r1.x, hc, R.r1.y * hs), fmaf(R.r2.x, hc, R.r2.y * hs));
    if constexpr (FLAT) {
        const float inv = rsq(fmaf(hw.x, hw.x, hw.y * hw.y));
        tx = v3(inv * hw.x, inv * hw.y, 0.f);
        ty = v3(-tx.y, tx.x, 0.f);
        vcx = fmaf(vcp.x, tx.x, vcp.y * tx.y);
        vcy = fmaf(vcp.y, tx.x, -vcp.x * tx.y);
    } else {
        const V3 t = fma3(-dot(hw, n), n, hw);
        tx = rsq(dot(t, t)) * t;
        ty = cross(n, tx);
        vcx = dot(vcp, tx);
        vcy = dot(vcp, ty);
    }
    const float w_i = w_spin;
    const float vden = fmaxf(vp.v_min, vp.slip_peak * fmaxf(fabsf(vcx), fabsf(w_i * r)));
    const float inv_vden = rcp(vden);
    const float sx = (w_i * r - vcx) * inv_vden, sy = -vcy * inv_vden;
    const float sig = fsqrt(fmaf(sx, sx, sy * sy));
    // g(sig) / sig, branch-free: below the peak (sig <= 1, inv_sig == 1) the first term is mu_s and the second adds
    // mu_s (1 - sig) -> mu_s (2 - sig); above it the second term vanishes.  (As a ?: the two short arms become
    // divergent control flow: five exec-mask instructions around six arithmetic ones, every wheel, every sub-step.)
    const float inv_sig = rcp(fmaxf(sig, 1.f));
    const float gq = fmaf(ec.mu_s, fmaxf(1.f - sig, 0.f), fmaf(ec.mu_s - ec.mu_d, inv_sig, ec.mu_d) * inv_sig);
    // explicit-stepping stability cap: at most half of this wheel's share of the body momentum per sub-step
    const float K = fminf(Fz * gq * inv_vden, ec.K_cap);
    const bool driven = (vp.drive == 1) || !front;
    const float d = driven ? ec.damp : 0.f;
    // DC-motor torque window at the current spin (IsaacLab DCMotor, hound.py:13-21)
    const float rel = w_i * vd.inv_wlim;
    const float tau_hi = clampf(vp.motor_sat * (1.f - rel), 0.f, vp.motor_limit);
    const float tau_lo = clampf(vp.motor_sat * (-1.f - rel), -vp.motor_limit, 0.f);
    const float Iw_h = vd.Iw_h;
    // implicit spin update with the tyre's secant stiffness (unconditionally stable)
    const float A = fmaf(K, vd.r2, vd.A0);
    const float rhs0 = fmaf(Iw_h, w_i, r * K * vcx);
    const float w_u = fmaf(d, wt, rhs0) * rcp(A + d);
    // tau = d (wt - w_u) cancels catastrophically near the target (d / A ~ 3e3 amplifies the rounding of w_u), so
    // the unclipped branch takes w_u itself and only a clipped (constant) torque is pushed through the wheel equation
    const float tau_u = d * (wt - w_u);
    const float tau = clampf(tau_u, tau_lo, tau_hi);
    float w_n = (tau == tau_u) ? w_u : (rhs0 + tau) * rcp(A);
    float Fx = K * fmaf(w_n, r, -vcx);
    float Fy = -K * vcy;
    const float Fmax = ec.mu_s * Fz;
    const float mag2 = fmaf(Fx, Fx, Fy * Fy);
    if (mag2 > Fmax * Fmax) {   // friction-circle saturation: re-solve the wheel against the force actually applied
        const float scale = Fmax * rsq(fmaxf(mag2, 1e-30f));
        Fx *= scale;
        Fy *= scale;
        const float rhs2 = fmaf(Iw_h, w_i, -r * Fx);
        const float w_u2 = fmaf(d, wt, rhs2) * (driven ? ec.inv_A0_damp : ec.inv_A0);   // 1 / (A0 + d), per-env constants
        const float tau_u2 = d * (wt - w_u2);
        const float tau2 = clampf(tau_u2, tau_lo, tau_hi);
        w_n = (tau2 == tau_u2) ? w_u2 : (rhs2 + tau2) * ec.inv_A0;
    }
    w_spin = w_n;
    if constexpr (FLAT) Fi = v3(fmaf(Fx, tx.x, -Fy * tx.y), fmaf(Fx, tx.y, Fy * tx.x), Fz);
    else Fi = fma3(Fz, n, fma3(Fx, tx, Fy * ty));
    Ti = cross(arm, Fi);
}

// ---- packed axle form (lane-per-env kernels) ---------------------------------------------------------------------------
// tools/microbench/valu_issue.hip: a wavefront that is alone on its SIMD issues one VALU instruction every ~4.5 cycles
// (8 when it depends on the previous one) whether it is v_fma_f32 or v_pk_fma_f32 -- the packed one is free there; with
// >= 2 wavefronts per SIMD the pipe takes 2 cycles for v_fma_f32 and 4 for v_pk_fma_f32 (same flops either way).  So
// packing pays exactly while the lane form runs at ~1 wavefront per SIMD (32 K .. ~130 K envs: 17.1 -> 13.8 us at
// 65 536) and is neutral beyond (1 M: 113 vs 114 us).  The two wheels of an axle run the same arithmetic on different
// data: held as float2 (.x = left, .y = right) their mul / add / fma become packed instructions; max / min / select /
// rcp / rsq / sqrt have no packed form and are issued per element.  wheel_force_axle is wheel_force, line by line,
// on pairs.
typedef float f2 __attribute__((ext_vector_type(2)));
struct V3p {
    f2 x, y, z;
};
WL_DEV f2 splat(float a) { return f2{a, a}; }
WL_DEV f2 pfma(f2 a, f2 b, f2 c) { return a * b + c; }     // contracts to v_pk_fma_f32
WL_DEV f2 pmax(f2 a, f2 b) { return f2{fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
WL_DEV f2 pmin(f2 a, f2 b) { return f2{fminf(a.x, b.x), fminf(a.y, b.y)}; }
WL_DEV f2 pabs(f2 a) { return f2{fabsf(a.x), fabsf(a.y)}; }
WL_DEV f2 prcp(f2 a) { return f2{rcp(a.x), rcp(a.y)}; }
WL_DEV f2 prsq(f2 a) { return f2{rsq(a.x), rsq(a.y)}; }
WL_DEV f2 psqrt(f2 a) { return f2{fsqrt(a.x), fsqrt(a.y)}; }
WL_DEV f2 pclamp(f2 a, f2 lo, f2 hi) { return pmin(pmax(a, lo), hi); }
WL_DEV f2 psel(bool c0, bool c1, f2 a, f2 b) { return f2{c0 ? a.x : b.x, c1 ? a.y : b.y}; }
WL_DEV f2 pdot(const V3p& a, const V3p& b) { return pfma(a.x, b.x, pfma(a.y, b.y, a.z * b.z)); }
WL_DEV f2 pdot(V3 a, const V3p& b) { return pfma(splat(a.x), b.x, pfma(splat(a.y), b.y, splat(a.z) * b.z)); }
WL_DEV V3p pcross(const V3p& a, const V3p& b) { return V3p{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
WL_DEV V3p pcross(V3 a, const V3p& b) {
    return V3p{splat(a.y) * b.z - splat(a.z) * b.y, splat(a.z) * b.x - splat(a.x) * b.z, splat(a.x) * b.y - splat(a.y) * b.x};
}

// the two wheels of one axle (front: steered).  F / T: their summed contact force / torque about the CoM (world).
template <bool FLAT>
WL_DEV void wheel_force_axle(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, const Mat3& R, V3 x, V3 v,
                             V3 ww, float cs, float sn, f2 zg, const V3p& n, bool front, f2 wt, f2& w_spin, V3& F, V3& T) {
    const f2 r = splat(vp.wheel_radius);
    const float bx = front ? vp.half_wheelbase_f : -vp.half_wheelbase_r;
    const f2 by = f2{vp.half_track, -vp.half_track};
    // arm_c = R pb with pb = (bx, +-half_track, zrel): the x / z parts are shared by the two wheels
    const V3p arm_c{pfma(splat(R.r0.y), by, splat(fmaf(R.r0.x, bx, R.r0.z * vd.zrel))),
                    pfma(splat(R.r1.y), by, splat(fmaf(R.r1.x, bx, R.r1.z * vd.zrel))),
                    pfma(splat(R.r2.y), by, splat(fmaf(R.r2.x, bx, R.r2.z * vd.zrel)))};
    const f2 cz = splat(x.z) + arm_c.z;
    f2 pen, vn, vcx, vcy;
    V3p arm, tx, ty;
    if constexpr (FLAT) {
        pen = r - cz;
        arm = V3p{arm_c.x, arm_c.y, arm_c.z - r};
    } else {
        pen = r - (cz - zg) * n.z;
        arm = V3p{arm_c.x - r * n.x, arm_c.y - r * n.y, arm_c.z - r * n.z};
    }
    const V3p wxa = pcross(ww, arm);
    const V3p vcp{splat(v.x) + wxa.x, splat(v.y) + wxa.y, splat(v.z) + wxa.z};
    if constexpr (FLAT) vn = vcp.z;
    else vn = pdot(vcp, n);
    const f2 fz_raw = pmax(pfma(splat(vp.susp_k), pen, -splat(vp.susp_c) * vn), splat(0.f));
    const f2 Fz = psel(pen.x > 0.f, pen.y > 0.f, fz_raw, splat(0.f));
    // wheel heading (shared by the axle) projected into each wheel's contact plane
    const float hc = front ? cs : 1.f, hs = front ? sn : 0.f;
    const V3 hw = v3(fmaf(R.r0.x, hc, R.r0.y * hs), fmaf(R.r1.x, hc, R.r1.y * hs), fmaf(R.r2.x, hc, R.r2.y * hs));
    if constexpr (FLAT) {   // one tangent frame for the axle
        const float inv = rsq(fmaf(hw.x, hw.x, hw.y * hw.y));
        const float tx_x = inv * hw.x, tx_y = inv * hw.y;
        tx = V3p{splat(tx_x), splat(tx_y), splat(0.f)};
        ty = V3p{splat(-tx_y), splat(tx_x), splat(0.f)};
        vcx = pfma(vcp.x, tx.x, vcp.y * tx.y);
        vcy = pfma(vcp.y, tx.x, -vcp.x * tx.y);
    } else {
        const f2 hn = pdot(hw, n);
        const V3p t{splat(hw.x) - hn * n.x, splat(hw.y) - hn * n.y, splat(hw.z) - hn * n.z};
        const f2 inv_t = prsq(pdot(t, t));
        tx = V3p{inv_t * t.x, inv_t * t.y, inv_t * t.z};
        ty = pcross(n, tx);
        vcx = pdot(vcp, tx);
        vcy = pdot(vcp, ty);
    }
    const f2 w_i = w_spin;
    const f2 vden = pmax(splat(vp.v_min), splat(vp.slip_peak) * pmax(pabs(vcx), pabs(w_i * r)));
    const f2 inv_vden = prcp(vden);
    const f2 sx = (w_i * r - vcx) * inv_vden, sy = -vcy * inv_vden;
    const f2 sig = psqrt(pfma(sx, sx, sy * sy));
    const f2 inv_sig = prcp(pmax(sig, splat(1.f)));
    const f2 gq = pfma(splat(ec.mu_s), pmax(splat(1.f) - sig, splat(0.f)),
                       pfma(splat(ec.mu_s - ec.mu_d), inv_sig, splat(ec.mu_d)) * inv_sig);
    const f2 K = pmin(Fz * gq * inv_vden, splat(ec.K_cap));
    const bool driven = (vp.drive == 1) || !front;
    const f2 d = splat(driven ? ec.damp : 0.f);
    const f2 rel = w_i * splat(vd.inv_wlim);
    const f2 tau_hi = pclamp(splat(vp.motor_sat) * (splat(1.f) - rel), splat(0.f), splat(vp.motor_limit));
    const f2 tau_lo = pclamp(splat(vp.motor_sat) * (splat(-1.f) - rel), splat(-vp.motor_limit), splat(0.f));
    const f2 Iw_h = splat(vd.Iw_h);
    const f2 A = pfma(K, splat(vd.r2), splat(vd.A0));
    const f2 rhs0 = pfma(Iw_h, w_i, r * K * vcx);
    const f2 w_u = pfma(d, wt, rhs0) * prcp(A + d);
    const f2 tau_u = d * (wt - w_u);
    const f2 tau = pclamp(tau_u, tau_lo, tau_hi);
    const f2 w_c = (rhs0 + tau) * prcp(A);
    f2 w_n = psel(tau.x == tau_u.x, tau.y == tau_u.y, w_u, w_c);
    f2 Fx = K * pfma(w_n, r, -vcx);
    f2 Fy = -K * vcy;
    const f2 Fmax = splat(ec.mu_s) * Fz;
    const f2 mag2 = pfma(Fx, Fx, Fy * Fy);
    const f2 fm2 = Fmax * Fmax;
    const bool sat0 = mag2.x > fm2.x, sat1 = mag2.y > fm2.y;
    if (sat0 || sat1) {   // friction-circle saturation: re-solve the wheel against the force actually applied
        const f2 scale = psel(sat0, sat1, Fmax * prsq(pmax(mag2, splat(1e-30f))), splat(1.f));
        Fx *= scale;
        Fy *= scale;
        const f2 inv_A2d = splat(driven ? ec.inv_A0_damp : ec.inv_A0);
        const f2 rhs2 = pfma(Iw_h, w_i, -r * Fx);
        const f2 w_u2 = pfma(d, wt, rhs2) * inv_A2d;
        const f2 tau_u2 = d * (wt - w_u2);
        const f2 tau2 = pclamp(tau_u2, tau_lo, tau_hi);
        const f2 w_c2 = (rhs2 + tau2) * splat(ec.inv_A0);
        const f2 w_s = psel(tau2.x == tau_u2.x, tau2.y == tau_u2.y, w_u2, w_c2);
        w_n = psel(sat0, sat1, w_s, w_n);
    }
    w_spin = w_n;
    V3p Fi;
    if constexpr (FLAT) Fi = V3p{pfma(Fx, tx.x, -Fy * tx.y), pfma(Fx, tx.y, Fy * tx.x), Fz};
    else Fi = V3p{pfma(Fz, n.x, pfma(Fx, tx.x, Fy * ty.x)), pfma(Fz, n.y, pfma(Fx, tx.y, Fy * ty.y)),
                  pfma(Fz, n.z, pfma(Fx, tx.z, Fy * ty.z))};
    const V3p Ti = pcross(arm, Fi);
    F = v3(Fi.x.x + Fi.x.y, Fi.y.x + Fi.y.y, Fi.z.x + Fi.z.y);
    T = v3(Ti.x.x + Ti.x.y, Ti.y.x + Ti.y.y, Ti.z.x + Ti.z.y);
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

// semi-implicit Euler of the rigid body under the summed contact force F / torque T (world, about the CoM)
WL_DEV void body_integrate(const VehDerived& vd, const EnvConst& ec, VehState& s, const Mat3& R, V3 F, V3 T) {
    F.z -= ec.weight;
    s.v = fma3(ec.h_inv_mass, F, s.v);
    const V3 Tb = mul_t(R, T);
    const V3 Iw = v3(ec.Ib.x * s.wb.x, ec.Ib.y * s.wb.y, ec.Ib.z * s.wb.z);
    const V3 gyro = cross(s.wb, Iw);
    s.wb = v3(fmaf(ec.h_inv_Ib.x, Tb.x - gyro.x, s.wb.x), fmaf(ec.h_inv_Ib.y, Tb.y - gyro.y, s.wb.y),
              fmaf(ec.h_inv_Ib.z, Tb.z - gyro.z, s.wb.z));
    const V3 w2 = mul(R, s.wb);
    s.x = fma3(vd.h, s.v, s.x);
    const float hh = vd.half_h;
    Quat q = s.q;
    Quat dq;
    dq.w = -w2.x * q.x - w2.y * q.y - w2.z * q.z;
    dq.x = w2.x * q.w + w2.y * q.z - w2.z * q.y;
    dq.y = -w2.x * q.z + w2.y * q.w + w2.z * q.x;
    dq.z = w2.x * q.y - w2.y * q.x + w2.z * q.w;
    q.w = fmaf(hh, dq.w, q.w);
    q.x = fmaf(hh, dq.x, q.x);
    q.y = fmaf(hh, dq.y, q.y);
    q.z = fmaf(hh, dq.z, q.z);
    const float inv_n = rsq(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    s.q = Quat{q.w * inv_n, q.x * inv_n, q.y * inv_n, q.z * inv_n};
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
//   LANES == 1: one lane owns the env and loops over its 4 wheels (throughput form: no redundant work).
//   LANES == 4: a quad of lanes owns the env, lane `wid` owns wheel `wid` (s.wheel[0] is ITS spin); the body state is
//               replicated, the wheel forces are summed across the quad with DPP.  Latency form for small env counts:
//               the critical path per sub-step drops from 4 wheels to 1.
template <int LANES, class Ground, bool PACKED = true>
WL_DEV void vehicle_substep(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, VehState& s,
                            const Ground& ground, int wid, float sn, float cs /* sin / cos of the steer angle in force */) {
    const Mat3 R = mat_from_quat(s.q);
    const V3 ww = mul(R, s.wb);
    V3 F = v3(0.f, 0.f, 0.f), T = v3(0.f, 0.f, 0.f);
    if constexpr (LANES == 1 && PACKED) {
        // two iterations (rear, front axle), the two wheels of an axle as packed pairs
#pragma unroll 1
        for (int ax = 0; ax < 2; ++ax) {
            const bool front = ax == 1;
            const float bx = front ? vp.half_wheelbase_f : -vp.half_wheelbase_r;
            const float sx_ = fmaf(R.r0.x, bx, R.r0.z * vd.zrel), sy_ = fmaf(R.r1.x, bx, R.r1.z * vd.zrel);
            const float cxl = s.x.x + fmaf(R.r0.y, vp.half_track, sx_), cxr = s.x.x + fmaf(R.r0.y, -vp.half_track, sx_);
            const float cyl = s.x.y + fmaf(R.r1.y, vp.half_track, sy_), cyr = s.x.y + fmaf(R.r1.y, -vp.half_track, sy_);
            float zl, zr;
            V3 nl, nr, Fa, Ta;
            ground.sample(cxl, cyl, zl, nl);
            ground.sample(cxr, cyr, zr, nr);
            const V3p n{f2{nl.x, nr.x}, f2{nl.y, nr.y}, f2{nl.z, nr.z}};
            f2 w = front ? f2{s.wheel[2], s.wheel[3]} : f2{s.wheel[0], s.wheel[1]};
            const f2 wt = front ? f2{ec.wheel_target[2], ec.wheel_target[3]} : f2{ec.wheel_target[0], ec.wheel_target[1]};
            wheel_force_axle<Ground::kFlat>(vp, vd, ec, R, s.x, s.v, ww, cs, sn, f2{zl, zr}, n, front, wt, w, Fa, Ta);
            s.wheel[0] = front ? s.wheel[0] : w.x;
            s.wheel[1] = front ? s.wheel[1] : w.y;
            s.wheel[2] = front ? w.x : s.wheel[2];
            s.wheel[3] = front ? w.y : s.wheel[3];
            F = F + Fa;
            T = T + Ta;
        }
    } else if constexpr (LANES == 1) {
        // rolled loop over the four wheels, one wheel's temporaries live at a time: the low-register variant (93 VGPRs -> 5
        // wavefronts per SIMD) for batches large enough to fill every SIMD several times over
#pragma unroll 1
        for (int i = 0; i < 4; ++i) {
            const bool front = i >= 2, left = (i & 1) == 0;
            const float bx = front ? vp.half_wheelbase_f : -vp.half_wheelbase_r, by = left ? vp.half_track : -vp.half_track;
            const float cx = s.x.x + fmaf(R.r0.x, bx, fmaf(R.r0.y, by, R.r0.z * vd.zrel));
            const float cy = s.x.y + fmaf(R.r1.x, bx, fmaf(R.r1.y, by, R.r1.z * vd.zrel));
            float zg;
            V3 n, Fi, Ti;
            ground.sample(cx, cy, zg, n);
            float w = i == 0 ? s.wheel[0] : i == 1 ? s.wheel[1] : i == 2 ? s.wheel[2] : s.wheel[3];
            const float wt = i == 0 ? ec.wheel_target[0] : i == 1 ? ec.wheel_target[1] : i == 2 ? ec.wheel_target[2] : ec.wheel_target[3];
            wheel_force<Ground::kFlat>(vp, vd, ec, R, s.x, s.v, ww, cs, sn, zg, n, front, left, wt, w, Fi, Ti);
            s.wheel[0] = i == 0 ? w : s.wheel[0];
            s.wheel[1] = i == 1 ? w : s.wheel[1];
            s.wheel[2] = i == 2 ? w : s.wheel[2];
            s.wheel[3] = i == 3 ? w : s.wheel[3];
            F = F + Fi;
            T = T + Ti;
        }
    } else {
        const bool front = wid >= 2, left = (wid & 1) == 0;
        const float bx = front ? vp.half_wheelbase_f : -vp.half_wheelbase_r, by = left ? vp.half_track : -vp.half_track;
        const float cx = s.x.x + fmaf(R.r0.x, bx, fmaf(R.r0.y, by, R.r0.z * vd.zrel));
        const float cy = s.x.y + fmaf(R.r1.x, bx, fmaf(R.r1.y, by, R.r1.z * vd.zrel));
        float zg;
        V3 n, Fi, Ti;
        ground.sample(cx, cy, zg, n);
        wheel_force<Ground::kFlat>(vp, vd, ec, R, s.x, s.v, ww, cs, sn, zg, n, front, left, ec.wt_lane, s.wheel[0], Fi, Ti);
        F = quad_sum(Fi);
        T = quad_sum(Ti);
    }
    body_integrate(vd, ec, s, R, F, T);
}

// decimation x substeps integrator sub-steps (everything in registers).  A variant that software-pipelined the steering
// joint one sub-step ahead measured no gain: a wavefront alone on its SIMD pays ~3 ns per instruction whatever the
// chain looks like (tools/microbench/valu_issue.hip), so only fewer instructions on the critical lane help.
template <int LANES, class Ground, bool PACKED = true>
WL_DEV void vehicle_integrate(const WlVehicleParams& vp, const VehDerived& vd, const EnvConst& ec, VehState& s,
                              const Ground& ground, int wid = 0) {
    for (int k = 0; k < vd.n_sub; ++k) {
        steer_update(vp, vd, ec, s);
        float sn, cs;
        sincos_fast(s.th, sn, cs);   // |th| <= tan(0.488) rad: hardware sin/cos, ~1e-6 abs
        vehicle_substep<LANES, Ground, PACKED>(vp, vd, ec, s, ground, wid, sn, cs);
    }
}
