// wl_drift_terms.h -- drift-task mdp terms as device functions (one lane = one env).
// Each function restates one reference function; citations are relative to
// /root/reference/source/wheeledlab_tasks/wheeledlab_tasks/drifting/mushr_drift_env_cfg.py unless noted.
#pragma once
#include "../../include/wheeledlab_amd.h"
#include "wl_math.h"

// ---- action term ----------------------------------------------------------------------------------------
// ClipAction.action (wheeledlab_rl/utils/clip_action.py:27) + AckermannAction.process_actions
// (wheeledlab/envs/mdp/actions/ackermann_actions.py:119-133)
WL_DEV void process_action(const WlActionParams& ap, float& a0, float& a1, float& v, float& delta) {
    if (ap.clip_wrapper) {
        a0 = clampf(a0, -1.f, 1.f);
        a1 = clampf(a1, -1.f, 1.f);
    }
    float b0 = a0, b1 = a1;
    if (ap.bounding == 1) {
        b0 = clampf(b0, -1.f, 1.f);
        b1 = clampf(b1, -1.f, 1.f);
    } else if (ap.bounding == 2) {
        b0 = tanhf(b0);
        b1 = tanhf(b1);
    }
    v = fmaf(b0, ap.scale[0], ap.offset[0]);
    delta = fmaf(b1, ap.scale[1], ap.offset[1]);
    if (ap.no_reverse) v = fmaxf(v, 0.f);
}

// RCCarRWDAction / RCCar4WDAction._calculate_ackermann_angles_and_velocities
// (wheeledlab/envs/mdp/actions/rc_car_actions.py:12-29, 36-64): steer joint target = tan(delta);
// wheel velocity targets in order bl, br, fl, fr.  map 2 = the base class (ackermann_actions.py:150-201): the 4WD wheel
// speeds; its two steer joints take the true Ackermann ANGLES (steer_pair below) -- the single-track vehicle model of the step
// kernels steers by their centre-line equivalent atan(L / R) = delta itself.
WL_DEV void joint_targets(const WlActionParams& ap, float v, float delta, float& steer, float w[4]) {
    const float t = tan_fast(delta);   // hardware sin/cos: |delta| <= scale[1] (0.488 rad), abs error ~1e-6
    steer = ap.map == 2 ? delta : t;
    const float inv_r = 1.f / ap.wheel_radius;
    if (ap.map == 0) {
        w[0] = w[1] = v * inv_r;
        w[2] = w[3] = 0.f;
    } else {
        const float L = ap.base_length, W2 = 0.5f * ap.base_width;
        const float R = (t == 0.f) ? 1e6f : L / t;
        const float inv_Rr = 1.f / (R * ap.wheel_radius);
        const float rl = sqrtf((R - W2) * (R - W2) + L * L), rr = sqrtf((R + W2) * (R + W2) + L * L);
        w[0] = v * fabsf((R - W2) * inv_Rr);
        w[1] = v * fabsf((R + W2) * inv_Rr);
        w[2] = v * fabsf(rl * inv_Rr);
        w[3] = v * fabsf(rr * inv_Rr);
    }
}

// the two steer-joint targets (left, right).  RC-car maps: both = tan(delta).  map 2, AckermannAction.
// _calculate_ackermann_angles_and_velocities (ackermann_actions.py:178-186): R = L / tan(delta) (1e6 where tan(delta) == 0),
// delta_left = atan(L / (R - W / 2)), delta_right = atan(L / (R + W / 2)).  Only the parity entry point wl_action_map uses it
// (no registered task selects the base class), so the library tan / atan are affordable here.
WL_DEV void steer_pair(const WlActionParams& ap, float delta, float steer, float& left, float& right) {
    left = right = steer;
    if (ap.map == 2) {
        const float t = tanf(delta);
        const float L = ap.base_length, W2 = 0.5f * ap.base_width;
        const float R = (t == 0.f) ? 1e6f : L / t;
        left = atanf(L / (R - W2));
        right = atanf(L / (R + W2));
    }
}

// ---- terminations -----------------------------------------------------------------------------------------
// cart_off_track (:343-348) = off_track (:210-217) OR in_range (:201-208)
WL_DEV bool cart_off_track(float x, float y, float straight, float r_in, float r_out) {
    if (fabsf(y) < straight) return fabsf(x) > r_out || fabsf(x) < r_in;
    const float dy = y > 0.f ? y - straight : y + straight;
    const float d2 = fmaf(dy, dy, x * x);
    return d2 > r_out * r_out || d2 < r_in * r_in;
}

// ---- rewards ------------------------------------------------------------------------------------------------
// side_slip (:219-230)
// `slip_angle` = atan2(v_by, v_bx), passed in so that the quad kernels can batch it with the Euler-angle atan2s
WL_DEV float side_slip_from_angle(float slip_angle, float vbx, float min_thresh, float max_thresh, float min_vel_x) {
    float ang = fabsf(slip_angle);
    if (fabsf(vbx) < min_vel_x || ang > max_thresh) ang = 0.f;
    return ang < min_thresh ? 0.f : ang;
}
WL_DEV float side_slip(V3 vb, float min_thresh, float max_thresh, float min_vel_x) {
    return side_slip_from_angle(atan2_fast(vb.y, vb.x), vb.x, min_thresh, max_thresh, min_vel_x);
}
// vel_dist (:167-171)
WL_DEV float vel_dist(V3 vb, float target, float offset) {
    const float gs = fsqrt(fmaf(vb.x, vb.x, vb.y * vb.y));   // v_sqrt_f32: 1 ulp
    return fmaf(gs - target, gs - target, offset);
}
// turn_left_go_right (:232-240); steer_mean = mean of the two steer joint positions
WL_DEV float turn_left_go_right(float steer_mean, float wbz, float thresh) {
    return fmaxf(-steer_mean * clampf(wbz, -thresh, thresh), 0.f);
}
// energy_through_turn (:195-199): 3-D speed squared on the corners
WL_DEV float energy_through_turn(float y, V3 vb, float straight) { return fabsf(y) > straight ? dot(vb, vb) : 0.f; }
// cross_track_dist (:173-193)
WL_DEV float cross_track_dist(float x, float y, float straight, float r, float offset, float p) {
    float d;
    if (fabsf(y) < straight) {
        d = fabsf(x > 0.f ? x - r : x + r);
    } else {
        const float dy = y > 0.f ? y - straight : y + straight;
        d = fabsf(fsqrt(fmaf(dy, dy, x * x)) - r);
    }
    const float ctd = d + offset;
    return p == 1.f ? ctd : powf(ctd, p);
}

struct DriftTerms {
    float t[WL_DR_NTERMS];
};

// all 7 unweighted reward terms (DriftRewardsCfg :246-299); is_terminated_term is IsaacLab's
// (terminated by a non-time-out term) * (not timed out)
WL_DEV DriftTerms drift_terms(const WlDriftParams& p, V3 pos, V3 vb, V3 wb, float wwz, float steer_mean, bool terminated,
                              bool timed_out, float slip_angle) {
    DriftTerms r;
    r.t[WL_DR_SIDE_SLIP] = side_slip_from_angle(slip_angle, vb.x, p.slip_min, p.slip_max, p.slip_min_vx);
    r.t[WL_DR_VEL] = vel_dist(vb, p.speed_target, p.speed_offset);
    r.t[WL_DR_PROGRESS] = wwz;   // track_progress_rate (:160-165): world-frame yaw rate of the root link
    r.t[WL_DR_TLGR] = turn_left_go_right(steer_mean, wb.z, p.tlgr_thresh);
    r.t[WL_DR_TURN_ENERGY] = energy_through_turn(pos.y, vb, p.straight);
    r.t[WL_DR_CROSS_TRACK] = cross_track_dist(pos.x, pos.y, p.straight, p.r_line, p.ctd_offset, p.ctd_p);
    r.t[WL_DR_TERM_PENS] = (terminated && !timed_out) ? 1.f : 0.f;
    return r;
}
