"""Drift-task mdp terms, numpy float32 restatement.  PARITY PINNED by tests/golden/drift_mdp_*.npz, actions.npz.

Every function cites the reference lines it follows (paths relative to /root/reference/source/).  Inputs are the
same state tensors the reference functions read through `isaaclab.envs.mdp` accessors:
  pos [N,3]  = mdp.root_pos_w (env-relative), v_b [N,3] = mdp.base_lin_vel, w_b [N,3] = mdp.base_ang_vel,
  w_w [N,3]  = asset.data.root_link_ang_vel_w, steer [N,2] = joint_pos[:, steer joints].
"""
import numpy as np

from .mathlib import F, f32, euler_xyz_from_quat


# ---- action term ---------------------------------------------------------------------------------------

def clip_action(a):
    """ClipAction.action, wheeledlab_rl/wheeledlab_rl/utils/clip_action.py:18-27 (low/high = -1/+1, scripts/train_rl.py:73-74)"""
    return np.clip(f32(a), F(-1), F(1))


def process_actions(a, ap):
    """AckermannAction.process_actions, wheeledlab/wheeledlab/envs/mdp/actions/ackermann_actions.py:119-133"""
    a = f32(a)
    scale, offset = f32(list(ap.scale)), f32(list(ap.offset))
    if ap.bounding == 1:
        b = np.clip(a, F(-1), F(1)) * scale + offset
    elif ap.bounding == 2:
        b = np.tanh(a) * scale + offset
    else:
        b = a * scale + offset
    if ap.no_reverse:
        b[:, 0] = np.maximum(b[:, 0], F(0))
    return b.astype(F)


def rwd_targets(v, delta, ap):
    """RCCarRWDAction._calculate_ackermann_angles_and_velocities, .../actions/rc_car_actions.py:12-29
    -> steer position target [N,2] (tan delta, both joints), wheel velocity target [N,2] (bl, br)"""
    t = np.tan(f32(delta))
    w = f32(v) / F(ap.wheel_radius)
    return np.stack([t, t], -1).astype(F), np.stack([w, w], -1).astype(F)


def fwd_targets(v, delta, ap):
    """RCCar4WDAction._calculate_ackermann_angles_and_velocities, .../actions/rc_car_actions.py:36-64
    -> steer target [N,2], wheel velocity target [N,4] in order bl, br, fl, fr"""
    v, t = f32(v), np.tan(f32(delta))
    L, W, r = F(ap.base_length), F(ap.base_width), F(ap.wheel_radius)
    with np.errstate(divide="ignore"):
        R = np.where(t == 0, F(1e6), L / t).astype(F)
    rl = np.sqrt((R - W / 2) ** 2 + L ** 2)
    rr = np.sqrt((R + W / 2) ** 2 + L ** 2)
    fl = v * np.abs(rl / (R * r))
    fr = v * np.abs(rr / (R * r))
    bl = v * np.abs((R - W / 2) / (R * r))
    br = v * np.abs((R + W / 2) / (R * r))
    return np.stack([t, t], -1).astype(F), np.stack([bl, br, fl, fr], -1).astype(F)


def ackermann_base_targets(v, delta, ap):
    """AckermannAction._calculate_ackermann_angles_and_velocities (base class, true Ackermann angles),
    ackermann_actions.py:150-201.  Not used by a registered task."""
    v, t = f32(v), np.tan(f32(delta))
    L, W, r = F(ap.base_length), F(ap.base_width), F(ap.wheel_radius)
    with np.errstate(divide="ignore"):
        R = np.where(t == 0, F(1e6), L / t).astype(F)
    dl, dr = np.arctan(L / (R - W / 2)), np.arctan(L / (R + W / 2))
    _, w = fwd_targets(v, delta, ap)
    return np.stack([dl, dr], -1).astype(F), w


# ---- terminations --------------------------------------------------------------------------------------

def in_range(pos, straight, r_in):
    """wheeledlab_tasks/wheeledlab_tasks/drifting/mushr_drift_env_cfg.py:201-208 (int 0/1)"""
    x, y = f32(pos)[:, 0], f32(pos)[:, 1]
    s, r2 = F(straight), F(r_in) ** 2
    return np.where(np.abs(y) < s, np.abs(x) < F(r_in),
                    np.where(y > 0, (y - s) ** 2 + x ** 2 < r2, (y + s) ** 2 + x ** 2 < r2)).astype(np.int64)


def off_track(pos, straight, r_out):
    """mushr_drift_env_cfg.py:210-217 (int 0/1)"""
    x, y = f32(pos)[:, 0], f32(pos)[:, 1]
    s, r2 = F(straight), F(r_out) ** 2
    return np.where(np.abs(y) < s, np.abs(x) > F(r_out),
                    np.where(y > 0, (y - s) ** 2 + x ** 2 > r2, (y + s) ** 2 + x ** 2 > r2)).astype(np.int64)


def cart_off_track(pos, straight, r_in, r_out):
    """mushr_drift_env_cfg.py:343-348"""
    return np.logical_or(off_track(pos, straight, r_out) > 0.5, in_range(pos, straight, r_in) > 0.5)


def time_out(episode_len, max_episode_length):
    """isaaclab mdp.time_out (cfg at mushr_drift_env_cfg.py:353) -- unpinned restatement"""
    return np.asarray(episode_len) >= max_episode_length


# ---- rewards -------------------------------------------------------------------------------------------

def side_slip(v_b, min_thresh, max_thresh, min_vel_x):
    """mushr_drift_env_cfg.py:219-230"""
    v_b = f32(v_b)
    ang = np.abs(np.arctan2(v_b[:, 1], v_b[:, 0]))
    ang = np.where(np.logical_or(np.abs(v_b[:, 0]) < F(min_vel_x), ang > F(max_thresh)), F(0), ang)
    return np.where(ang < F(min_thresh), F(0), ang).astype(F)


def vel_dist(v_b, speed_target, offset):
    """mushr_drift_env_cfg.py:167-171"""
    v_b = f32(v_b)
    gs = np.sqrt(v_b[:, 0] ** 2 + v_b[:, 1] ** 2)
    return ((gs - F(speed_target)) ** 2 + F(offset)).astype(F)


def track_progress_rate(w_w):
    """mushr_drift_env_cfg.py:160-165"""
    return f32(w_w)[:, 2]


def turn_left_go_right(steer, w_b, thresh):
    """mushr_drift_env_cfg.py:232-240"""
    sp = f32(steer).mean(-1)
    av = np.clip(f32(w_b)[:, 2], -F(thresh), F(thresh))
    return np.maximum(sp * av * F(-1), F(0)).astype(F)


def energy_through_turn(pos, v_b, straight):
    """mushr_drift_env_cfg.py:195-199 (3-D speed squared)"""
    v_b = f32(v_b)
    sp = np.sqrt((v_b ** 2).sum(-1))
    return np.where(np.abs(f32(pos)[:, 1]) > F(straight), sp ** 2, F(0)).astype(F)


def cross_track_dist(pos, straight, track_radius, offset, p):
    """mushr_drift_env_cfg.py:173-193"""
    x, y = f32(pos)[:, 0], f32(pos)[:, 1]
    s, r = F(straight), F(track_radius)
    sq = np.where(np.abs(y) < s,
                  np.where(x > 0, (x - r) ** 2, (x + r) ** 2),
                  np.where(y > 0, (np.sqrt((y - s) ** 2 + x ** 2) - r) ** 2,
                           (np.sqrt((y + s) ** 2 + x ** 2) - r) ** 2))
    ctd = np.sqrt(sq) + F(offset)
    return np.power(ctd, F(p)).astype(F)


def is_terminated_term(terminated, timed_out):
    """isaaclab mdp.rewards.is_terminated_term (cfg mushr_drift_env_cfg.py:295-299) -- unpinned restatement"""
    return (np.asarray(terminated).astype(F) * (~np.asarray(timed_out, bool)).astype(F)).astype(F)


def drift_terms(p, pos, v_b, w_b, w_w, steer, terminated, timed_out):
    """all 7 unweighted terms in WlDriftRewTerm order -> [7, N]"""
    return np.stack([
        side_slip(v_b, p.slip_min, p.slip_max, p.slip_min_vx),
        vel_dist(v_b, p.speed_target, p.speed_offset),
        track_progress_rate(w_w),
        turn_left_go_right(steer, w_b, p.tlgr_thresh),
        energy_through_turn(pos, v_b, p.straight),
        cross_track_dist(pos, p.straight, p.r_line, p.ctd_offset, p.ctd_p),
        is_terminated_term(terminated, timed_out),
    ]).astype(F)


def reward_sum(p, terms):
    """RewardManager.compute (IsaacLab, unpinned): sum_i w_i * f_i * step_dt, terms with w == 0 skipped"""
    dt = F(p.sim_dt) * F(p.decimation)
    r = np.zeros(terms.shape[1], F)
    contrib = np.zeros_like(terms)
    for i in range(terms.shape[0]):
        w = F(p.weight[i])
        if w == 0:
            continue
        contrib[i] = terms[i] * w * dt
        r = r + contrib[i]
    return r.astype(F), contrib.astype(F)


# ---- observation -----------------------------------------------------------------------------------------

def root_euler_xyz(quat):
    """wheeledlab/wheeledlab/envs/mdp/observations.py:9-12"""
    return np.stack(euler_xyz_from_quat(quat), -1).astype(F)


def blind_obs(p, pos, quat, v_b, w_b, last_action, normals=None):
    """BlindObsCfg.PolicyCfg, wheeledlab_tasks/common/observations.py:24-54: 14-dim
    [pos(3) | euler xyz(3) | base lin vel(3) | base ang vel(3) | last action clipped to +-1 (2)],
    Gaussian corruption std (0.1, 0.1, 0.5, 0.4) when enabled (noise -> clip -> scale order, IsaacLab)."""
    parts = [f32(pos), root_euler_xyz(quat), f32(v_b), f32(w_b)]
    if p.enable_corruption and normals is not None:
        normals = f32(normals)  # [12, N]
        parts = [parts[k] + F(p.noise_std[k]) * normals[3 * k:3 * k + 3].T for k in range(4)]
    parts.append(np.clip(f32(last_action), F(-1), F(1)))
    return np.concatenate(parts, -1).astype(F)
