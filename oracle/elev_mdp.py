"""Elevation-task mdp terms, numpy float32.  PARITY PINNED by tests/golden/elevation_mdp.npz.
Citations: /root/reference/source/wheeledlab_tasks/wheeledlab_tasks/elevation/mushr_elevation_env_cfg.py."""
import numpy as np

from .mathlib import F, f32, matrix_from_quat


def world_height_map(sensor_z, hit_z, root_z, offset=0.084, plane_init_value=0.19):
    """:44-48 with isaaclab mdp.height_scan = sensor_z - hit_z - offset (unpinned helper): [N,K]"""
    hs = -(f32(sensor_z)[:, None] - f32(hit_z) - F(offset))
    return (hs + (f32(root_z) - F(plane_init_value))[:, None]).astype(F)


def goal_relative_xyz(pos, cmd):
    """:50-55 -- NB the command is in the base frame, the position in the world frame (reference quirk, reproduced)"""
    rel = f32(cmd)[:, :2] - f32(pos)[:, :2]
    return np.nan_to_num(rel, nan=0.0).astype(F)


def goal_progress_rate(pos, vel_w, cmd):
    """:239-249"""
    g = f32(cmd)[:, :2] - f32(pos)[:, :2]
    v = f32(vel_w)[:, :2]
    with np.errstate(divide="ignore", invalid="ignore"):
        return (F(5) + (v * g).sum(-1) / np.sqrt((g * g).sum(-1))).astype(F)


def higher_elevation(pos, v_b):
    """:166-173"""
    z = f32(pos)[:, 2] - F(0.19)
    cond = (z > F(0.1)) & (f32(v_b)[:, 0] > F(0.1))
    return np.clip(np.where(cond, z, F(0)), F(0), F(1)).astype(F)


def is_falling_penalty(v_b, max_body_z_vel=0.10):
    """:251-254 (the second definition is the live one)"""
    return f32(v_b)[:, 2] > F(max_body_z_vel)


def forward_vel(v_b):
    """:155-157"""
    return np.minimum(f32(v_b)[:, 0], F(1.2)).astype(F)


def stuck(v_b, throttle_joint_vel, min_vel=0.02, wheel_spin_thr=5.0):
    """:342-347 (second definition); throttle_joint_vel [N,4]"""
    return np.logical_and(forward_vel(v_b) < F(min_vel), f32(throttle_joint_vel).sum(-1) > F(wheel_spin_thr))


def upright_penalty(quat, thresh_deg):
    """:217-222"""
    up = matrix_from_quat(quat)[:, 2, 2]
    ang = np.rad2deg(np.arccos(np.clip(up, -1, 1))).astype(F)
    return np.where(ang > F(thresh_deg), ang - F(thresh_deg), F(0)).astype(F)


def upright_bool(quat, thresh_deg=60.0):
    """:339-340"""
    return upright_penalty(quat, thresh_deg) > 0


def close_to_goal(pos, cmd, dist=0.5):
    """:268-273"""
    d = f32(cmd)[:, :2] - f32(pos)[:, :2]
    return np.sqrt((d * d).sum(-1)) < F(dist)


def root_height_below_minimum(pos, minimum_height=0.15):
    """isaaclab mdp.root_height_below_minimum (cfg :356-359) -- unpinned restatement"""
    return f32(pos)[:, 2] < F(minimum_height)
