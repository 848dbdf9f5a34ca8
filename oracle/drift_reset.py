"""Reset-along-track event and the reward-weight curriculum.  PARITY PINNED by tests/golden/reset_track.npz and
curriculum.npz."""
import math

import numpy as np

from .mathlib import F, f32, quat_from_euler_xyz


def reference_poses(u, track_radius=0.8, straight=0.8):
    """reset_root_state_along_track.generate_reference_poses,
    wheeledlab_tasks/wheeledlab_tasks/drifting/mdp/events.py:33-100.
    u: the `num_points` uniforms in [0,1) the reference draws with torch.rand -> [num_points, 2, 3] (pos | euler deg)"""
    r, s = F(track_radius), F(straight)
    dist_track = F(2.0) * F(math.pi) * r + F(4.0) * s
    d = f32(u) * dist_track
    z = np.zeros_like(d)
    pi = F(math.pi)
    a1 = (d - 2 * s) / r
    rem = d - 2 * s - pi * r
    a2 = (d - 4 * s - pi * r) / r
    c1p = np.stack([np.full_like(d, r), d - s, z], -1)
    c1o = np.stack([z, z, np.full_like(d, 90)], -1)
    c2p = np.stack([r * np.cos(a1), s + r * np.sin(a1), z], -1)
    c2o = np.stack([z, z, 90 + a1 * 180 / pi], -1)
    c3p = np.stack([np.full_like(d, -r), s - rem, z], -1)
    c3o = np.stack([z, z, np.full_like(d, 270)], -1)
    c4p = np.stack([-r * np.cos(a2), -s - r * np.sin(a2), z], -1)
    c4o = np.stack([z, z, 270 + a2 * 180 / pi], -1)
    m1 = (d < 2 * s)[:, None]
    m2 = (d < 2 * s + pi * r)[:, None]
    m3 = (d < 4 * s + pi * r)[:, None]
    pos = np.where(m1, c1p, np.where(m2, c2p, np.where(m3, c3p, c4p)))
    ori = np.where(m1, c1o, np.where(m2, c2o, np.where(m3, c3o, c4o)))
    return np.stack([pos, ori], 1).astype(F)


def reset_pose(ref_poses, idx, u_xy, u_yaw, pos_noise, yaw_noise):
    """reset_root_state_along_track.__call__, events.py:119-133.
    idx int [M]; u_xy [M,2], u_yaw [M] uniforms in [0,1) -> pose [M,7] (pos | quat wxyz), vel [M,6] = 0"""
    ref = f32(ref_poses)[np.asarray(idx)]
    xy = (F(2) * f32(u_xy) - F(1)) * F(pos_noise)
    pos = ref[:, 0, :] + np.concatenate([xy, np.zeros_like(xy[:, :1])], -1)
    yn = (F(2) * f32(u_yaw) - F(1)) * F(yaw_noise)
    rpy = np.deg2rad(ref[:, 1, :]).astype(F)
    q = quat_from_euler_xyz(rpy[:, 0], rpy[:, 1], rpy[:, 2] + yn)
    return np.concatenate([pos, q], -1).astype(F), np.zeros((len(idx), 6), F)


def ref_pose_table(ref_poses):
    """[P,2,3] (pos | euler deg) -> the [3][32] device table (x, y, yaw rad) used in-kernel"""
    P = ref_poses.shape[0]
    assert P <= 32
    t = np.zeros((3, 32), F)
    t[0, :P], t[1, :P] = ref_poses[:, 0, 0], ref_poses[:, 0, 1]
    t[2, :P] = np.deg2rad(f32(ref_poses[:, 1, 2])).astype(F)
    return t


def increase_reward_weight_over_time(common_step_counter, max_episode_length, weight, increase,
                                     episodes_per_increase=1, max_increases=math.inf):
    """wheeledlab/wheeledlab/envs/mdp/curriculums.py:10-35 -> new weight (evaluated on steps where >= 1 env resets)"""
    num_episodes = common_step_counter // max_episode_length
    num_increases = num_episodes // episodes_per_increase
    if num_increases > max_increases:
        return weight
    if common_step_counter % max_episode_length != 0:
        return weight
    if (num_episodes + 1) % episodes_per_increase == 0:
        return weight + increase
    return weight
