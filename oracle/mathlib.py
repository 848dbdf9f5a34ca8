"""Quaternion / Euler helpers, numpy float32.  PARITY UNPINNED: these restate the published IsaacLab v2.0.2
`isaaclab.utils.math` definitions (un-vendored dependency, reference README.md:31) which the reference calls at
wheeledlab/envs/mdp/observations.py:11 (euler_xyz_from_quat), drifting/mdp/events.py:130 (quat_from_euler_xyz),
elevation/mushr_elevation_env_cfg.py:218 (matrix_from_quat).  Quaternions are (w, x, y, z)."""
import numpy as np

F = np.float32
TWO_PI = F(2.0 * np.pi)


def f32(a):
    return np.asarray(a, dtype=np.float32)


def euler_xyz_from_quat(q):
    """-> roll, pitch, yaw, each wrapped to [0, 2pi) (IsaacLab 2.0.2 behaviour)."""
    q = f32(q)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    roll = np.arctan2(F(2) * (w * x + y * z), F(1) - F(2) * (x * x + y * y))
    sp = F(2) * (w * y - z * x)
    pitch = np.where(np.abs(sp) >= 1, np.copysign(F(np.pi / 2), sp), np.arcsin(np.clip(sp, -1, 1)))
    yaw = np.arctan2(F(2) * (w * z + x * y), F(1) - F(2) * (y * y + z * z))
    return (np.mod(roll, TWO_PI).astype(F), np.mod(pitch, TWO_PI).astype(F), np.mod(yaw, TWO_PI).astype(F))


def quat_from_euler_xyz(roll, pitch, yaw):
    roll, pitch, yaw = f32(roll), f32(pitch), f32(yaw)
    cy, sy = np.cos(yaw * F(0.5)), np.sin(yaw * F(0.5))
    cr, sr = np.cos(roll * F(0.5)), np.sin(roll * F(0.5))
    cp, sp = np.cos(pitch * F(0.5)), np.sin(pitch * F(0.5))
    return np.stack([cy * cr * cp + sy * sr * sp, cy * sr * cp - sy * cr * sp,
                     cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp], -1).astype(F)


def matrix_from_quat(q):
    """unit quaternion -> rotation matrix [..., 3, 3] (body -> world)."""
    q = f32(q)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two = F(2)
    m = np.stack([
        1 - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w),
        two * (x * y + z * w), 1 - two * (x * x + z * z), two * (y * z - x * w),
        two * (x * z - y * w), two * (y * z + x * w), 1 - two * (x * x + y * y)], -1)
    return m.reshape(q.shape[:-1] + (3, 3)).astype(F)


def rotate(q, v):
    """world = R(q) body"""
    return np.einsum("...ij,...j->...i", matrix_from_quat(q), f32(v)).astype(F)


def rotate_inverse(q, v):
    """body = R(q)^T world  (== IsaacLab quat_rotate_inverse, behind root_lin_vel_b / root_ang_vel_b)"""
    return np.einsum("...ji,...j->...i", matrix_from_quat(q), f32(v)).astype(F)
