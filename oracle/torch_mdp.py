"""torch-CPU restatement of the drift mdp path, op for op in the style the reference executes it (many small
elementwise torch ops over [N,k] tensors).  Used ONLY as `bench.py`'s cpu_baseline ("port") -- the reference's own
torch functions cannot travel to the GPU box -- and pinned to the golden vectors by tests/test_oracle_torch_mdp.py.
Citations: wheeledlab_tasks/drifting/mushr_drift_env_cfg.py unless noted."""
import math

import torch


def process_actions(a, scale, offset):                     # clip_action.py:27 + ackermann_actions.py:119-133
    a = torch.clip(a, -1.0, 1.0)
    b = torch.clip(a, min=-1.0, max=1.0) * scale + offset
    b[:, 0] = torch.clamp(b[:, 0], min=0.0)
    return a, b


def rwd_targets(v, delta, r=0.05):                         # rc_car_actions.py:12-29
    t = torch.tan(delta)
    w = v / r
    return torch.stack([t, t], dim=1), torch.stack([w, w], dim=1)


def off_track(pos, s, ro):                                 # :210-217
    return torch.where(torch.abs(pos[..., 1]) < s, torch.where(torch.abs(pos[..., 0]) > ro, 1, 0),
                       torch.where(pos[..., 1] > 0,
                                   torch.where((pos[..., 1] - s) ** 2 + pos[..., 0] ** 2 > ro ** 2, 1, 0),
                                   torch.where((pos[..., 1] + s) ** 2 + pos[..., 0] ** 2 > ro ** 2, 1, 0)))


def in_range(pos, s, ri):                                  # :201-208
    return torch.where(torch.abs(pos[..., 1]) < s, torch.where(torch.abs(pos[..., 0]) < ri, 1, 0),
                       torch.where(pos[..., 1] > 0,
                                   torch.where((pos[..., 1] - s) ** 2 + pos[..., 0] ** 2 < ri ** 2, 1, 0),
                                   torch.where((pos[..., 1] + s) ** 2 + pos[..., 0] ** 2 < ri ** 2, 1, 0)))


def cart_off_track(pos, s=0.8, ri=0.3, ro=2.0):            # :343-348
    return torch.logical_or(off_track(pos, s, ro) > 0.5, in_range(pos, s, ri) > 0.5)


def side_slip(vb, lo=0.25, hi=0.55, min_vx=1.0):           # :219-230
    ang = torch.abs(torch.atan2(vb[..., 1], vb[..., 0]))
    ang = torch.where(torch.logical_or(torch.abs(vb[..., 0]) < min_vx, ang > hi), 0.0, ang)
    return torch.where(ang < lo, 0.0, ang)


def vel_dist(vb, target=3.0, offset=-9.0):                 # :167-171
    return (torch.norm(vb[..., :2], dim=-1) - target) ** 2 + offset


def turn_left_go_right(steer, wb, th=1.0):                 # :232-240
    return torch.clamp(steer.mean(dim=-1) * torch.clamp(wb[..., 2], max=th, min=-th) * -1.0, min=0.0)


def energy_through_turn(pos, vb, s=0.8):                   # :195-199
    return torch.where(torch.abs(pos[..., 1]) > s, torch.norm(vb, dim=-1) ** 2, 0.0)


def cross_track_dist(pos, s=0.8, r=0.8, offset=-1.0, p=1.0):   # :173-193
    sq = torch.where(torch.abs(pos[..., 1]) < s,
                     torch.where(pos[..., 0] > 0, (pos[..., 0] - r) ** 2, (pos[..., 0] + r) ** 2),
                     torch.where(pos[..., 1] > 0,
                                 (torch.sqrt((pos[..., 1] - s) ** 2 + pos[..., 0] ** 2) - r) ** 2,
                                 (torch.sqrt((pos[..., 1] + s) ** 2 + pos[..., 0] ** 2) - r) ** 2))
    return torch.pow(torch.sqrt(sq) + offset, p)


def euler_xyz(q):                                          # wheeledlab/envs/mdp/observations.py:9-12 (IsaacLab math)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = torch.atan2(2.0 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    sp = 2.0 * (w * y - z * x)
    pitch = torch.where(torch.abs(sp) >= 1, torch.copysign(torch.full_like(sp, math.pi / 2), sp), torch.asin(sp))
    yaw = torch.atan2(2.0 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return torch.stack([roll % (2 * math.pi), pitch % (2 * math.pi), yaw % (2 * math.pi)], dim=-1)


def mdp_step(pos, quat, vb, wb, ww, steer, actions, ep_len, weights, scale, offset, noise_std=(0.1, 0.1, 0.5, 0.4),
             step_dt=0.02, max_len=250):
    """one pass of the drift mdp path on given state tensors (no physics exists on the reference's CPU side):
    action term -> terminations -> 7 rewards (RewardManager) -> 14-dim noisy observation (ObservationManager)"""
    a_raw, proc = process_actions(actions, scale, offset)
    steer_t, wheel_t = rwd_targets(proc[:, 0], proc[:, 1])
    truncated = ep_len >= max_len
    terminated = cart_off_track(pos)
    terms = (side_slip(vb), vel_dist(vb), ww[..., 2], turn_left_go_right(steer, wb), energy_through_turn(pos, vb),
             cross_track_dist(pos), (terminated * (~truncated)).float())
    rew = torch.zeros(pos.shape[0])
    for w, t in zip(weights, terms):
        if w == 0.0:
            continue
        rew += t * w * step_dt
    parts = [pos + noise_std[0] * torch.randn_like(pos), euler_xyz(quat) + noise_std[1] * torch.randn(pos.shape[0], 3),
             vb + noise_std[2] * torch.randn_like(vb), wb + noise_std[3] * torch.randn_like(wb),
             torch.clip(a_raw, -1.0, 1.0)]
    obs = torch.cat(parts, dim=-1)
    return obs, rew, terminated, truncated, steer_t, wheel_t
