"""Fused elevation env.step() -- numpy restatement of what `wl_elev_step` (+ the height-scan kernel) does, in the
order of IsaacLab's ManagerBasedRLEnv.step() with the reference's elevation plugins
(wheeledlab_tasks/elevation/mushr_elevation_env_cfg.py)."""
import numpy as np

from . import drift_mdp as M
from . import elev_mdp as E
from . import heightfield as H
from . import philox as PH
from . import vehicle as V
from .drift_step import ACT0, DAMP, EPSUM0, MASS, MU_D, MU_S, PX, QW, STEER_POS, STEER_VEL, VX, WHEEL, WX
from .mathlib import F, f32, matrix_from_quat, euler_xyz_from_quat
from .params import NS, mushr_action, mushr_vehicle

CMD_BX, CMD_BY, TGT_X, TGT_Y, TGT_H, CMD_TIMER, S_COUNT = 35, 36, 37, 38, 39, 40, 41
M_EPSUM0, M_RESETS, M_TIMEOUTS, M_TERM0, M_NONFINITE, M_EPLEN = 0, 8, 9, 10, 14, 15
S_RESET, S_CMD_RESET, S_CMD_RESAMPLE = 0, 1, 2
N_RAYS = 26
OBS_DIM = 13 + N_RAYS * N_RAYS


def elev_params():
    """MushrElevationRLEnvCfg (:438-469) flattened"""
    import math
    return NS(
        sim_dt=0.01, decimation=10, max_episode_length=math.ceil(20.0 / (0.01 * 10)),          # :461-465
        action=mushr_action(1), vehicle=mushr_vehicle(drive=1, motor_limit=0.25, substeps=1, ground_mu=(1.0, 1.0), implicit=1),
        weight=[200.0, 5000.0, 0.0, -200.0, 0.0, 0.0, 0.0, 0.0],                                # :286-305
        min_height=0.15, stuck_min_vel=0.02, stuck_wheel_spin=5.0, stuck_vel_cap=1.2,           # :356-366, :155-157
        upright_cos=math.cos(math.radians(60.0)), goal_dist=0.5,                                # :368-376
        fall_vel=0.10, elev_z0=0.19, elev_min=0.1, elev_min_vel=0.1, progress_offset=5.0,       # :166-173, :239-254
        reset_xy=19.0, reset_yaw=3.14, reset_vel=[0.1, 0.2], reset_z=0.25, spawn_clearance=0.06,  # :409-419, :147-149
        cmd_xy=19.0, cmd_heading=3.14, cmd_resample_s=10.0,                                     # :425-435
        scan_size=2.5, scan_res=0.1, scan_offset=0.084, obs_clip=10.0,                          # :74-82, :139
        log_episode_sums=1,
    )


def ground_fn(hf, probe=None):
    """probe (see vehicle.substep): `cell_margin` -- per env the smallest distance, in cells, of a wheel's sample point from a
    cell line of the grid (where the bilinear surface's normal jumps), minimised over every call"""
    h, x0, y0, cell = hf

    def g(xy):
        z, n, _ = H.sample(h, x0, y0, cell, xy[:, 0], xy[:, 1], outside=0.0)
        if probe is not None:
            u = (f32(xy[:, 0]) - F(x0)) / F(cell)
            v = (f32(xy[:, 1]) - F(y0)) / F(cell)
            d = np.minimum(np.abs(u - np.rint(u)), np.abs(v - np.rint(v)))
            probe["cell_margin"] = np.minimum(probe.get("cell_margin", np.inf), d)
        return z, n
    return g


def yaw_cs(q):
    """cos / sin of the yaw of IsaacLab's yaw_quat(q), without the atan2 round trip"""
    q = f32(q)
    a = F(1) - F(2) * (q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])
    b = F(2) * (q[:, 0] * q[:, 3] + q[:, 1] * q[:, 2])
    inv = F(1) / np.sqrt(a * a + b * b)
    return (a * inv).astype(F), (b * inv).astype(F)


def update_command(state, ids=slice(None)):
    """UniformPose2dCommand._update_command: target in the yaw-aligned base frame (isaaclab, unpinned)"""
    c, s = yaw_cs(state[QW:QW + 4, ids].T)
    dx = state[TGT_X, ids] - state[PX, ids]
    dy = state[TGT_Y, ids] - state[PX + 1, ids]
    state[CMD_BX, ids] = c * dx + s * dy
    state[CMD_BY, ids] = -s * dx + c * dy


def sym(u, a):
    return (F(2) * u - F(1)) * F(a)


def reset_envs(p, state, episode_len, hf, ids, seed, step, env_offset=0):
    if len(ids) == 0:
        return
    gid = np.asarray(ids) + env_offset
    u = PH.uniform4(gid, step, S_RESET, seed)
    x, y = sym(u[0], p.reset_xy), sym(u[1], p.reset_xy)
    zt, _, _ = H.sample(*hf, x, y, outside=0.0)
    state[PX, ids], state[PX + 1, ids] = x, y
    state[PX + 2, ids] = np.maximum(F(p.reset_z), zt + F(p.spawn_clearance))
    yaw = sym(u[2], p.reset_yaw)
    state[QW, ids], state[QW + 1, ids], state[QW + 2, ids], state[QW + 3, ids] = np.cos(yaw * F(.5)), 0, 0, np.sin(yaw * F(.5))
    state[VX:VX + 6, ids] = 0
    lo, hi = F(p.reset_vel[0]), F(p.reset_vel[1])
    c = PH.uniform4(gid, step, S_CMD_RESET, seed)
    state[VX, ids] = lo + u[3] * (hi - lo)
    state[VX + 1, ids] = lo + c[3] * (hi - lo)
    state[ACT0:ACT0 + 2, ids] = 0
    state[EPSUM0:EPSUM0 + 8, ids] = 0
    episode_len[ids] = 0
    state[TGT_X, ids], state[TGT_Y, ids] = sym(c[0], p.cmd_xy), sym(c[1], p.cmd_xy)
    state[TGT_H, ids] = sym(c[2], p.cmd_heading)
    state[CMD_TIMER, ids] = F(p.cmd_resample_s)


def height_map(p, state, hf):
    """E1: 26 x 26 yaw-aligned grid of terrain heights relative to the base plane, clipped (x fastest)"""
    n = state.shape[1]
    k = np.arange(N_RAYS, dtype=np.float32)
    g = (F(-0.5 * p.scan_size) + k * F(p.scan_res)).astype(F)
    lx = np.tile(g, N_RAYS)
    ly = np.repeat(g, N_RAYS)
    c, s = yaw_cs(state[QW:QW + 4].T)
    wx = state[PX][:, None] + c[:, None] * lx[None] - s[:, None] * ly[None]
    wy = state[PX + 1][:, None] + s[:, None] * lx[None] + c[:, None] * ly[None]
    z, _, inside = H.sample(*hf, wx.reshape(-1), wy.reshape(-1), outside=0.0)
    z, inside = z.reshape(n, -1), inside.reshape(n, -1)
    rz = state[PX + 2]
    val = E.world_height_map(rz, z, rz, p.scan_offset, p.elev_z0)
    val = np.where(inside, val, F(np.inf))
    return np.clip(val, -F(p.obs_clip), F(p.obs_clip)).astype(F)


def observe(p, state, hf):
    q = state[QW:QW + 4].T
    R = matrix_from_quat(q)
    v_b = np.einsum("nji,nj->ni", R, state[VX:VX + 3].T).astype(F)
    w_b = np.einsum("nji,nj->ni", R, state[WX:WX + 3].T).astype(F)
    goal = E.goal_relative_xyz(state[PX:PX + 3].T, state[CMD_BX:CMD_BX + 2].T)
    c = F(p.obs_clip)
    return np.concatenate([goal, np.stack(euler_xyz_from_quat(q), -1), np.clip(v_b, -c, c), np.clip(w_b, -c, c),
                           np.clip(state[ACT0:ACT0 + 2].T, F(-1), F(1)), height_map(p, state, hf)], -1).astype(F)


def step(p, state, episode_len, hf, actions, seed, step_count, metrics=None, env_offset=0, probe=None):
    n = state.shape[1]
    vp = p.vehicle
    a_raw = M.clip_action(actions) if p.action.clip_wrapper else f32(actions)
    state[ACT0:ACT0 + 2] = a_raw.T
    proc = M.process_actions(a_raw, p.action)
    steer2, wheel_t = M.fwd_targets(proc[:, 0], proc[:, 1], p.action)
    steer_t = steer2[:, 0]
    q = state[QW:QW + 4].T.copy()
    R = matrix_from_quat(q)
    cvec = f32([0, 0, vp.cg_z])
    x = (state[PX:PX + 3].T + R @ cvec).astype(F)
    v = state[VX:VX + 3].T.copy()
    wb = np.einsum("nji,nj->ni", R, state[WX:WX + 3].T).astype(F)
    wheel = state[WHEEL:WHEEL + 4].T.copy()
    th, om = state[STEER_POS].copy(), state[STEER_VEL].copy()
    h = F(p.sim_dt) / F(vp.substeps)
    g = ground_fn(hf, probe)
    for _ in range(p.decimation * vp.substeps):
        x, q, v, wb, wheel, th, om = V.substep(x, q, v, wb, wheel, th, om, steer_t, wheel_t.astype(F), state[MASS],
                                               state[MU_S], state[MU_D], state[DAMP], vp, h, g, probe)
    R = matrix_from_quat(q)
    ww = np.einsum("nij,nj->ni", R, wb).astype(F)
    pos = (x - R @ cvec).astype(F)
    state[PX:PX + 3], state[QW:QW + 4], state[VX:VX + 3], state[WX:WX + 3] = pos.T, q.T, v.T, ww.T
    state[WHEEL:WHEEL + 4] = wheel.T
    state[STEER_POS], state[STEER_VEL] = th, om

    episode_len += 1
    truncated = episode_len >= p.max_episode_length
    finite = np.isfinite(state[:19]).all(0)
    v_b = np.einsum("nji,nj->ni", R, v).astype(F)
    cmd = state[CMD_BX:CMD_BX + 2].T                     # command as left by the PREVIOUS step's command update
    t_low = E.root_height_below_minimum(pos, p.min_height)
    t_stuck = np.logical_and(np.minimum(v_b[:, 0], F(p.stuck_vel_cap)) < F(p.stuck_min_vel),
                             wheel.sum(-1) > F(p.stuck_wheel_spin))
    t_roll = R[:, 2, 2] < F(p.upright_cos)
    t_goal = E.close_to_goal(pos, cmd, p.goal_dist)
    terminated = t_low | t_stuck | t_roll | t_goal | ~finite
    with np.errstate(invalid="ignore", divide="ignore"):
        terms = np.stack([E.goal_progress_rate(pos, v, cmd), E.higher_elevation(pos, v_b),
                          E.is_falling_penalty(v_b, p.fall_vel).astype(F),
                          (t_stuck & ~truncated).astype(F)]).astype(F)
    terms = np.where(finite[None], terms, F(0)).astype(F)
    step_dt = F(p.sim_dt) * F(p.decimation)
    reward = np.zeros(n, F)
    for i in range(4):
        w = F(p.weight[i])
        if w == 0:
            continue
        c = terms[i] * w * step_dt
        reward += c
        if p.log_episode_sums:
            state[EPSUM0 + i] += c
    done = terminated | truncated
    ids = np.nonzero(done)[0]
    if metrics is not None and len(ids):
        metrics[M_EPSUM0:M_EPSUM0 + 8] += state[EPSUM0:EPSUM0 + 8, ids].astype(np.float64).sum(1)
        metrics[M_RESETS] += len(ids)
        metrics[M_TIMEOUTS] += truncated.sum()
        for k, t in enumerate((t_low, t_stuck, t_roll, t_goal)):
            metrics[M_TERM0 + k] += t.sum()
        metrics[M_NONFINITE] += (~finite).sum()
        metrics[M_EPLEN] += episode_len[ids].sum()
    if (~finite).any():
        bad = np.nonzero(~finite)[0]
        state[:19, bad] = 0
        state[QW, bad] = 1
    reset_envs(p, state, episode_len, hf, ids, seed, step_count, env_offset)
    # command manager: count down, resample expired targets, re-express every target in the base frame
    state[CMD_TIMER] -= step_dt
    exp = state[CMD_TIMER] <= 0
    if exp.any():
        u = PH.uniform4(np.arange(n) + env_offset, step_count, S_CMD_RESAMPLE, seed)
        state[TGT_X] = np.where(exp, sym(u[0], p.cmd_xy), state[TGT_X])
        state[TGT_Y] = np.where(exp, sym(u[1], p.cmd_xy), state[TGT_Y])
        state[TGT_H] = np.where(exp, sym(u[2], p.cmd_heading), state[TGT_H])
        state[CMD_TIMER] = np.where(exp, F(p.cmd_resample_s), state[CMD_TIMER])
    update_command(state)
    obs = observe(p, state, hf)
    return obs, reward.astype(F), terminated, truncated, dict(terms=terms, terms_flags=(t_low, t_stuck, t_roll, t_goal),
                                                              finite=finite)


def init_state(p, n, seed=0, stride=None, env_offset=0):
    """startup events (:387-407): wheel friction (2.0, 1.0) fixed, base mass += U(0.2, 0.5); throttle damping 1000
    (hound.py:19, not randomised in this task).  Keyed by the global env id (oracle/startup.py)."""
    from . import startup
    stride = stride or ((n + 63) // 64) * 64
    s = np.zeros((S_COUNT, stride), F)
    s[QW] = 1
    s[MU_S, :n], s[MU_D, :n], s[DAMP, :n], s[MASS, :n], _ = startup.draw(n, seed, env_offset, **startup.ELEV)
    return s
