"""Fused drift env.step() -- numpy restatement of what ONE launch of `wl_drift_step` does, in the order of
IsaacLab's ManagerBasedRLEnv.step() with the reference's plugins.  Operates on the same
SoA state matrix as the kernel (rows = include/wheeledlab_amd.h WlStateField)."""
import numpy as np

from . import drift_mdp as M
from . import philox as PH
from . import vehicle as V
from .mathlib import F, f32, matrix_from_quat

# row indices (include/wheeledlab_amd.h)
PX, QW, VX, WX, WHEEL, STEER_POS, STEER_VEL, ACT0, TIMER_HF, TIMER_LF, MU_S, MU_D, DAMP, MASS, EPSUM0, S_COUNT = \
    0, 3, 7, 10, 13, 17, 18, 19, 21, 22, 23, 24, 25, 26, 27, 35
M_EPSUM0, M_RESETS, M_TIMEOUTS, M_TERM0, M_NONFINITE, M_EPLEN, M_COUNT = 0, 8, 9, 10, 14, 15, 16


def targets(p, a_raw):
    proc = M.process_actions(a_raw, p.action)
    if p.action.map == 0:
        steer, wr = M.rwd_targets(proc[:, 0], proc[:, 1], p.action)
        wheel = np.concatenate([wr, np.zeros_like(wr)], -1)
    else:
        steer, wheel = M.fwd_targets(proc[:, 0], proc[:, 1], p.action)
        if p.action.map == 2:   # base-class AckermannAction: the joints take angles; the single-track model steers by their
            steer = np.stack([proc[:, 1], proc[:, 1]], -1)   # centre-line equivalent atan(L / R) = delta
    return steer[:, 0].astype(F), wheel.astype(F)


def reset_envs(p, state, episode_len, ref_table, ids, seed, step, env_offset=0):
    """in-kernel reset (drifting/mdp/events.py:119-133 + manager resets); ids = env indices to reset"""
    if len(ids) == 0:
        return
    gid = np.asarray(ids) + env_offset
    u = PH.uniform8(gid, step, PH.S_DRIFT_EVENTS, seed)    # [0..3] index, x, y, yaw | [4..5] timers | [6..7] the lf push's (step)
    n_ref = p.num_ref_points
    idx = np.minimum((u[0] * F(n_ref)).astype(np.int32), n_ref - 1)
    state[PX + 0, ids] = ref_table[0, idx] + (F(2) * u[1] - F(1)) * F(p.pos_noise)
    state[PX + 1, ids] = ref_table[1, idx] + (F(2) * u[2] - F(1)) * F(p.pos_noise)
    state[PX + 2, ids] = 0
    yaw = ref_table[2, idx] + (F(2) * u[3] - F(1)) * F(p.yaw_noise)
    state[QW, ids] = np.cos(yaw * F(0.5))
    state[QW + 1, ids] = 0
    state[QW + 2, ids] = 0
    state[QW + 3, ids] = np.sin(yaw * F(0.5))
    state[VX:VX + 6, ids] = 0
    state[ACT0:ACT0 + 2, ids] = 0
    state[EPSUM0:EPSUM0 + 8, ids] = 0
    episode_len[ids] = 0
    state[TIMER_HF, ids] = F(p.hf_interval[0]) + u[4] * (F(p.hf_interval[1]) - F(p.hf_interval[0]))
    state[TIMER_LF, ids] = F(p.lf_interval[0]) + u[5] * (F(p.lf_interval[1]) - F(p.lf_interval[0]))


def observe(p, state, normals):
    n = state.shape[1]
    q = state[QW:QW + 4].T
    R = matrix_from_quat(q)
    v_b = np.einsum("nji,nj->ni", R, state[VX:VX + 3].T).astype(F)
    w_b = np.einsum("nji,nj->ni", R, state[WX:WX + 3].T).astype(F)
    return M.blind_obs(p, state[PX:PX + 3].T, q, v_b, w_b, state[ACT0:ACT0 + 2].T, normals)


def step(p, state, episode_len, ref_table, actions, seed, step_count, metrics=None, noise=None, env_offset=0,
         ground=V.flat_ground):
    """state [S_COUNT, n] float32 and episode_len [n] int32 are updated IN PLACE.
    -> obs [n,14], reward [n], terminated [n] bool, truncated [n] bool, info dict"""
    n = state.shape[1]
    vp = p.vehicle
    a_raw = M.clip_action(actions) if p.action.clip_wrapper else f32(actions)
    state[ACT0:ACT0 + 2] = a_raw.T
    steer_t, wheel_t = targets(p, a_raw)

    q = state[QW:QW + 4].T.copy()
    R = matrix_from_quat(q)
    c = f32([0, 0, vp.cg_z])
    x = (state[PX:PX + 3].T + R @ c).astype(F)
    v = state[VX:VX + 3].T.copy()
    wb = np.einsum("nji,nj->ni", R, state[WX:WX + 3].T).astype(F)
    wheel = state[WHEEL:WHEEL + 4].T.copy()
    th, om = state[STEER_POS].copy(), state[STEER_VEL].copy()
    mass, mu_s, mu_d, damp = state[MASS], state[MU_S], state[MU_D], state[DAMP]
    h = F(p.sim_dt) / F(vp.substeps)
    for _ in range(p.decimation * vp.substeps):
        x, q, v, wb, wheel, th, om = V.substep(x, q, v, wb, wheel, th, om, steer_t, wheel_t, mass, mu_s, mu_d, damp,
                                               vp, h, ground)
    R = matrix_from_quat(q)
    ww = np.einsum("nij,nj->ni", R, wb).astype(F)
    pos = (x - R @ c).astype(F)
    state[PX:PX + 3] = pos.T
    state[QW:QW + 4] = q.T
    state[VX:VX + 3] = v.T
    state[WX:WX + 3] = ww.T
    state[WHEEL:WHEEL + 4] = wheel.T
    state[STEER_POS], state[STEER_VEL] = th, om

    episode_len += 1
    truncated = M.time_out(episode_len, p.max_episode_length)
    finite = np.isfinite(state[:19]).all(0)
    terminated = np.logical_or(M.cart_off_track(pos, p.straight, p.r_in, p.r_out), ~finite)

    v_b = np.einsum("nji,nj->ni", R, v).astype(F)
    steer2 = np.stack([th, th], -1)
    terms = M.drift_terms(p, pos, v_b, wb, ww, steer2, terminated, truncated)
    terms = np.where(finite[None, :], terms, F(0)).astype(F)
    reward, contrib = M.reward_sum(p, terms)
    if p.log_episode_sums:
        state[EPSUM0:EPSUM0 + 7] += contrib

    done = np.logical_or(terminated, truncated)
    ids = np.nonzero(done)[0]
    if metrics is not None and len(ids):
        metrics[M_EPSUM0:M_EPSUM0 + 8] += state[EPSUM0:EPSUM0 + 8, ids].astype(np.float64).sum(1)
        metrics[M_RESETS] += len(ids)
        metrics[M_TIMEOUTS] += truncated.sum()
        metrics[M_TERM0] += terminated.sum()
        metrics[M_NONFINITE] += (~finite).sum()
        metrics[M_EPLEN] += episode_len[ids].sum()
    if (~finite).any():  # scrub so the reset below starts from clean rows
        bad = np.nonzero(~finite)[0]
        state[:19, bad] = 0
        state[QW, bad] = 1
    reset_envs(p, state, episode_len, ref_table, ids, seed, step_count, env_offset)

    step_dt = F(p.sim_dt) * F(p.decimation)
    if p.enable_pushes:
        gid = np.arange(n) + env_offset
        state[TIMER_HF] -= step_dt
        fire = state[TIMER_HF] < F(1e-6)
        u = PH.uniform8(gid, step_count, PH.S_NOISE1, seed)[4:]       # words z, w of the block whose x, y are normals 8..11
        sym = lambda uu, a: (F(2) * uu - F(1)) * F(a)
        state[VX] += np.where(fire, sym(u[0], p.hf_vel_x), F(0))
        state[VX + 1] += np.where(fire, sym(u[1], p.hf_vel_y), F(0))
        state[WX + 2] += np.where(fire, sym(u[2], p.hf_vel_yaw), F(0))
        state[TIMER_HF] = np.where(fire, F(p.hf_interval[0]) + u[3] * (F(p.hf_interval[1]) - F(p.hf_interval[0])),
                                   state[TIMER_HF])
        state[TIMER_LF] -= step_dt
        fire = state[TIMER_LF] < F(1e-6)
        u = PH.uniform8(gid, step_count, PH.S_DRIFT_EVENTS, seed)[6:]  # word w of the event block
        state[WX + 2] += np.where(fire, sym(u[0], p.lf_vel_yaw), F(0))
        state[TIMER_LF] = np.where(fire, F(p.lf_interval[0]) + u[1] * (F(p.lf_interval[1]) - F(p.lf_interval[0])),
                                   state[TIMER_LF])

    normals = None
    if p.enable_corruption:
        normals = noise if noise is not None else PH.normal12(np.arange(n) + env_offset, step_count, seed)
    obs = observe(p, state, normals)
    return obs, reward, terminated, truncated, dict(terms=terms, done=done, finite=finite)


def init_state(p, n, seed=0, stride=None, env_offset=0):
    """startup events (mushr_drift_env_cfg.py:98-119,145-154): wheel friction U(0.3,0.5) in 20 buckets with
    mu_d <= mu_s, rear throttle damping U(10,50), base mass += U(0.3,0.5) -> fresh state matrix.  Keyed by the global
    env id (oracle/startup.py), like the product's wl_startup_randomize."""
    from . import startup
    stride = stride or ((n + 63) // 64) * 64
    s = np.zeros((S_COUNT, stride), F)
    s[QW] = 1
    s[MU_S, :n], s[MU_D, :n], s[DAMP, :n], s[MASS, :n], _ = startup.draw(n, seed, env_offset, **startup.DRIFT)
    return s
