"""Fused visual env.step() + camera observation -- numpy restatement of `wl_visual_step` (order: IsaacLab
ManagerBasedRLEnv.step() with the plugins of wheeledlab_tasks/visual/mushr_visual_env_cfg.py).
The camera is DESIGNED: the reference renders an RGB image of the black/white traversability plane with RTX
(:230-246) and augments it with torchvision (mdp_sensors/observations.py:75-87); here each pixel is one ray against
the z = 0 plane with a map lookup, followed by the same brightness / contrast / Gaussian-blur / grayscale / normalise
chain (saturation and hue jitter are identities on a grey image)."""
import math

import numpy as np

from . import drift_mdp as M
from . import philox as PH
from . import vehicle as V
from . import visual_mdp as VM
from .drift_step import ACT0, DAMP, EPSUM0, MASS, MU_D, MU_S, PX, QW, STEER_POS, STEER_VEL, VX, WHEEL, WX
from .mathlib import F, f32, matrix_from_quat
from .params import NS, mushr_action, mushr_vehicle

S_COUNT = 41
IMG_H, IMG_W, CROP = 60, 80, 20
N_PIX = (IMG_H - CROP) * IMG_W
OBS_DIM = N_PIX + 8
M_EPSUM0, M_RESETS, M_TIMEOUTS, M_TERM0, M_NONFINITE, M_EPLEN = 0, 8, 9, 10, 14, 15


def visual_params():
    """MushrVisualRLEnvCfg (:412-444)"""
    return NS(
        sim_dt=0.02, decimation=10, max_episode_length=math.ceil(10.0 / (0.02 * 10)),           # :435-439
        action=mushr_action(1), vehicle=mushr_vehicle(drive=1, motor_limit=0.25, substeps=1, ground_mu=(2.0, 2.0), implicit=1),   # h = sim.dt
        weight=[5.0, 7.0, 0, 0, 0, 0, 0, 0],                                                    # :375-385
        map_rows=500, map_cols=500, row_spacing=0.5, col_spacing=0.5,                          # :70-76
        reset_z=0.1,                                                                            # :203
        cam_pos=[0.23, 0.0, 0.18], fx=80 * 1.9299999475479126 / 3.8959999084472656,
        fy=60 * 1.9299999475479126 / 2.453000068664551, cx=40.0, cy=30.0,                      # :230-241 (pose: designed)
        sky=0.5, brightness=1.0, contrast=1.0, blur_sigma=0.0, contrast_first=0,
        log_episode_sums=1,
    )


def spawn_cells(trav):
    """indices of traversable cells in np.nonzero order: [M, 2] = (iy, ix) (utils/__init__.py:193)"""
    ys, xs = np.asarray(trav).nonzero()
    return np.stack([ys, xs], -1).astype(np.int32)


def reset_envs(p, state, episode_len, cells, ids, seed, step, env_offset=0, hf=None):
    """hf: the visual-depth extension task's heightfield terrain -- the reset pose is lifted onto it (z 0.1 above the ground)"""
    if len(ids) == 0:
        return
    gid = np.asarray(ids) + env_offset
    u = PH.uniform4(gid, step, 0, seed)
    k = np.minimum((u[0] * F(len(cells))).astype(np.int64), len(cells) - 1)
    iy, ix = cells[k, 0], cells[k, 1]
    state[PX, ids] = (ix.astype(F) - F(p.map_cols // 2)) * F(p.row_spacing)          # generate_random_poses :198-199
    state[PX + 1, ids] = (iy.astype(F) - F(p.map_rows // 2)) * F(p.col_spacing)
    state[PX + 2, ids] = F(p.reset_z)
    if hf is not None:
        from . import heightfield as H
        zt, _, _ = H.sample(hf[0], hf[1], hf[2], hf[3], state[PX, ids], state[PX + 1, ids], outside=0.0)
        state[PX + 2, ids] = (F(p.reset_z) + zt.astype(F)).astype(F)
    yaw = u[1] * F(2.0 * math.pi)                                                     # U(0, 360) deg
    state[QW, ids], state[QW + 1, ids], state[QW + 2, ids], state[QW + 3, ids] = np.cos(yaw * F(.5)), 0, 0, np.sin(yaw * F(.5))
    state[VX:VX + 6, ids] = 0
    state[ACT0:ACT0 + 2, ids] = 0
    state[EPSUM0:EPSUM0 + 8, ids] = 0
    episode_len[ids] = 0


def gaussian_kernel5(sigma):
    x = np.linspace(-2, 2, 5)
    k = np.exp(-0.5 * (x / sigma) ** 2)
    return (k / k.sum()).astype(F)


def camera(p, state, trav):
    """-> [n, 3200] normalised grey image rows (cropped 40 x 80, row-major)"""
    n = state.shape[1]
    R = matrix_from_quat(state[QW:QW + 4].T)
    o = state[PX:PX + 3].T + np.einsum("nij,j->ni", R, f32(p.cam_pos))
    rows = np.arange(CROP, IMG_H, dtype=np.float32)
    cols = np.arange(IMG_W, dtype=np.float32)
    dy = -((cols + F(0.5) - F(p.cx)) / F(p.fx))               # body y (left positive) of the ray, x_cam = right
    dz = -((rows + F(0.5) - F(p.cy)) / F(p.fy))               # body z, y_cam = down
    db = np.stack([np.ones((len(rows), len(cols)), F), np.broadcast_to(dy[None, :], (len(rows), len(cols))),
                   np.broadcast_to(dz[:, None], (len(rows), len(cols)))], -1).reshape(-1, 3)
    dw = np.einsum("nij,pj->npi", R, db).astype(F)            # [n, pix, 3]
    hit = dw[..., 2] < F(-1e-6)
    t = np.where(hit, -o[:, None, 2] / np.where(hit, dw[..., 2], F(-1)), F(0)).astype(F)
    hx = o[:, None, 0] + t * dw[..., 0]
    hy = o[:, None, 1] + t * dw[..., 1]
    half_w, half_h = F(p.map_rows * p.row_spacing / 2), F(p.map_cols * p.col_spacing / 2)
    on_map = hit & (np.abs(hx) <= half_w) & (np.abs(hy) <= half_h)
    tv = VM.get_traversability(trav, np.stack([hx.reshape(-1), hy.reshape(-1)], -1), num_rows=p.map_rows,
                               num_cols=p.map_cols, row_spacing=p.row_spacing, col_spacing=p.col_spacing).reshape(n, -1)
    img = np.where(hit, np.where(on_map & tv, F(1), F(0)), F(p.sky)).astype(F)
    # torchvision ColorJitter.forward: the ops run in the order of a torch.randperm(4) drawn per call (mdp_sensors/observations.py:21,
    # 82: one call per batch).  adjust_brightness: blend with zeros = clamp(b x); adjust_contrast: blend with the mean of the
    # grayscale image, clamp.  Hue is the identity on a grey image, saturation a scale within 1.8e-4 of 1 (tests/test_oracle_
    # golden_elev_visual.py::test_saturation_and_hue_on_a_grey_image): only the brightness / contrast order is modelled.
    def brightness(x):
        return np.clip(x * F(p.brightness), 0, 1)

    def contrast(x):
        if p.contrast == 1.0:
            return x
        mean = (x * F(0.9999)).mean(-1, keepdims=True, dtype=np.float32)    # grayscale mean (0.2989+0.587+0.114)
        return np.clip(F(p.contrast) * x + (F(1) - F(p.contrast)) * mean, 0, 1)
    img = brightness(contrast(img)) if getattr(p, "contrast_first", 0) else contrast(brightness(img))
    if p.blur_sigma > 0:
        k = gaussian_kernel5(p.blur_sigma)
        im = img.reshape(n, IMG_H - CROP, IMG_W)
        pad = np.pad(im, ((0, 0), (0, 0), (2, 2)), mode="reflect")
        im = sum(k[j] * pad[:, :, j:j + IMG_W] for j in range(5))
        pad = np.pad(im, ((0, 0), (2, 2), (0, 0)), mode="reflect")
        im = sum(k[j] * pad[:, j:j + IMG_H - CROP, :] for j in range(5))
        img = im.reshape(n, -1)
    return ((img * F(0.9999) - F(0.5)) / F(0.5)).astype(F)


def observe(p, state, trav):
    R = matrix_from_quat(state[QW:QW + 4].T)
    v_b = np.einsum("nji,nj->ni", R, state[VX:VX + 3].T).astype(F)
    w_b = np.einsum("nji,nj->ni", R, state[WX:WX + 3].T).astype(F)
    return np.concatenate([camera(p, state, trav), v_b, w_b, np.clip(state[ACT0:ACT0 + 2].T, F(-1), F(1))], -1).astype(F)


def observe_depth(p, state, hf, max_depth):
    """observation of the visual-depth extension task (BASELINE config 5): distance_to_image_plane 60 x 80 against the heightfield
    (oracle/depth.c) | base_lin_vel | base_ang_vel | last_action"""
    from . import depth as D
    R = matrix_from_quat(state[QW:QW + 4].T)
    v_b = np.einsum("nji,nj->ni", R, state[VX:VX + 3].T).astype(F)
    w_b = np.einsum("nji,nj->ni", R, state[WX:WX + 3].T).astype(F)
    img = D.depth(p, state[PX:PX + 3].T.copy(), state[QW:QW + 4].T.copy(), hf, max_depth).reshape(state.shape[1], -1)
    return np.concatenate([img, v_b, w_b, np.clip(state[ACT0:ACT0 + 2].T, F(-1), F(1))], -1).astype(F)


def step(p, state, episode_len, trav, cells, actions, seed, step_count, metrics=None, env_offset=0, hf=None, max_depth=None, probe=None):
    """hf / max_depth: the visual-depth extension task -- the same step on the heightfield terrain `hf` (wheel contacts through
    heightfield.sample, reset onto the terrain), observation = observe_depth"""
    n = state.shape[1]
    vp = p.vehicle
    a_raw = M.clip_action(actions) if p.action.clip_wrapper else f32(actions)
    state[ACT0:ACT0 + 2] = a_raw.T
    proc = M.process_actions(a_raw, p.action)
    steer2, wheel_t = M.fwd_targets(proc[:, 0], proc[:, 1], p.action)
    q = state[QW:QW + 4].T.copy()
    R = matrix_from_quat(q)
    cvec = f32([0, 0, vp.cg_z])
    x = (state[PX:PX + 3].T + R @ cvec).astype(F)
    v = state[VX:VX + 3].T.copy()
    wb = np.einsum("nji,nj->ni", R, state[WX:WX + 3].T).astype(F)
    wheel = state[WHEEL:WHEEL + 4].T.copy()
    th, om = state[STEER_POS].copy(), state[STEER_VEL].copy()
    h = F(p.sim_dt) / F(vp.substeps)
    ground = V.flat_ground
    if hf is not None:
        from .elev_step import ground_fn
        ground = ground_fn(hf, probe)
    for _ in range(p.decimation * vp.substeps):
        x, q, v, wb, wheel, th, om = V.substep(x, q, v, wb, wheel, th, om, steer2[:, 0], wheel_t.astype(F), state[MASS],
                                               state[MU_S], state[MU_D], state[DAMP], vp, h, ground, probe)
    R = matrix_from_quat(q)
    ww = np.einsum("nij,nj->ni", R, wb).astype(F)
    pos = (x - R @ cvec).astype(F)
    state[PX:PX + 3], state[QW:QW + 4], state[VX:VX + 3], state[WX:WX + 3] = pos.T, q.T, v.T, ww.T
    state[WHEEL:WHEEL + 4] = wheel.T
    state[STEER_POS], state[STEER_VEL] = th, om
    episode_len += 1
    truncated = episode_len >= p.max_episode_length
    finite = np.isfinite(state[:19]).all(0)
    width, height = p.map_rows * p.row_spacing, p.map_cols * p.col_spacing
    oom = VM.out_of_map(pos, width, height)
    terminated = oom | ~finite
    v_b = np.einsum("nji,nj->ni", R, v).astype(F)
    safe = np.where(finite[:, None], pos, 0).astype(F)
    kw = dict(num_rows=p.map_rows, num_cols=p.map_cols, row_spacing=p.row_spacing, col_spacing=p.col_spacing)
    terms = np.stack([np.where(VM.get_traversability(trav, safe[:, :2], **kw), F(1), F(-1)), v_b[:, 0]]).astype(F)
    terms = np.where(finite[None], terms, F(0)).astype(F)
    step_dt = F(p.sim_dt) * F(p.decimation)
    reward = np.zeros(n, F)
    for i in range(2):
        w = F(p.weight[i])
        if w == 0:
            continue
        c = terms[i] * w * step_dt
        reward += c
        if p.log_episode_sums:
            state[EPSUM0 + i] += c
    ids = np.nonzero(terminated | truncated)[0]
    if metrics is not None and len(ids):
        metrics[M_EPSUM0:M_EPSUM0 + 8] += state[EPSUM0:EPSUM0 + 8, ids].astype(np.float64).sum(1)
        metrics[M_RESETS] += len(ids)
        metrics[M_TIMEOUTS] += truncated.sum()
        metrics[M_TERM0] += (oom & finite).sum()
        metrics[M_NONFINITE] += (~finite).sum()
        metrics[M_EPLEN] += episode_len[ids].sum()
    if (~finite).any():
        bad = np.nonzero(~finite)[0]
        state[:19, bad] = 0
        state[QW, bad] = 1
    reset_envs(p, state, episode_len, cells, ids, seed, step_count, env_offset, hf)
    obs = observe(p, state, trav) if hf is None else observe_depth(p, state[:, :n], hf, max_depth)
    return obs, reward.astype(F), terminated, truncated, dict(terms=terms, finite=finite)


def init_state(p, n, seed=0, stride=None, wheel_mu=(0.5, 0.5), mass=3.0):
    """VisualEventsCfg (:253-262) has only the reset event: no randomisation; wheel material = PhysX default 0.5/0.5
    (the USD's own material is unknown), throttle damping 1000 (hound.py:19)"""
    stride = stride or ((n + 63) // 64) * 64
    s = np.zeros((S_COUNT, stride), F)
    s[QW] = 1
    s[MU_S], s[MU_D], s[DAMP], s[MASS] = wheel_mu[0], wheel_mu[1], 1000.0, mass
    return s
