"""mdp term library: the names the reference's task configs import from `isaaclab.envs.mdp`, `wheeledlab.envs.mdp`
and define at module level in their cfg files -- here as *kernel-backed* callables.

Each built-in term `f(env, **params) -> Tensor[N]` carries `wl_kernel_slot` metadata: when it appears in a task
config the flattening step (`flatten.py`) turns (func, params, weight) into fields of the fused kernel's parameter
struct, so the hot path never calls it.  Called directly (user code, tests), it evaluates through the terms-only HIP
entry point `wl_drift_mdp` on the env's current state -- there is one implementation of the arithmetic, in csrc/."""
from __future__ import annotations

import math

import torch

from .. import _abi as A
from .managers_cfg import ManagerTermBase, SceneEntityCfg

_ROBOT = SceneEntityCfg("robot")

# ---- state accessors (isaaclab.envs.mdp public API used by the reference) ---------------------------------------


def root_pos_w(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.root_pos_w - env.scene.env_origins


def root_quat_w(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.root_quat_w


def root_lin_vel_w(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.root_lin_vel_w


def base_lin_vel(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.root_lin_vel_b


def base_ang_vel(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.root_ang_vel_b


def joint_pos(env, asset_cfg=_ROBOT):
    return env.scene[asset_cfg.name].data.joint_pos[:, asset_cfg.joint_ids]


def joint_vel(env, asset_cfg=_ROBOT):
    if asset_cfg.joint_names is not None and isinstance(asset_cfg.joint_ids, slice):
        asset_cfg.resolve(env.scene)
    return env.scene[asset_cfg.name].data.joint_vel[:, asset_cfg.joint_ids]


def last_action(env, action_name=None):
    return env.action_manager.action


def generated_commands(env, command_name):
    return env.command_manager.get_command(command_name)


def euler_xyz_from_quat(q):
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = torch.atan2(2.0 * (w * x + y * z), 1 - 2 * (x * x + y * y))
    sp = 2.0 * (w * y - z * x)
    pitch = torch.where(sp.abs() >= 1, torch.copysign(torch.full_like(sp, math.pi / 2), sp), torch.asin(sp.clamp(-1, 1)))
    yaw = torch.atan2(2.0 * (w * z + x * y), 1 - 2 * (y * y + z * z))
    return roll % (2 * math.pi), pitch % (2 * math.pi), yaw % (2 * math.pi)


def root_euler_xyz(env, asset_cfg=_ROBOT):
    """wheeledlab.envs.mdp.root_euler_xyz (wheeledlab/envs/mdp/observations.py:9-12)"""
    return torch.stack(euler_xyz_from_quat(root_quat_w(env, asset_cfg)), dim=-1)


def _kernel_term(kind, slot, param_map=None):
    """decorator: tag a built-in term with its slot in the fused kernel (+ cfg-param -> struct-field map)"""
    def deco(fn):
        fn.wl_kind, fn.wl_slot, fn.wl_params = kind, slot, dict(param_map or {})
        return fn
    return deco


def _drift_row(env, slot, overrides):
    """evaluate one unweighted drift reward term through wl_drift_mdp on the current state"""
    return env._eval_drift_terms(overrides)[0][slot]


# ---- drift task reward terms (reference: wheeledlab_tasks/drifting/mushr_drift_env_cfg.py:160-240) --------------

@_kernel_term("reward", 0, {"min_thresh": "slip_min", "max_thresh": "slip_max", "min_vel_x": "slip_min_vx"})
def side_slip(env, min_thresh: float, max_thresh: float, min_vel_x: float = 0.5):
    return _drift_row(env, 0, dict(slip_min=min_thresh, slip_max=max_thresh, slip_min_vx=min_vel_x))


@_kernel_term("reward", 1, {"speed_target": "speed_target", "offset": "speed_offset"})
def vel_dist(env, speed_target: float = 3.0, offset: float = -9.0):
    return _drift_row(env, 1, dict(speed_target=speed_target, speed_offset=offset))


@_kernel_term("reward", 2)
def track_progress_rate(env):
    return _drift_row(env, 2, {})


@_kernel_term("reward", 3, {"ang_vel_thresh": "tlgr_thresh"})
def turn_left_go_right(env, ang_vel_thresh: float = math.pi / 4):
    return _drift_row(env, 3, dict(tlgr_thresh=ang_vel_thresh))


@_kernel_term("reward", 4, {"straight": "straight"})
def energy_through_turn(env, straight: float):
    return _drift_row(env, 4, dict(straight=straight))


@_kernel_term("reward", 5, {"straight": "straight", "track_radius": "r_line", "offset": "ctd_offset", "p": "ctd_p"})
def cross_track_dist(env, straight: float, track_radius: float = 1.15, offset: float = -1.0, p: float = 1.0):
    return _drift_row(env, 5, dict(straight=straight, r_line=track_radius, ctd_offset=offset, ctd_p=p))


class is_terminated_term(ManagerTermBase):
    """isaaclab.envs.mdp.rewards.is_terminated_term: 1 where one of `term_keys` fired and the env did not time out"""
    wl_kind, wl_slot, wl_params = "reward", 6, {}

    def __call__(self, env, term_keys=".*"):
        tm = env.termination_manager
        return (tm.terminated & ~tm.time_outs).float()


class rewards:  # namespace alias: the reference spells it `mdp.rewards.is_terminated_term`
    is_terminated_term = is_terminated_term


# ---- terminations ------------------------------------------------------------------------------------------------

@_kernel_term("termination", "time_out")
def time_out(env):
    return env.episode_length_buf >= env.max_episode_length


@_kernel_term("termination", 0, {"straight": "straight", "corner_in_radius": "r_in", "corner_out_radius": "r_out"})
def cart_off_track(env, straight: float, corner_in_radius: float, corner_out_radius: float):
    return env._eval_drift_terms(dict(straight=straight, r_in=corner_in_radius, r_out=corner_out_radius))[1]


def off_track(env, straight, corner_out_radius):
    """mushr_drift_env_cfg.py:210-217 -- int 0/1; only the outer bound (inner radius disabled).  The visual cfg module
    defines the same function (visual/mushr_visual_env_cfg.py:352-359): on a non-drift env it is evaluated with torch."""
    if getattr(env, "_task", None) == "drift":
        return env._eval_drift_terms(dict(straight=straight, r_in=0.0, r_out=corner_out_radius))[1].long()
    pos = root_pos_w(env)
    x, y = pos[..., 0], pos[..., 1]
    corner = (y - torch.where(y > 0, straight, -straight)) ** 2 + x ** 2 > corner_out_radius ** 2
    return torch.where(y.abs() < straight, x.abs() > corner_out_radius, corner).long()


def in_range(env, straight, corner_in_radius):
    """mushr_drift_env_cfg.py:201-208 -- int 0/1; only the inner bound (outer radius disabled)"""
    return env._eval_drift_terms(dict(straight=straight, r_in=corner_in_radius, r_out=1e30))[1].long()


# ---- events (startup / reset / interval): markers consumed by flatten.py ------------------------------------------

def _event(kind):
    def deco(fn):
        fn.wl_kind, fn.wl_event = "event", kind
        return fn
    return deco


class reset_root_state_along_track(ManagerTermBase):
    """drifting/mdp/events.py:10-133 -- executed inside the fused kernel (and by wl_drift_reset); this class only
    carries the parameters and exposes the pre-sampled `reference_poses` like the reference's term does."""
    wl_kind, wl_event = "event", "reset_along_track"

    def __init__(self, cfg, env):
        super().__init__(cfg, env)
        self.num_points = cfg.params.get("num_points", 20)

    @property
    def reference_poses(self):
        t = self._env._batch.ref_table[:, : self.num_points]  # x, y, yaw(rad)
        pos = torch.stack([t[0], t[1], torch.zeros_like(t[0])], -1)
        ori = torch.stack([torch.zeros_like(t[0]), torch.zeros_like(t[0]), torch.rad2deg(t[2])], -1)
        return torch.stack([pos, ori], dim=1)

    def __call__(self, env, env_ids, track_radius=0.8, track_straight_dist=0.8, num_points=20, asset_cfg=_ROBOT,
                 pos_noise=0.0, yaw_noise=0.0):
        mask = torch.zeros(env.num_envs, dtype=torch.uint8, device=env.device)
        mask[env_ids] = 1
        env._batch.reset(mask)


@_event("wheel_friction")
def randomize_rigid_body_material(env, env_ids, static_friction_range, dynamic_friction_range, restitution_range,
                                  num_buckets, asset_cfg=None, make_consistent=False):
    raise RuntimeError("startup event: applied by the env constructor (flatten.py), not callable per step")


@_event("actuator_gains")
def randomize_actuator_gains(env, env_ids, asset_cfg=None, stiffness_distribution_params=None,
                             damping_distribution_params=None, operation="abs", distribution="uniform"):
    raise RuntimeError("startup event: applied by the env constructor (flatten.py), not callable per step")


@_event("base_mass")
def randomize_rigid_body_mass(env, env_ids, asset_cfg=None, mass_distribution_params=None, operation="add",
                              distribution="uniform", recompute_inertia=True):
    raise RuntimeError("startup event: applied by the env constructor (flatten.py), not callable per step")


@_event("push")
def push_by_setting_velocity(env, env_ids, velocity_range, asset_cfg=_ROBOT):
    """isaaclab mdp.push_by_setting_velocity: root vel (world) += U(range); interval pushes run in-kernel, this
    direct call is for user code"""
    b = env._batch
    n = len(env_ids) if not isinstance(env_ids, slice) else b.n
    for key, row in (("x", A.S_VX), ("y", A.S_VY), ("z", A.S_VZ), ("roll", A.S_WX), ("pitch", A.S_WY), ("yaw", A.S_WZ)):
        lo, hi = velocity_range.get(key, (0.0, 0.0))
        if lo != 0.0 or hi != 0.0:
            b.state[row, env_ids] += torch.rand(n, device=b.device) * (hi - lo) + lo


# ---- curriculum (wheeledlab/envs/mdp/curriculums.py:10-35) ----------------------------------------------------------

def increase_reward_weight_over_time(env, env_ids, reward_term_name: str, increase: float,
                                     episodes_per_increase: int = 1, max_increases=math.inf):
    """Raise a reward weight every `episodes_per_increase` episodes (host scalar logic, evaluated on steps where at
    least one env resets).  NB: performs max_increases + 1 increments, like the reference (`>` not `>=`)."""
    num_episodes = env.common_step_counter // env.max_episode_length
    num_increases = num_episodes // episodes_per_increase
    if num_increases > max_increases:
        return
    if env.common_step_counter % env.max_episode_length != 0:
        return
    if (num_episodes + 1) % episodes_per_increase == 0:
        term_cfg = env.reward_manager.get_term_cfg(reward_term_name)
        term_cfg.weight += increase
        env.reward_manager.set_term_cfg(reward_term_name, term_cfg)


# =====================================================================================================================
# Elevation task terms (reference: wheeledlab_tasks/elevation/mushr_elevation_env_cfg.py) -- kernel-backed markers.
# Direct calls evaluate through `wl_elev_mdp` on the env's current state.
# =====================================================================================================================

def _elev(env):
    return env._eval_elev_terms()


@_kernel_term("reward", 0)
def goal_progress_rate(env):                                    # :239-249
    return _elev(env)["terms"][0]


@_kernel_term("reward", 1)
def higher_elevation(env):                                      # :166-173
    return _elev(env)["terms"][1]


@_kernel_term("reward", 2, {"max_body_z_vel": "fall_vel"})
def is_falling_penalty(env, max_body_z_vel: float = 0.10):      # :251-254 (the live, second definition)
    return _elev(env)["terms"][2] > 0.5


def forward_vel_capped(env, cap: float = 1.2):                  # elevation :155-157 (used inside `stuck`)
    return torch.clamp(base_lin_vel(env)[:, 0], max=cap)


@_kernel_term("termination", 0, {"minimum_height": "min_height"})
def root_height_below_minimum(env, minimum_height: float, asset_cfg=_ROBOT):   # isaaclab mdp (cfg :356-359)
    return _elev(env)["flags"][0]


@_kernel_term("termination", 1, {"min_vel": "stuck_min_vel", "wheel_spin_thr": "stuck_wheel_spin"})
def stuck(env, min_vel, wheel_spin_thr):                        # :342-347
    return _elev(env)["flags"][1]


@_kernel_term("termination", 2, {"thresh_deg": "upright_cos"})
def upright_bool(env, thresh_deg):                              # :339-340 via upright_penalty :217-222
    return _elev(env)["flags"][2]


@_kernel_term("termination", 3, {"dist": "goal_dist"})
def close_to_goal(env, dist):                                   # :268-273
    return _elev(env)["flags"][3]


def goal_relative_xyz(env):                                     # :50-55
    return _elev(env)["goal_rel"]


def world_height_map(env, sensor_cfg=None, offset: float = 0.084, plane_init_value: float = 0.19):   # :44-48
    return env._batch.observe()[:, 13:]


def height_scan(env, sensor_cfg=None, offset: float = 0.5):
    """isaaclab mdp.height_scan: sensor height - hit height - offset, [N, rays] (un-vendored).  The
    hits come from the scan kernel through the scene's sensor view (envs/scene.py::RayCasterData)."""
    sensor = env.scene.sensors[sensor_cfg.name if sensor_cfg is not None else "height_scanner"]
    return sensor.data.pos_w[:, 2].unsqueeze(1) - sensor.data.ray_hits_w[..., 2] - offset


# ---- reward terms the reference DEFINES for this task but does not wire into its reward cfg (:159-164, 175-231, 256-266):
# plain torch on the state views, so a config override that registers one runs through the custom-term path ----------

def _elev_z(env, base: float = 0.19):
    return root_pos_w(env)[:, 2] - base


def forward_wheel_spin(env, asset_cfg=_ROBOT):                  # :159-164: summed throttle-joint speed, capped at 200
    robot = env.scene[asset_cfg.name]
    ids = robot.find_joints(".*_throttle")[0]
    return robot.data.joint_vel[:, ids].sum(-1).clamp(max=200.0)


def change_in_elevation(env):                                   # :175-178: upward world velocity only
    return root_lin_vel_w(env)[:, 2].clamp(min=0.0)


def steep_penalty(env, thresh_pitch):                           # :180-186: pitch (wrapped to [0, 2 pi)) above a threshold
    return (euler_xyz_from_quat(root_quat_w(env))[1] - thresh_pitch).clamp(min=0.0)


class _ElevationContinuity:
    """:188-203: +-50 x the change of elevation since the previous call while above `threshold_elev`.  The reference keeps
    the previous elevation in a function attribute (one global for the process); here it lives per env object."""
    __name__ = "elevation_continuity"

    def __call__(self, env, threshold_elev):
        z = _elev_z(env)
        prev = getattr(env, "_elev_continuity_prev", None)
        if prev is None or prev.shape != z.shape:
            prev = z.clone()
        dz = z - prev
        env._elev_continuity_prev = z.clone()
        return torch.where(z > threshold_elev, 50.0 * dz.abs(), torch.zeros_like(z))


elevation_continuity = _ElevationContinuity()


def yaw_change_onElev(env, threshold_yaw, threshold_z):         # :206-212
    wz = base_ang_vel(env)[:, 2].abs()
    return torch.where((_elev_z(env) > threshold_z) & (wz > threshold_yaw), 2.0 * wz * wz, torch.zeros_like(wz))


def upright_penalty(env, thresh_deg):                           # :217-222: tilt of the body z axis beyond thresh_deg [deg]
    q = root_quat_w(env)
    up = 1.0 - 2.0 * (q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2])   # R[2, 2]
    tilt = torch.rad2deg(torch.arccos(up))
    return torch.where(tilt > thresh_deg, tilt - thresh_deg, torch.zeros_like(tilt))


def roll_on_elev(env, z_start, roll_rate_thresh):               # :224-231
    wx = base_ang_vel(env)[:, 0].abs()
    return torch.where((_elev_z(env) > z_start) & (wx > roll_rate_thresh), 2.0 * wx, torch.zeros_like(wx))


def ascending(env):                                             # :256-259
    return root_lin_vel_w(env)[:, 2].clamp(min=0.0)


def low_vel_penalty(env, min_vel: float = 0.1):                 # :261-265
    return (base_lin_vel(env)[:, 0] < min_vel).float()


@_event("reset_uniform")
def reset_root_state_uniform(env, env_ids, pose_range, velocity_range, asset_cfg=_ROBOT):   # isaaclab mdp.events
    mask = torch.zeros(env.num_envs, dtype=torch.uint8, device=env.device)
    mask[env_ids] = 1
    env._batch.reset(mask)


# =====================================================================================================================
# Visual task terms (reference: wheeledlab_tasks/visual/mushr_visual_env_cfg.py, mdp_sensors/observations.py)
# =====================================================================================================================

@_kernel_term("reward", 0)
def traversable_reward(env):                                    # :309-312
    return env._eval_visual_terms()["terms"][0]


@_kernel_term("reward", 1)
def forward_vel(env):                                           # visual :370-371: body-frame forward speed
    return base_lin_vel(env)[:, 0]


def is_traversable(env):                                        # :304-307
    return (env._eval_visual_terms()["terms"][0] > 0).float()


@_kernel_term("termination", 0)
def out_of_map(env):                                            # :390-398
    return env._eval_visual_terms()["out_of_map"]


# ---- terms the visual cfg module DEFINES but does not register (visual/mushr_visual_env_cfg.py:314-368,400-403): torch terms
# on the state views, so that a config override can wire them in (the elevation twin: forward_wheel_spin ... low_vel_penalty above)

def _traversability_at(env, xy):
    """TraversabilityHashmapUtil.get_traversability / get_map_id (visual/utils/traversability_utils.py:68-88) for arbitrary
    points [M, 2]: x_idx = clamp(trunc((x + W / 2 + dx / 2) / dx), 0, rows - 1), likewise y; map[y_idx, x_idx].
    `env.traversability` -> (bool / uint8 map [rows, cols] on the env's device, (row_spacing, col_spacing))."""
    tmap, (rs, cs) = env.traversability
    rows, cols = tmap.shape
    width, height = rows * rs, cols * cs
    x_idx = ((xy[:, 0] + width / 2.0 + rs / 2.0) / rs).long().clamp(0, rows - 1)
    y_idx = ((xy[:, 1] + height / 2 + cs / 2) / cs).long().clamp(0, cols - 1)
    return tmap[y_idx, x_idx].bool()


def bool_is_not_traversable(env):                               # :314-322: off until 1000 episodes have passed
    if env.common_step_counter // env.max_episode_length < 1000:
        return torch.zeros(env.num_envs, dtype=torch.bool, device=env.device)
    return ~_traversability_at(env, root_pos_w(env)[..., :2])


def is_traversable_speed_scaled(env):                           # :324-325
    return _traversability_at(env, root_pos_w(env)[..., :2]).float() * base_lin_vel(env)[:, 0]


def _wheels_traversable(env):
    """[N, 4] bool: the four wheel links' xy on the map (:327-333)"""
    cfg = SceneEntityCfg("robot", body_names=".*wheel_link").resolve(env.scene)
    xy = env.scene[cfg.name].data.body_pos_w[:, cfg.body_ids][:, :, :2]
    B, nb = xy.shape[:2]
    return _traversability_at(env, xy.reshape(-1, 2)).reshape(B, nb)


def is_traversable_wheels(env):                                 # :327-334: +1 per wheel on the path, -5 per wheel off it
    t = _wheels_traversable(env)
    return torch.where(t, 1.0, -5.0).sum(dim=-1)


def binary_is_traversable_wheels(env):                          # :336-344: True when NO wheel is on the path
    return _wheels_traversable(env).float().sum(dim=-1) == 0


def vel_rew_trav(env, speed_target_on_trav: float = 1.0, speed_target_on_non_trav: float = 2.0):   # :346-354
    trav = _traversability_at(env, root_pos_w(env)[..., :2])
    target = torch.where(trav, speed_target_on_trav, speed_target_on_non_trav)
    d = -((torch.norm(base_lin_vel(env), dim=-1) - target) ** 2) + target ** 2
    return torch.where(d > 0.0, d, torch.zeros_like(d))


def low_speed_penalty(env, low_speed_thresh: float = 0.3):      # :361-364
    return (torch.norm(base_lin_vel(env), dim=-1) < low_speed_thresh).float()


def roll_over(env):                                             # :400-403: roll (wrapped to [0, 2 pi)) - pi inside (-pi/2, pi/2)
    roll = euler_xyz_from_quat(root_quat_w(env))[0] - math.pi
    return (roll < math.pi / 2) & (roll > -math.pi / 2)


_CAMERA = SceneEntityCfg("camera")


def camera_data_rgb_flattened_aug(env, sensor_cfg=None):        # mdp_sensors/observations.py:75-87
    return env._batch.observe()[:, : A.VIS_NPIX]


def camera_data_rgb_flattened(env, sensor_cfg=None):            # mdp_sensors/observations.py:64-73 (no augmentation)
    b = env._batch
    keep = (b.p.brightness, b.p.contrast, b.p.blur_sigma)
    b.p.brightness, b.p.contrast, b.p.blur_sigma = 1.0, 1.0, 0.0
    out = b.observe()[:, : A.VIS_NPIX].clone()
    b.p.brightness, b.p.contrast, b.p.blur_sigma = keep
    return out


def camera_data_rgb(env, sensor_cfg=_CAMERA):                   # mdp_sensors/observations.py:60-62
    """the camera's un-flattened image [N, 60, 80, 3] uint8: what `camera_data_rgb_flattened` crops, greys and normalises (the designed
    camera renders the black / white traversability plane, sky grey: the three channels are equal)"""
    return env.scene.sensors[sensor_cfg.name].data.output["rgb"]


def lidar_ranges(env, sensor_cfg):                              # mdp_sensors/observations.py:25-30
    """`linear_depth` of a lidar-like sensor of the scene, as is.  (No registered scene carries one -- the reference's F1Tenth task
    switches its RTX lidars off, drifting/disable_lidar.py -- the term works with any sensor whose data exposes that output.)"""
    return env.scene.sensors[sensor_cfg.name].data.output["linear_depth"]


def lidar_ranges_normalized(env, sensor_cfg):                   # mdp_sensors/observations.py:32-58
    """ranges + N(0, 0.1) noise, clipped to the sensor's [min_range, max_range], mapped to [0, 1]"""
    sensor = env.scene.sensors[sensor_cfg.name]
    r = sensor.data.output["linear_depth"]
    lo, hi = sensor.cfg.min_range, sensor.cfg.max_range
    noisy = r + torch.normal(mean=0.0, std=0.1, size=r.shape, device=r.device)
    return (torch.clip(noisy, min=lo, max=hi) - lo) / (hi - lo)


def camera_data_depth(env, sensor_cfg=_CAMERA):                 # mdp_sensors/observations.py:89-91
    """distance_to_image_plane of the robot's camera [N, 60, 80, 1] (defined but unwired in the reference's VisualObsCfg)"""
    return env.scene.sensors[sensor_cfg.name].data.output["distance_to_image_plane"]


def raycast_depth(env, sensor_cfg=_CAMERA, heightfield=None, max_depth: float | None = None):   # mdp_sensors/observations.py:93-95
    """the same data; `heightfield` / `max_depth` (not in the reference): render against another terrain / range"""
    if heightfield is None and max_depth is None:
        return camera_data_depth(env, sensor_cfg)
    far = env.scene.sensors[sensor_cfg.name].data.far if max_depth is None else max_depth
    if heightfield is None:
        return env.scene.sensors[sensor_cfg.name].data._camera().render(env._batch, far).unsqueeze(-1)
    from ..core import _cached_depth_camera          # any task's batch: the camera is built on (and cached with) the batch
    return _cached_depth_camera(env._batch, heightfield).render(env._batch, far).unsqueeze(-1)


@_event("reset_traversable")
def reset_root_state(env, env_ids, asset_cfg=_ROBOT):           # visual/mdp/events.py:11-42
    mask = torch.zeros(env.num_envs, dtype=torch.uint8, device=env.device)
    mask[env_ids] = 1
    env._batch.reset(mask)
