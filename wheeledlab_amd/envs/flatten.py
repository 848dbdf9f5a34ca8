"""Flatten a manager-based env config into the POD parameter struct of the fused kernel (SURVEY.md section 5 "config /
flags": "a flattening step that turns term cfgs into a POD params struct for the kernel").  Unknown term functions
are not silently dropped: custom reward terms are returned for post-kernel evaluation on the device, anything else
that the kernel cannot express raises."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

from .. import _abi as A
from ..params import MUSHR_CHASSIS_MASS, mushr_vehicle
from .configclass import fields_of
from .managers_cfg import AdditiveGaussianNoiseCfg

INT_MAX = 2 ** 31 - 1


@dataclass
class StartupSpec:
    """startup-mode events (domain randomisation applied once at construction)"""
    wheel_mu_s: tuple = (0.4, 0.4)
    wheel_mu_d: tuple = (0.4, 0.4)
    mu_buckets: int = 1
    mu_consistent: bool = True
    damping: tuple = (30.0, 30.0)
    mass_add: tuple = (0.0, 0.0)
    wheel_mass: tuple = (0.0, 0.0)      # each wheel link's mass ("abs"); (0, 0): wheels counted in chassis_mass
    chassis_mass: float = MUSHR_CHASSIS_MASS
    track_radius: float = 0.8
    track_straight: float = 0.8


@dataclass
class FlatDriftCfg:
    params: A.WlDriftParams
    startup: StartupSpec
    reward_names: list = field(default_factory=list)        # [(cfg name, slot)]
    custom_rewards: list = field(default_factory=list)      # [(name, term_cfg)] evaluated with torch after the kernel
    custom_terminations: list = field(default_factory=list) # [(name, term_cfg)] torch, on the post-step state views
    custom_obs: list = field(default_factory=list)          # [(name, term_cfg)] torch, concatenated after the fused block
    termination_names: dict = field(default_factory=dict)   # {"time_out": name, 0: name}
    curriculum: list = field(default_factory=list)          # [(name, term_cfg)]
    obs_dim: int = 14
    task: str = "drift"
    extra: dict = field(default_factory=dict)


def _terms(cfg):
    return [] if cfg is None else [(k, v) for k, v in fields_of(cfg) if hasattr(v, "func")]


def flatten_drift_cfg(cfg) -> FlatDriftCfg:
    p = A.WlDriftParams()
    su = StartupSpec()
    flat = FlatDriftCfg(p, su)
    p.sim_dt, p.decimation = float(cfg.sim.dt), int(cfg.decimation)
    step_dt = float(cfg.sim.dt) * int(cfg.decimation)   # python doubles, exactly as IsaacLab computes max_episode_length
    p.max_episode_length = math.ceil(cfg.episode_length_s / step_dt)

    # ---- action term ----
    act_terms = _named(cfg.actions)
    if len(act_terms) != 1:
        raise NotImplementedError("exactly one action term (throttle_steer) is supported")
    acfg = act_terms[0][1]
    if getattr(acfg.class_type, "wl_map", None) is None:
        raise NotImplementedError(f"action term {acfg.class_type.__name__} has no HIP implementation")
    a = p.action
    a.scale[0], a.scale[1] = acfg.scale
    a.offset[0], a.offset[1] = acfg.offset
    a.bounding = {"clip": 1, "tanh": 2, None: 0}[acfg.bounding_strategy]
    a.no_reverse, a.clip_wrapper, a.map = int(acfg.no_reverse), 0, acfg.class_type.wl_map
    a.base_length, a.base_width, a.wheel_radius = acfg.base_length, acfg.base_width, acfg.wheel_radius

    # ---- vehicle: actuator constants from the robot cfg, friction from the terrain material ----
    robot = cfg.scene.robot
    acts = robot.actuators
    thr = acts["throttle_joints"]
    rwd = "passive_joints" in acts or any(e.startswith("back_") for e in thr.joint_names_expr)
    mat = cfg.scene.terrain.physics_material
    if mat.friction_combine_mode != "multiply":
        raise NotImplementedError("only the 'multiply' friction combine mode is modelled")
    substeps = max(1, math.ceil(float(cfg.sim.dt) / 0.0101))
    p.vehicle = mushr_vehicle(drive=0 if rwd else 1, motor_limit=float(thr.effort_limit), substeps=substeps,
                              ground_mu=(mat.static_friction, mat.dynamic_friction))
    v = p.vehicle
    v.motor_sat, v.motor_vel_limit = float(thr.saturation_effort), float(thr.velocity_limit)
    st = acts["steering_joints"]
    v.steer_kp, v.steer_kd, v.steer_effort, v.steer_vel_limit = st.stiffness, st.damping, st.effort_limit, st.velocity_limit
    v.half_wheelbase_f = v.half_wheelbase_r = acfg.base_length / 2
    v.half_track = acfg.base_width / 2
    v.wheel_radius = acfg.wheel_radius
    su.damping = (float(thr.damping),) * 2

    # ---- rewards ----
    for i in range(A.WL_MAX_REW_TERMS):
        p.weight[i] = 0.0
    # kernel defaults so that disabled terms stay finite
    p.straight, p.r_in, p.r_out, p.r_line = 0.8, 0.0, 1e30, 0.8   # r_in is squared in-kernel: 0 disables it
    p.slip_min, p.slip_max, p.slip_min_vx = 0.25, 0.55, 0.5
    p.speed_target, p.speed_offset, p.tlgr_thresh, p.ctd_offset, p.ctd_p = 3.0, -9.0, math.pi / 4, -1.0, 1.0
    straight_seen = {}
    for name, term in _terms(cfg.rewards):
        f = term.func
        if getattr(f, "wl_kind", None) != "reward":
            flat.custom_rewards.append((name, term))
            continue
        p.weight[f.wl_slot] = float(term.weight)
        flat.reward_names.append((name, f.wl_slot))
        _apply_params(p, f, term.params, straight_seen, name)

    # ---- terminations ----
    has_timeout = False
    for name, term in _terms(cfg.terminations):
        f = term.func
        if getattr(f, "wl_kind", None) != "termination":   # any f(env, **params) -> bool[N]: torch fallback on the state views
            flat.custom_terminations.append((name, term))
            continue
        if f.wl_slot == "time_out":
            if not term.time_out:
                raise NotImplementedError("mdp.time_out must be registered with time_out=True")
            has_timeout = True
        else:
            _apply_params(p, f, term.params, straight_seen, name)
        flat.termination_names[f.wl_slot] = name
    if not has_timeout:
        p.max_episode_length = INT_MAX

    # ---- observations: the built-in 14-dim layout ----
    pol = cfg.observations.policy
    obs_terms = _terms(pol)
    want = ["root_pos_w", "root_euler_xyz", "base_lin_vel", "base_ang_vel", "last_action"]
    got = [getattr(t.func, "__name__", str(t.func)) for _, t in obs_terms]
    if got[:len(want)] != want:
        raise NotImplementedError(f"the policy observation must start with the fused kernel's layout {want} (got {got}); "
                                  "further terms are evaluated with torch and concatenated behind it")
    flat.custom_obs = obs_terms[len(want):]
    p.enable_corruption = int(bool(pol.enable_corruption))
    for i, (_, t) in enumerate(obs_terms[:4]):
        n = t.noise
        if n is None:
            p.noise_std[i] = 0.0
        elif isinstance(n, AdditiveGaussianNoiseCfg) and n.mean == 0.0:
            p.noise_std[i] = float(n.std)
        else:
            raise NotImplementedError("only zero-mean additive Gaussian observation noise is fused")
        if t.clip is not None or t.scale is not None:
            raise NotImplementedError("clip / scale on the proprioceptive terms is not fused")
    if tuple(obs_terms[4][1].clip or ()) != (-1.0, 1.0):
        raise NotImplementedError("last_action must be clipped to (-1, 1) (common/observations.py:52)")

    # ---- events ----
    p.num_ref_points, p.pos_noise, p.yaw_noise = 20, 0.0, 0.0
    p.enable_pushes = 0
    p.hf_interval[0], p.hf_interval[1], p.lf_interval[0], p.lf_interval[1] = 1e9, 1e9, 1e9, 1e9
    p.hf_vel_x = p.hf_vel_y = p.hf_vel_yaw = p.lf_vel_yaw = 0.0
    pushes = []
    for name, term in _terms(cfg.events):
        ev = getattr(term.func, "wl_event", None)
        pr = term.params
        if ev == "reset_along_track" and term.mode == "reset":
            p.num_ref_points = int(pr.get("num_points", 20))
            p.pos_noise, p.yaw_noise = float(pr.get("pos_noise", 0.0)), float(pr.get("yaw_noise", 0.0))
            su.track_radius = float(pr.get("track_radius", 0.8))
            su.track_straight = float(pr.get("track_straight_dist", 0.8))
        elif ev == "wheel_friction" and term.mode == "startup":
            su.wheel_mu_s, su.wheel_mu_d = tuple(pr["static_friction_range"]), tuple(pr["dynamic_friction_range"])
            su.mu_buckets, su.mu_consistent = int(pr["num_buckets"]), bool(pr.get("make_consistent", False))
        elif ev == "actuator_gains" and term.mode == "startup":
            if pr.get("operation", "abs") != "abs":
                raise NotImplementedError("randomize_actuator_gains: only operation='abs'")
            su.damping = tuple(pr["damping_distribution_params"])
        elif ev == "base_mass" and term.mode == "startup":
            if pr.get("operation", "add") != "add":
                raise NotImplementedError("randomize_rigid_body_mass: only operation='add'")
            su.mass_add = tuple(pr["mass_distribution_params"])
        elif ev == "push" and term.mode == "interval":
            pushes.append((name, term))
        else:
            raise NotImplementedError(f"event term '{name}' ({getattr(term.func, '__name__', term.func)}, mode {term.mode}) "
                                      "has no HIP implementation")
    if len(pushes) > 2:
        raise NotImplementedError("at most two interval push events are fused")
    for slot, (name, term) in enumerate(pushes):
        vr = {k: tuple(v) for k, v in term.params["velocity_range"].items()}
        for k, (lo, hi) in vr.items():
            if abs(lo + hi) > 1e-9:
                raise NotImplementedError(f"push '{name}': only symmetric ranges are fused")
        lo_s, hi_s = term.interval_range_s
        if slot == 0:
            if set(vr) - {"x", "y", "yaw"}:
                raise NotImplementedError(f"push '{name}': only x / y / yaw components are fused")
            p.hf_interval[0], p.hf_interval[1] = lo_s, hi_s
            p.hf_vel_x, p.hf_vel_y, p.hf_vel_yaw = (vr.get(k, (0, 0))[1] for k in ("x", "y", "yaw"))
        else:
            if set(vr) - {"yaw"}:
                raise NotImplementedError(f"push '{name}': the second interval event is yaw-only in the fused kernel")
            p.lf_interval[0], p.lf_interval[1] = lo_s, hi_s
            p.lf_vel_yaw = vr.get("yaw", (0, 0))[1]
        p.enable_pushes = 1
    if p.num_ref_points > 32:
        raise NotImplementedError("at most 32 reference poses")

    flat.curriculum = _terms(cfg.curriculum)
    p.log_episode_sums = 1
    return flat


def _named(cfg):
    return [(k, v) for k, v in fields_of(cfg) if hasattr(v, "class_type")]


def _apply_params(p, func, params, straight_seen, name):
    for key, fld in func.wl_params.items():
        if key not in params:
            continue
        val = float(params[key])
        if fld in ("straight",):  # shared by several terms: they must agree (they do in the reference, STRAIGHT = 0.8)
            if fld in straight_seen and abs(straight_seen[fld] - val) > 1e-9:
                raise NotImplementedError(f"term '{name}': '{key}'={val} conflicts with {straight_seen[fld]} of another term")
            straight_seen[fld] = val
        setattr(p, fld, val)


# =====================================================================================================================
# elevation / visual tasks
# =====================================================================================================================

@dataclass
class FlatTaskCfg:
    task: str
    params: object
    startup: StartupSpec
    reward_names: list = field(default_factory=list)
    custom_rewards: list = field(default_factory=list)
    custom_terminations: list = field(default_factory=list)
    custom_obs: list = field(default_factory=list)
    termination_names: dict = field(default_factory=dict)
    curriculum: list = field(default_factory=list)
    obs_dim: int = 0
    extra: dict = field(default_factory=dict)


def _common_vehicle(cfg, p, acfg):
    robot = cfg.scene.robot
    acts = robot.actuators
    thr = acts["throttle_joints"]
    rwd = "passive_joints" in acts or any(e.startswith("back_") for e in thr.joint_names_expr)
    mat = cfg.scene.terrain.physics_material
    if mat.friction_combine_mode != "multiply":
        raise NotImplementedError("only the 'multiply' friction combine mode is modelled")
    # the linearly implicit integrator steps once per sim.dt (the reference's physics rate) up to h = 20 ms, its validated range
    substeps = max(1, math.ceil(float(cfg.sim.dt) / 0.0201))
    p.vehicle = mushr_vehicle(drive=0 if rwd else 1, motor_limit=float(thr.effort_limit), substeps=substeps,
                              ground_mu=(mat.static_friction, mat.dynamic_friction), implicit=1)
    v = p.vehicle
    v.motor_sat, v.motor_vel_limit = float(thr.saturation_effort), float(thr.velocity_limit)
    st = acts["steering_joints"]
    v.steer_kp, v.steer_kd, v.steer_effort, v.steer_vel_limit = st.stiffness, st.damping, st.effort_limit, st.velocity_limit
    v.half_wheelbase_f = v.half_wheelbase_r = acfg.base_length / 2
    v.half_track = acfg.base_width / 2
    v.wheel_radius = acfg.wheel_radius
    return float(thr.damping)


def _action(cfg, p):
    act_terms = _named(cfg.actions)
    if len(act_terms) != 1 or getattr(act_terms[0][1].class_type, "wl_map", None) is None:
        raise NotImplementedError("exactly one RCCarRWD / RCCar4WD action term is supported")
    acfg = act_terms[0][1]
    a = p.action
    a.scale[0], a.scale[1] = acfg.scale
    a.offset[0], a.offset[1] = acfg.offset
    a.bounding = {"clip": 1, "tanh": 2, None: 0}[acfg.bounding_strategy]
    a.no_reverse, a.clip_wrapper, a.map = int(acfg.no_reverse), 0, acfg.class_type.wl_map
    a.base_length, a.base_width, a.wheel_radius = acfg.base_length, acfg.base_width, acfg.wheel_radius
    return acfg


def _startup_events(cfg, su, allowed_reset):
    reset_term = None
    for name, term in _terms(cfg.events):
        ev = getattr(term.func, "wl_event", None)
        pr = term.params
        if ev in allowed_reset and term.mode == "reset":
            reset_term = term
        elif ev == "wheel_friction" and term.mode == "startup":
            su.wheel_mu_s, su.wheel_mu_d = tuple(pr["static_friction_range"]), tuple(pr["dynamic_friction_range"])
            su.mu_buckets, su.mu_consistent = int(pr["num_buckets"]), bool(pr.get("make_consistent", False))
        elif ev == "base_mass" and term.mode == "startup" and "wheel" in str(getattr(pr.get("asset_cfg"), "body_names", "")):
            # randomize_rigid_body_mass on the wheel links (visual/mushr_visual_env_cfg.py:289-298): the four link masses add to the
            # vehicle's mass row; the wheels' ROTATIONAL inertia stays the model's one `wheel_inertia` (not randomised per wheel)
            if pr.get("operation", "add") != "abs":
                raise NotImplementedError("randomize_rigid_body_mass on the wheel links: operation abs")
            su.wheel_mass = tuple(pr["mass_distribution_params"])
        elif ev == "base_mass" and term.mode == "startup":
            if pr.get("operation", "add") == "add":
                su.mass_add = tuple(pr["mass_distribution_params"])
            elif pr["operation"] == "abs":
                su.chassis_mass, su.mass_add = 0.0, tuple(pr["mass_distribution_params"])
            else:
                raise NotImplementedError("randomize_rigid_body_mass: operation add / abs")
        elif ev == "actuator_gains" and term.mode == "startup":
            su.damping = tuple(pr["damping_distribution_params"])
        else:
            raise NotImplementedError(f"event term '{name}' ({getattr(term.func, '__name__', term.func)}, mode {term.mode}) "
                                      "has no HIP implementation for this task")
    return reset_term


def _sym(rng, what):
    lo, hi = rng
    if abs(lo + hi) > 1e-9:
        raise NotImplementedError(f"{what}: only symmetric ranges are fused")
    return float(hi)


def flatten_elev_cfg(cfg) -> FlatTaskCfg:
    """MushrElevationRLEnvCfg-shaped config -> WlElevParams (reference: elevation/mushr_elevation_env_cfg.py)"""
    from ..params import elev_params
    from . import mdp
    p = elev_params()                                   # term constants that have no config field keep the reference values
    su = StartupSpec(wheel_mu_s=(2.0, 2.0), wheel_mu_d=(1.0, 1.0), mass_add=(0.0, 0.0))
    flat = FlatTaskCfg("elevation", p, su, obs_dim=A.ELEV_OBS_DIM)
    p.sim_dt, p.decimation = float(cfg.sim.dt), int(cfg.decimation)
    p.max_episode_length = math.ceil(cfg.episode_length_s / (float(cfg.sim.dt) * int(cfg.decimation)))
    acfg = _action(cfg, p)
    su.damping = (_common_vehicle(cfg, p, acfg),) * 2
    for i in range(A.WL_MAX_REW_TERMS):
        p.weight[i] = 0.0
    for name, term in _terms(cfg.rewards):
        f = term.func
        if f is mdp.is_terminated_term:
            if term.params.get("term_keys") not in ("stuck", ["stuck"]):
                raise NotImplementedError("is_terminated_term is fused for term_keys='stuck' (:301-305)")
            slot = 3
        elif getattr(f, "wl_kind", None) == "reward" and f in (mdp.goal_progress_rate, mdp.higher_elevation, mdp.is_falling_penalty):
            slot = f.wl_slot
            for key, fld in f.wl_params.items():
                if key in term.params:
                    setattr(p, fld, float(term.params[key]))
        else:
            flat.custom_rewards.append((name, term))
            continue
        p.weight[slot] = float(term.weight)
        flat.reward_names.append((name, slot))
    has_timeout = False
    p.min_height, p.goal_dist, p.upright_cos = -1e30, -1.0, -2.0      # disabled unless the cfg registers the term
    p.stuck_min_vel, p.stuck_wheel_spin = -1e30, 1e30
    for name, term in _terms(cfg.terminations):
        f = term.func
        if f is mdp.time_out:
            has_timeout = True
            flat.termination_names["time_out"] = name
            continue
        if getattr(f, "wl_kind", None) != "termination" or f not in (mdp.root_height_below_minimum, mdp.stuck, mdp.upright_bool, mdp.close_to_goal):
            flat.custom_terminations.append((name, term))     # torch fallback on the state views
            continue
        for key, fld in f.wl_params.items():
            val = float(term.params[key])
            setattr(p, fld, math.cos(math.radians(val)) if fld == "upright_cos" else val)
        flat.termination_names[f.wl_slot] = name
    if not has_timeout:
        p.max_episode_length = INT_MAX
    # observation layout
    want = ["goal_relative_xyz", "root_euler_xyz", "base_lin_vel", "base_ang_vel", "last_action", "world_height_map"]
    obs_terms = _terms(cfg.observations.policy)
    got = [getattr(t.func, "__name__", str(t.func)) for _, t in obs_terms]
    if got[:len(want)] != want or cfg.observations.policy.enable_corruption:
        raise NotImplementedError(f"the policy observation must start with the fused kernel's layout {want} (no corruption); got {got}")
    flat.custom_obs = obs_terms[len(want):]
    hm = obs_terms[5][1]
    p.scan_offset, p.elev_z0 = float(hm.params.get("offset", 0.084)), float(hm.params.get("plane_init_value", 0.19))
    p.obs_clip = float(hm.clip[1])
    if tuple(obs_terms[2][1].clip) != (-p.obs_clip, p.obs_clip) or tuple(obs_terms[3][1].clip) != (-p.obs_clip, p.obs_clip):
        raise NotImplementedError("velocity clips must equal the height-map clip (10)")
    pat = cfg.scene.height_scanner.pattern_cfg
    p.scan_size, p.scan_res = float(pat.size[0]), float(pat.resolution)
    if round(p.scan_size / p.scan_res) + 1 != A.ELEV_SCAN_N or pat.size[0] != pat.size[1]:
        raise NotImplementedError(f"the fused height scan is {A.ELEV_SCAN_N} x {A.ELEV_SCAN_N} rays")
    # events + command
    rt = _startup_events(cfg, su, ("reset_uniform",))
    if rt is not None:
        pr, vr = rt.params["pose_range"], rt.params["velocity_range"]
        p.reset_xy, p.reset_yaw = _sym(pr["x"], "reset x"), _sym(pr["yaw"], "reset yaw")
        if tuple(pr["x"]) != tuple(pr["y"]) or tuple(vr["x"]) != tuple(vr["y"]):
            raise NotImplementedError("reset ranges must be equal in x and y")
        p.reset_vel[0], p.reset_vel[1] = vr["x"]
    p.reset_z = float(cfg.scene.terrain.height)
    cmd = cfg.commands.goal_pose
    p.cmd_xy, p.cmd_heading = _sym(cmd.ranges.pos_x, "command x"), _sym(cmd.ranges.heading, "command heading")
    p.cmd_resample_s = float(cmd.resampling_time_range[1])
    flat.curriculum = _terms(cfg.curriculum)
    flat.extra["heightfield"] = getattr(cfg.scene.terrain, "heightfield", None)
    return flat


def flatten_visual_cfg(cfg) -> FlatTaskCfg:
    """MushrVisualRLEnvCfg-shaped config -> WlVisualParams (reference: visual/mushr_visual_env_cfg.py)"""
    from ..params import visual_params
    from . import mdp
    p = visual_params()
    su = StartupSpec(wheel_mu_s=(0.5, 0.5), wheel_mu_d=(0.5, 0.5), mass_add=(0.0, 0.0))   # PhysX default material
    flat = FlatTaskCfg("visual", p, su, obs_dim=A.VIS_OBS_DIM)
    p.sim_dt, p.decimation = float(cfg.sim.dt), int(cfg.decimation)
    p.max_episode_length = math.ceil(cfg.episode_length_s / (float(cfg.sim.dt) * int(cfg.decimation)))
    acfg = _action(cfg, p)
    su.damping = (_common_vehicle(cfg, p, acfg),) * 2
    for i in range(A.WL_MAX_REW_TERMS):
        p.weight[i] = 0.0
    for name, term in _terms(cfg.rewards):
        if term.func is mdp.traversable_reward:
            slot = 0
        elif term.func is mdp.forward_vel:
            slot = 1
        else:
            flat.custom_rewards.append((name, term))
            continue
        p.weight[slot] = float(term.weight)
        flat.reward_names.append((name, slot))
    has_timeout, has_oom = False, False
    for name, term in _terms(cfg.terminations):
        if term.func is mdp.time_out:
            has_timeout = True
            flat.termination_names["time_out"] = name
        elif term.func is mdp.out_of_map:
            has_oom = True
            flat.termination_names[0] = name
        else:
            flat.custom_terminations.append((name, term))     # torch fallback on the state views
    if not has_timeout:
        p.max_episode_length = INT_MAX
    flat.extra["out_of_map"] = has_oom
    depth_task = getattr(cfg, "wl_task", "visual") == "visual_depth"
    want = ["raycast_depth" if depth_task else "camera_data_rgb_flattened_aug", "base_lin_vel", "base_ang_vel", "last_action"]
    vis_obs = _terms(cfg.observations.policy)
    got = [getattr(t.func, "__name__", str(t.func)) for _, t in vis_obs]
    if got[:len(want)] != want or cfg.observations.policy.enable_corruption:
        raise NotImplementedError(f"the policy observation must start with the fused kernel's layout {want} (no corruption); got {got}")
    flat.custom_obs = vis_obs[len(want):]
    cam = cfg.scene.camera
    if (cam.height, cam.width) != (60, 80):
        raise NotImplementedError("the fused camera is 60 x 80")
    p.fx = cam.width * cam.spawn.focal_length / cam.spawn.horizontal_aperture
    p.fy = cam.height * cam.spawn.focal_length / cam.spawn.vertical_aperture
    p.cx, p.cy = cam.width / 2, cam.height / 2
    p.cam_pos[0], p.cam_pos[1], p.cam_pos[2] = cam.body_pos
    _startup_events(cfg, su, ("reset_traversable",))
    flat.curriculum = _terms(cfg.curriculum)
    t = cfg.scene.terrain
    flat.extra.update(map=t.traversability_hashmap, map_size=(t.num_rows, t.num_cols), env_size=(t.env_num_rows, t.env_num_cols),
                      group=(t.group_num_rows, t.group_num_cols), walkers=t.num_walkers, spacing=(t.row_spacing, t.col_spacing),
                      augment=bool(getattr(cfg, "augment_camera", True)))
    if depth_task:      # extension task: heightfield terrain + the depth image as observation (tasks/visual_depth)
        flat.task, flat.obs_dim = "visual_depth", A.VISDEPTH_OBS_DIM
        clip = getattr(cam.spawn, "clipping_range", None) or (0.01, 20.0)
        flat.extra.update(heightfield=getattr(t, "heightfield", None), max_depth=float(clip[1]), augment=False)
    return flat


def flatten_cfg(cfg):
    task = getattr(cfg, "wl_task", "drift")
    if task == "elevation":
        return flatten_elev_cfg(cfg)
    if task in ("visual", "visual_depth"):
        return flatten_visual_cfg(cfg)
    f = flatten_drift_cfg(cfg)
    f.task, f.extra = "drift", {}
    return f
