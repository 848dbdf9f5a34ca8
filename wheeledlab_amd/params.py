"""Task constants as C-ABI parameter structs.  Values restate the reference configs (cited per field; paths relative
to /root/reference/source/) or, for the vehicle model, the designed constants of DESIGN.md section 4."""
from __future__ import annotations

import math

from ._abi import WlActionParams, WlDriftParams, WlElevParams, WlVehicleParams

MUSHR_NOMINAL_MASS = 3.4  # 3.0 kg chassis + mean U(0.3, 0.5) added base mass (mushr_drift_env_cfg.py:145-154)
MUSHR_CHASSIS_MASS = 3.0


def mushr_vehicle(drive: int = 0, motor_limit: float = 0.5, substeps: int = 1,
                  ground_mu: tuple[float, float] = (1.1, 1.0), implicit: int = 0) -> WlVehicleParams:
    """MuSHR-class 1/10 car.  Geometry: wheeledlab_tasks/common/actions.py:17-20 (L 0.325, W 0.2, r 0.05).
    Actuators: wheeledlab_assets/wheeledlab_assets/hound.py:4-52.  Mass / inertia / compliance: designed (USD missing)."""
    v = WlVehicleParams()
    k, g, r = 3000.0, 9.81, 0.05
    v.gravity = g
    v.half_wheelbase_f = v.half_wheelbase_r = 0.325 / 2
    v.half_track = 0.2 / 2
    v.wheel_radius = r
    v.wheel_z = r - MUSHR_NOMINAL_MASS * g / (4 * k)  # root-link origin rests on the ground at nominal load
    v.cg_z = 0.06
    v.gyr_x, v.gyr_y, v.gyr_z = 0.06, 0.12, 0.13
    v.wheel_inertia, v.wheel_damping = 8e-5, 1e-4
    v.susp_k, v.susp_c = k, 60.0
    v.susp_fmax = 24.0 * MUSHR_NOMINAL_MASS * g / 4.0   # 200 N: 24 x the static wheel load = 6.7 cm of penetration, more than the wheel's radius (designed; see include/wheeledlab_amd.h)
    v.ground_mu_s, v.ground_mu_d = ground_mu        # mushr_drift_env_cfg.py:45-50 ("multiply" combine)
    v.slip_peak, v.v_min = 0.12, 0.25
    v.motor_sat, v.motor_limit, v.motor_vel_limit = 1.05, motor_limit, 450.0   # hound.py:13-21, 40-43
    v.drive = drive                                  # hound.py:44-51: front throttle joints passive in 2WD
    v.steer_kp, v.steer_kd, v.steer_effort, v.steer_vel_limit, v.steer_inertia = 100.0, 10.0, 3.2, 10.0, 2e-4
    v.substeps = substeps
    v.implicit = implicit                            # integrator: csrc/wl_vehicle.h (0 explicit h <= 5 ms, 1 linearly implicit)
    return v


def mushr_action(map_: int = 0, base_length: float = 0.325, base_width: float = 0.2,
                 scale: tuple[float, float] = (3.0, 0.488)) -> WlActionParams:
    """MushrRWDActionCfg / Mushr4WDActionCfg (wheeledlab_tasks/common/actions.py:5-57)"""
    a = WlActionParams()
    a.scale[0], a.scale[1] = scale
    a.offset[0] = a.offset[1] = 0.0
    a.bounding, a.no_reverse, a.clip_wrapper, a.map = 1, 1, 1, map_
    a.base_length, a.base_width, a.wheel_radius = base_length, base_width, 0.05
    return a


def drift_params() -> WlDriftParams:
    """MushrDriftRLEnvCfg (wheeledlab_tasks/drifting/mushr_drift_env_cfg.py:369-404) flattened to the kernel struct"""
    p = WlDriftParams()
    p.sim_dt, p.decimation = 0.005, 4                                  # :393-394
    p.max_episode_length = math.ceil(5.0 / (0.005 * 4))                # :396
    p.action = mushr_action(0)                                         # :377, :397
    p.vehicle = mushr_vehicle(drive=0, motor_limit=0.5)                # :59 MUSHR_SUS_2WD_CFG
    p.straight, p.r_in, p.r_out, p.r_line = 0.8, 0.3, 2.0, 0.8         # :27-30
    for i, w in enumerate((10.0, -5.0, 40.0, 0.0, 20.0, -50.0, -5000.0, 0.0)):   # :246-299
        p.weight[i] = w
    p.slip_min, p.slip_max, p.slip_min_vx = 0.25, 0.55, 1.0            # :249-253
    p.speed_target, p.speed_offset = 3.0, -9.0                         # :167, :256-263
    p.tlgr_thresh = 1.0                                                # :272
    p.ctd_offset, p.ctd_p = -1.0, 1.0                                  # :284-293
    p.enable_corruption = 1                                            # :399
    for i, s in enumerate((0.1, 0.1, 0.5, 0.4)):                       # common/observations.py:27-45
        p.noise_std[i] = s
    p.num_ref_points, p.pos_noise, p.yaw_noise = 20, 0.5, 1.0          # :82-93
    p.enable_pushes = 1                                                # :121-143
    p.hf_interval[0], p.hf_interval[1] = 0.1, 0.4
    p.hf_vel_x, p.hf_vel_y, p.hf_vel_yaw = 0.1, 0.03, 0.3
    p.lf_interval[0], p.lf_interval[1] = 0.8, 1.2
    p.lf_vel_yaw = 0.6
    p.log_episode_sums = 1
    return p


def elev_params() -> WlElevParams:
    """MushrElevationRLEnvCfg (wheeledlab_tasks/elevation/mushr_elevation_env_cfg.py:438-469) flattened"""
    p = WlElevParams()
    p.sim_dt, p.decimation = 0.01, 10                                   # :461-462
    p.max_episode_length = math.ceil(20.0 / (0.01 * 10))                # :465
    p.action = mushr_action(1)                                          # :446 Mushr4WDActionCfg
    p.vehicle = mushr_vehicle(drive=1, motor_limit=0.25, substeps=1, ground_mu=(1.0, 1.0), implicit=1)   # :130, :102-107; h = sim.dt
    for i, w in enumerate((200.0, 5000.0, 0.0, -200.0, 0.0, 0.0, 0.0, 0.0)):                  # :286-305
        p.weight[i] = w
    p.min_height = 0.15                                                 # :356-359
    p.stuck_min_vel, p.stuck_wheel_spin, p.stuck_vel_cap = 0.02, 5.0, 1.2     # :360-366, :157
    p.upright_cos, p.goal_dist = math.cos(math.radians(60.0)), 0.5     # :368-376
    p.fall_vel = 0.10                                                   # :251-254
    p.elev_z0, p.elev_min, p.elev_min_vel, p.progress_offset = 0.19, 0.1, 0.1, 5.0   # :166-173, :249
    p.reset_xy, p.reset_yaw = 19.0, 3.14                                # :409-419
    p.reset_vel[0], p.reset_vel[1] = 0.1, 0.2
    p.reset_z, p.spawn_clearance = 0.25, 0.06                           # :147-149 (terrain.height); clearance: designed
    p.cmd_xy, p.cmd_heading, p.cmd_resample_s = 19.0, 3.14, 10.0        # :425-435
    p.scan_size, p.scan_res, p.scan_offset, p.obs_clip = 2.5, 0.1, 0.084, 10.0   # :139, :74-82
    p.log_episode_sums = 1
    return p


def struct_to_dict(s) -> dict:
    """ctypes struct -> nested plain dict (for logging / equality tests)"""
    out = {}
    for name, _ in s._fields_:
        v = getattr(s, name)
        if hasattr(v, "_fields_"):
            out[name] = struct_to_dict(v)
        elif hasattr(v, "__len__"):
            out[name] = [float(x) for x in v]
        else:
            out[name] = v
    return out


def visual_params():
    """MushrVisualRLEnvCfg (wheeledlab_tasks/visual/mushr_visual_env_cfg.py:412-444) flattened"""
    from ._abi import WlVisualParams
    p = WlVisualParams()
    p.sim_dt, p.decimation = 0.02, 10                                   # :435-436
    p.max_episode_length = math.ceil(10.0 / (0.02 * 10))                # :439
    p.action = mushr_action(1)
    p.vehicle = mushr_vehicle(drive=1, motor_limit=0.25, substeps=1, ground_mu=(2.0, 2.0), implicit=1)   # :130-135; h = sim.dt
    for i, w in enumerate((5.0, 7.0, 0, 0, 0, 0, 0, 0)):                # :375-385
        p.weight[i] = w
    p.reset_z = 0.1                                                     # :203
    p.cam_pos[0], p.cam_pos[1], p.cam_pos[2] = 0.23, 0.0, 0.18          # designed (camera_link pose is in the missing USD)
    p.fx = 80 * 1.9299999475479126 / 3.8959999084472656                 # :236-238
    p.fy = 60 * 1.9299999475479126 / 2.453000068664551
    p.cx, p.cy = 40.0, 30.0
    p.sky, p.brightness, p.contrast, p.blur_sigma, p.contrast_first = 0.5, 1.0, 1.0, 0.0, 0
    p.log_episode_sums = 1
    return p
