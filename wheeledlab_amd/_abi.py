"""ctypes mirror of include/wheeledlab_amd.h (the C-ABI drop-in boundary) and the loader of the HIP library.

The product path has NO fallback: if ``libwheeledlab_amd.so`` is missing or a symbol cannot be resolved this module
raises at import / call time (``HipExtensionMissing``); nothing under ``wheeledlab_amd`` imports ``oracle``.
"""
from __future__ import annotations

import ctypes as C
import os

WL_ABI_VERSION = 23
WL_MAX_REW_TERMS = 8

# WlStateField
(S_PX, S_PY, S_PZ, S_QW, S_QX, S_QY, S_QZ, S_VX, S_VY, S_VZ, S_WX, S_WY, S_WZ, S_WHEEL_BL, S_WHEEL_BR, S_WHEEL_FL,
 S_WHEEL_FR, S_STEER_POS, S_STEER_VEL, S_ACT0, S_ACT1, S_TIMER_HF, S_TIMER_LF, S_MU_S, S_MU_D, S_DAMP, S_MASS,
 S_EPSUM0) = range(28)
S_CMD_BX, S_CMD_BY, S_TGT_X, S_TGT_Y, S_TGT_H, S_CMD_TIMER = range(S_EPSUM0 + 8, S_EPSUM0 + 14)
S_COUNT = S_EPSUM0 + 14
N_DYN = 23
ELEV_SCAN_N = 26
ELEV_OBS_DIM = 13 + ELEV_SCAN_N * ELEV_SCAN_N
ELEV_TERM_NAMES = ("vel_towards_goal", "height_z", "falling_penalty", "termination_penalty")
ELEV_DONE_NAMES = ("cart_out_of_bounds", "stuck", "rollover", "at_goal")
# WlMetric
M_EPSUM0, M_RESETS, M_TIMEOUTS, M_TERM0, M_NONFINITE, M_EPLEN, M_COUNT = 0, 8, 9, 10, 14, 15, 16
# WlEnvBuffers.flags (WL_FLAG_*): force an instantiation the launchers otherwise pick from the batch size
FLAG_STREAM, FLAG_NO_STREAM, FLAG_SCAN_LDS, FLAG_SCAN_GATHER = 1, 2, 4, 8
M_SHARDS = 32   # WL_M_SHARDS: an accumulator vector is [M_SHARDS][M_COUNT], its value the sum over shards
# WlDriftRewTerm
DRIFT_TERM_NAMES = ("side_slip", "vel", "progress", "tlgr", "turn_energy", "cross_track", "term_pens")

ERRORS = {0: "ok", -1: "invalid argument", -2: "kernel launch failed", -3: "buffer alignment / stride violation",
          -4: "no HIP device"}


class HipExtensionMissing(RuntimeError):
    pass


class WlError(RuntimeError):
    pass


class WlVehicleParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in (
        "gravity", "half_wheelbase_f", "half_wheelbase_r", "half_track", "wheel_radius", "wheel_z", "cg_z", "gyr_x",
        "gyr_y", "gyr_z", "wheel_inertia", "wheel_damping", "susp_k", "susp_c", "ground_mu_s", "ground_mu_d",
        "slip_peak", "v_min", "motor_sat", "motor_limit", "motor_vel_limit")] + [("drive", C.c_int32)] + [
        (n, C.c_float) for n in ("steer_kp", "steer_kd", "steer_effort", "steer_vel_limit", "steer_inertia")] + [
        ("substeps", C.c_int32), ("implicit", C.c_int32), ("susp_fmax", C.c_float)]


class WlActionParams(C.Structure):
    _fields_ = [("scale", C.c_float * 2), ("offset", C.c_float * 2), ("bounding", C.c_int32),
                ("no_reverse", C.c_int32), ("clip_wrapper", C.c_int32), ("map", C.c_int32),
                ("base_length", C.c_float), ("base_width", C.c_float), ("wheel_radius", C.c_float)]


class WlDriftParams(C.Structure):
    _fields_ = [
        ("sim_dt", C.c_float), ("decimation", C.c_int32), ("max_episode_length", C.c_int32),
        ("action", WlActionParams), ("vehicle", WlVehicleParams),
        ("straight", C.c_float), ("r_in", C.c_float), ("r_out", C.c_float), ("r_line", C.c_float),
        ("weight", C.c_float * WL_MAX_REW_TERMS),
        ("slip_min", C.c_float), ("slip_max", C.c_float), ("slip_min_vx", C.c_float),
        ("speed_target", C.c_float), ("speed_offset", C.c_float), ("tlgr_thresh", C.c_float),
        ("ctd_offset", C.c_float), ("ctd_p", C.c_float),
        ("enable_corruption", C.c_int32), ("noise_std", C.c_float * 4),
        ("num_ref_points", C.c_int32), ("pos_noise", C.c_float), ("yaw_noise", C.c_float),
        ("enable_pushes", C.c_int32), ("hf_interval", C.c_float * 2), ("hf_vel_x", C.c_float),
        ("hf_vel_y", C.c_float), ("hf_vel_yaw", C.c_float), ("lf_interval", C.c_float * 2),
        ("lf_vel_yaw", C.c_float), ("log_episode_sums", C.c_int32),
    ]


class WlHeightField(C.Structure):
    _fields_ = [("height", C.c_void_p), ("nx", C.c_int32), ("ny", C.c_int32), ("x0", C.c_float), ("y0", C.c_float),
                ("cell", C.c_float), ("outside_z", C.c_float), ("z_scale", C.c_float), ("pair", C.c_void_p)]


class WlElevParams(C.Structure):
    _fields_ = [
        ("sim_dt", C.c_float), ("decimation", C.c_int32), ("max_episode_length", C.c_int32),
        ("action", WlActionParams), ("vehicle", WlVehicleParams), ("weight", C.c_float * WL_MAX_REW_TERMS),
        ("min_height", C.c_float), ("stuck_min_vel", C.c_float), ("stuck_wheel_spin", C.c_float),
        ("stuck_vel_cap", C.c_float), ("upright_cos", C.c_float), ("goal_dist", C.c_float), ("fall_vel", C.c_float),
        ("elev_z0", C.c_float), ("elev_min", C.c_float), ("elev_min_vel", C.c_float), ("progress_offset", C.c_float),
        ("reset_xy", C.c_float), ("reset_yaw", C.c_float), ("reset_vel", C.c_float * 2), ("reset_z", C.c_float),
        ("spawn_clearance", C.c_float), ("cmd_xy", C.c_float), ("cmd_heading", C.c_float),
        ("cmd_resample_s", C.c_float), ("scan_size", C.c_float), ("scan_res", C.c_float), ("scan_offset", C.c_float),
        ("obs_clip", C.c_float), ("log_episode_sums", C.c_int32),
    ]


class WlTravMap(C.Structure):
    _fields_ = [("map", C.c_void_p), ("cells", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32),
                ("n_cells", C.c_int32), ("row_spacing", C.c_float), ("col_spacing", C.c_float), ("bits", C.c_void_p)]


class WlVisualParams(C.Structure):
    _fields_ = [
        ("sim_dt", C.c_float), ("decimation", C.c_int32), ("max_episode_length", C.c_int32),
        ("action", WlActionParams), ("vehicle", WlVehicleParams), ("weight", C.c_float * WL_MAX_REW_TERMS),
        ("reset_z", C.c_float), ("cam_pos", C.c_float * 3), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
        ("cy", C.c_float), ("sky", C.c_float), ("brightness", C.c_float), ("contrast", C.c_float),
        ("blur_sigma", C.c_float), ("contrast_first", C.c_int32), ("log_episode_sums", C.c_int32),
    ]


VIS_NPIX = 40 * 80
VIS_OBS_DIM = VIS_NPIX + 8
VISDEPTH_NPIX = 60 * 80                 # visual-depth extension task (BASELINE config 5): the uncropped depth image ...
VISDEPTH_OBS_DIM = VISDEPTH_NPIX + 8    # ... | base_lin_vel | base_ang_vel | last_action


class WlEnvBuffers(C.Structure):
    _fields_ = [("state", C.c_void_p), ("episode_len", C.c_void_p), ("ref_poses", C.c_void_p),
                ("metrics", C.c_void_p), ("stride", C.c_int64), ("n_envs", C.c_int32), ("env_offset", C.c_int32),
                ("metrics_slots", C.c_int32), ("lanes", C.c_int32), ("flags", C.c_int32)]


class WlStepOut(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("reward", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p),
                ("dones", C.c_void_p)]


ACT_RELU, ACT_ELU = 0, 1


class WlMlp(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p),
                ("b3", C.c_void_p), ("in_dim", C.c_int32), ("out_dim", C.c_int32), ("hidden", C.c_int32),
                ("activation", C.c_int32)]


class WlPolicyRollout(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("actions", C.c_void_p), ("mu", C.c_void_p), ("log_prob", C.c_void_p),
                ("reward", C.c_void_p), ("terminated", C.c_void_p), ("truncated", C.c_void_p), ("dones", C.c_void_p)]


PPO_NUM_PARAMS, PPO_PARTIAL_STRIDE, PPO_BLOCKS, PPO_OPERAND_FLOATS = 10437, 10440, 256, 23296
PPO_CTRL_LR, PPO_CTRL_NORM2, PPO_CTRL_STATS = 0, 2, 4


class WlPpoBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs", "actions", "mu_old", "logp_old", "adv", "returns", "values_old", "perm",
                                          "sigma_old")]


class WlPpoParams(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("clip", "value_loss_coef", "entropy_coef", "desired_kl", "max_grad_norm", "beta1",
                                         "beta2", "eps", "lr_min", "lr_max")] + [("use_clipped_value_loss", C.c_int32),
                                                                                 ("adaptive", C.c_int32)]


class WlStartupParams(C.Structure):
    _fields_ = [("wheel_mu_s", C.c_float * 2), ("wheel_mu_d", C.c_float * 2), ("mu_buckets", C.c_int32),
                ("mu_consistent", C.c_int32), ("damping", C.c_float * 2), ("chassis_mass", C.c_float),
                ("mass_add", C.c_float * 2), ("randomize", C.c_int32), ("wheel_mass", C.c_float * 2)]


class WlPpoState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("partials", "grad", "adam_m", "adam_v", "ctrl", "operands")]


class WlCollectIo(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs_in", "actions", "mu", "log_prob", "values")]


class WlActScratch(C.Structure):
    _fields_ = [("w_hi", C.c_void_p), ("w_lo", C.c_void_p), ("partials", C.c_void_p), ("dp", C.c_int32), ("splits", C.c_int32),
                ("rows_capacity", C.c_int32), ("reserved", C.c_int32)]


class WlPpoWideState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("xt_hi", "xt_lo", "w_hi", "w_lo", "h1", "dt_hi", "dt_lo", "dw_partials",
                                          "partials", "narrow", "grad", "adam_m", "adam_v", "ctrl", "operands")] + [
        (n, C.c_int32) for n in ("in_dim", "dp", "capacity", "mb_capacity", "splits")]


_P = C.POINTER
_vp, _u64, _i32, _i64 = C.c_void_p, C.c_uint64, C.c_int32, C.c_int64

# every symbol include/wheeledlab_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "wl_version": (C.c_int, []),
    "wl_device_count": (C.c_int, []),
    "wl_strerror": (C.c_char_p, [C.c_int]),
    "wl_drift_step": (C.c_int, [_P(WlDriftParams), _P(WlEnvBuffers), _vp, _vp, _P(WlStepOut), _u64, _u64, _vp]),
    "wl_drift_rollout": (C.c_int, [_P(WlDriftParams), _P(WlEnvBuffers), _vp, _P(WlStepOut), _i64, _i64, _i32, _u64,
                                   _u64, _vp]),
    "wl_drift_rollout_persistent": (C.c_int, [_P(WlDriftParams), _P(WlEnvBuffers), _vp, _P(WlStepOut), _i64, _i64, _i32,
                                              _u64, _u64, _vp]),
    "wl_mlp_forward": (C.c_int, [_P(WlMlp), _i32, _vp, _vp, _vp]),
    "wl_drift_rollout_policy": (C.c_int, [_P(WlDriftParams), _P(WlEnvBuffers), _P(WlMlp), _vp, _P(WlPolicyRollout), _i32,
                                          _u64, _u64, _vp]),
    "wl_actor_critic_act": (C.c_int, [_P(WlMlp), _P(WlMlp), _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _u64, _u64, _i32, _i32,
                                      _vp]),
    "wl_actor_critic_planes": (C.c_int, [_P(WlMlp), _P(WlMlp), _P(WlActScratch), _vp]),
    "wl_actor_critic_act_planes": (C.c_int, [_P(WlMlp), _P(WlMlp), _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _u64, _u64, _i32,
                                             _i32, _P(WlActScratch), _vp]),
    "wl_gae": (C.c_int, [_i32, _i32, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp]),
    "wl_rollout_bookkeeping": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _vp, _vp]),
    "wl_ppo_gradients": (C.c_int, [_P(WlMlp), _P(WlMlp), _vp, _P(WlPpoBatch), _i32, _i32, _P(WlPpoParams), _P(WlPpoState), _i32,
                                   _vp]),
    "wl_ppo_minibatch": (C.c_int, [_P(WlMlp), _P(WlMlp), _vp, _P(WlPpoBatch), _i32, _i32, _P(WlPpoParams), _P(WlPpoState), _i32,
                                   _i32, _vp]),
    "wl_ppo_apply": (C.c_int, [_P(WlMlp), _P(WlMlp), _vp, _i32, _P(WlPpoParams), _P(WlPpoState), _i32, _i32, _vp]),
    "wl_ppo_wide_num_params": (C.c_int32, [_i32]),
    "wl_ppo_wide_stage": (C.c_int, [_vp, _vp, _i32, _P(WlPpoWideState), _vp]),
    "wl_ppo_wide_gradients": (C.c_int, [_P(WlMlp), _P(WlMlp), _vp, _P(WlPpoBatch), _i32, _i32, _P(WlPpoParams),
                                        _P(WlPpoWideState), _i32, _vp]),
    "wl_ppo_wide_apply": (C.c_int, [_P(WlMlp), _P(WlMlp), _vp, _i32, _P(WlPpoParams), _P(WlPpoWideState), _i32, _i32, _vp]),
    "wl_ppo_wide_minibatch": (C.c_int, [_P(WlMlp), _P(WlMlp), _vp, _P(WlPpoBatch), _i32, _i32, _P(WlPpoParams),
                                        _P(WlPpoWideState), _i32, _i32, _vp]),
    "wl_drift_mdp": (C.c_int, [_P(WlDriftParams), _i32, _i64] + [_vp] * 12),
    "wl_action_map": (C.c_int, [_P(WlActionParams), _i32, _vp, _vp, _vp, _vp, _vp]),
    "wl_drift_reset": (C.c_int, [_P(WlDriftParams), _P(WlEnvBuffers), _vp, _u64, _u64, _vp]),
    "wl_drift_observe": (C.c_int, [_P(WlDriftParams), _P(WlEnvBuffers), _vp, _vp, _u64, _u64, _vp]),
    "wl_philox_uniform": (C.c_int, [_i32, _u64, _u64, C.c_uint32, _vp, _vp]),
    "wl_startup_randomize": (C.c_int, [_P(WlStartupParams), _P(WlEnvBuffers), _u64, _vp]),
    "wl_elev_step": (C.c_int, [_P(WlElevParams), _P(WlEnvBuffers), _P(WlHeightField), _vp, _P(WlStepOut), _u64, _u64, _vp]),
    "wl_elev_rollout": (C.c_int, [_P(WlElevParams), _P(WlEnvBuffers), _P(WlHeightField), _vp, _P(WlStepOut), _i64, _i64,
                                  _i32, _u64, _u64, _vp]),
    "wl_elev_collect_step": (C.c_int, [_P(WlElevParams), _P(WlEnvBuffers), _P(WlHeightField), _P(WlMlp), _P(WlMlp), _vp, _P(WlCollectIo),
                                       _P(WlStepOut), _i32, _u64, _u64, _vp]),
    "wl_elev_collect_rollout": (C.c_int, [_P(WlElevParams), _P(WlEnvBuffers), _P(WlHeightField), _P(WlMlp), _P(WlMlp), _vp, _P(WlCollectIo),
                                          _P(WlStepOut), _i32, _i32, _u64, _u64, _vp]),
    "wl_elev_rollout_persistent": (C.c_int, [_P(WlElevParams), _P(WlEnvBuffers), _P(WlHeightField), _vp, _P(WlStepOut), _i64, _i64,
                                             _i32, _u64, _u64, _vp]),
    "wl_elev_reset": (C.c_int, [_P(WlElevParams), _P(WlEnvBuffers), _P(WlHeightField), _vp, _u64, _u64, _vp]),
    "wl_elev_observe": (C.c_int, [_P(WlElevParams), _P(WlEnvBuffers), _P(WlHeightField), _vp, _vp]),
    "wl_elev_mdp": (C.c_int, [_P(WlElevParams), _i32, _i64] + [_vp] * 7 + [_i32] + [_vp] * 7),
    "wl_visual_step": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlTravMap), _vp, _P(WlStepOut), _u64, _u64, _vp]),
    "wl_visual_rollout": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlTravMap), _vp, _P(WlStepOut), _i64, _i64,
                                    _i32, _u64, _u64, _vp]),
    "wl_visual_rollout_persistent": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlTravMap), _vp, _P(WlStepOut), _i64, _i64,
                                               _i32, _u64, _u64, _vp]),
    "wl_visual_reset": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlTravMap), _vp, _u64, _u64, _vp]),
    "wl_visual_observe": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlTravMap), _vp, _vp]),
    "wl_visual_mdp": (C.c_int, [_P(WlVisualParams), _P(WlTravMap), _i32, _i64] + [_vp] * 7),
    "wl_heightfield_pairs": (C.c_int, [_P(WlHeightField), _vp, _vp]),
    "wl_heightfield_pyramid_floats": (C.c_int64, [_i32, _i32]),
    "wl_heightfield_build_pyramid": (C.c_int, [_P(WlHeightField), _vp, _vp]),
    "wl_visual_depth": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlHeightField), _vp, C.c_float, _vp, _vp]),
    "wl_visual_depth_rows": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlHeightField), _vp, C.c_float, _vp, _i64, _vp]),
    "wl_visual_step_hf": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlTravMap), _P(WlHeightField), _vp, _P(WlStepOut), _u64, _u64, _vp]),
    "wl_visual_reset_hf": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlTravMap), _P(WlHeightField), _vp, _u64, _u64, _vp]),
    "wl_visual_depth_step": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlTravMap), _P(WlHeightField), _vp, C.c_float, _vp,
                                       _P(WlStepOut), _u64, _u64, _vp]),
    "wl_visual_depth_observe": (C.c_int, [_P(WlVisualParams), _P(WlEnvBuffers), _P(WlHeightField), _vp, C.c_float, _vp, _vp]),
}

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwheeledlab_amd.so")
_lib = None


def load(path: str | None = None):
    """dlopen the HIP library and bind every declared symbol.  Raises HipExtensionMissing -- never falls back."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    # PyTorch (the owner of the HBM allocations and streams we are handed) bundles its own libamdhip64.so with the same
    # SONAME as /opt/rocm's.  It must be in the process BEFORE our library is dlopen'ed so that both bind to ONE HIP
    # runtime; the other order gives two runtimes and every launch on torch memory fails.
    import torch  # noqa: F401
    if not os.path.exists(path):
        raise HipExtensionMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        lib = C.CDLL(path)
    except OSError as e:  # e.g. libamdhip64 missing
        raise HipExtensionMissing(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipExtensionMissing(f"{path} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    v = lib.wl_version()
    if v != WL_ABI_VERSION:
        raise HipExtensionMissing(f"{path} has ABI version {v}, python expects {WL_ABI_VERSION}: rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        raise WlError(f"{what} failed: {ERRORS.get(rc, rc)} ({rc})")
