"""Parameter sets of the oracle (plain namespaces).  Values restate the reference configs (cited) or, for the
vehicle model, the designed constants of DESIGN.md section 4.  Field names mirror include/wheeledlab_amd.h so a
ctypes struct from the product can be passed to the oracle functions instead (duck typing)."""
import math
from types import SimpleNamespace as NS


def mushr_vehicle(drive=0, motor_limit=0.5, substeps=1, ground_mu=(1.1, 1.0), implicit=0):
    k = 3000.0
    m_nom = 3.4                       # 3.0 kg chassis + mean of the U(0.3,0.5) added mass (mushr_drift_env_cfg.py:145-154)
    g = 9.81
    r = 0.05                          # common/actions.py:20
    return NS(
        gravity=g, half_wheelbase_f=0.1625, half_wheelbase_r=0.1625, half_track=0.1, wheel_radius=r,
        wheel_z=r - m_nom * g / (4 * k), cg_z=0.06, gyr_x=0.06, gyr_y=0.12, gyr_z=0.13,
        wheel_inertia=8e-5, wheel_damping=1e-4, susp_k=k, susp_c=60.0, susp_fmax=24.0 * m_nom * g / 4.0,
        ground_mu_s=ground_mu[0], ground_mu_d=ground_mu[1], slip_peak=0.12, v_min=0.25,
        motor_sat=1.05, motor_limit=motor_limit, motor_vel_limit=450.0,   # hound.py:13-21,40-43
        drive=drive,
        steer_kp=100.0, steer_kd=10.0, steer_effort=3.2, steer_vel_limit=10.0, steer_inertia=2e-4,  # hound.py:5-12
        substeps=substeps, implicit=implicit,   # integrator: oracle/vehicle.py
    )


def mushr_action(map_=0, base_length=0.325, base_width=0.2):
    # common/actions.py:5-57 ; scale override mushr_drift_env_cfg.py:397
    return NS(scale=[3.0, 0.488], offset=[0.0, 0.0], bounding=1, no_reverse=1, clip_wrapper=1, map=map_,
              base_length=base_length, base_width=base_width, wheel_radius=0.05)


def drift_params():
    """RSS_DRIFT_CONFIG env (mushr_drift_env_cfg.py)"""
    return NS(
        sim_dt=0.005, decimation=4, max_episode_length=math.ceil(5.0 / (0.005 * 4)),   # :393-396
        action=mushr_action(0), vehicle=mushr_vehicle(drive=0, motor_limit=0.5),
        straight=0.8, r_in=0.3, r_out=2.0, r_line=0.8,                                 # :27-30
        weight=[10.0, -5.0, 40.0, 0.0, 20.0, -50.0, -5000.0, 0.0],                     # :246-299
        slip_min=0.25, slip_max=0.55, slip_min_vx=1.0,                                 # :249-253
        speed_target=3.0, speed_offset=-9.0,                                           # :167, :256-263
        tlgr_thresh=1.0,                                                               # :272
        ctd_offset=-1.0, ctd_p=1.0,                                                    # :284-293
        enable_corruption=1, noise_std=[0.1, 0.1, 0.5, 0.4],                           # common/observations.py:27-45, :399
        num_ref_points=20, pos_noise=0.5, yaw_noise=1.0,                               # :82-93
        enable_pushes=1, hf_interval=[0.1, 0.4], hf_vel_x=0.1, hf_vel_y=0.03, hf_vel_yaw=0.3,   # :121-132
        lf_interval=[0.8, 1.2], lf_vel_yaw=0.6,                                        # :134-143
        log_episode_sums=1,
    )
