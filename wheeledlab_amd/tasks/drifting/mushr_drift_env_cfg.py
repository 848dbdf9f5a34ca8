"""Drift task configuration (RSS_DRIFT_CONFIG).  Same config surface as the reference's
wheeledlab_tasks/drifting/mushr_drift_env_cfg.py (class names, term names, parameter values; line citations inline)
so Hydra-style overrides such as `env.rewards.side_slip.weight=100` keep their meaning.  All term functions are the
kernel-backed ones of `wheeledlab_amd.envs.mdp`; nothing here does tensor math."""
from ...assets import MUSHR_SUS_2WD_CFG
from ...envs import mdp
from ...envs.configclass import configclass
from ...envs.managers_cfg import CurriculumTermCfg as CurrTerm
from ...envs.managers_cfg import EventTermCfg as EventTerm
from ...envs.managers_cfg import (InteractiveSceneCfg, ManagerBasedRLEnvCfg, RigidBodyMaterialCfg, SceneEntityCfg,
                                  TerrainImporterCfg)
from ...envs.managers_cfg import RewardTermCfg as RewTerm
from ...envs.managers_cfg import TerminationTermCfg as DoneTerm
from ..common import BlindObsCfg, MushrRWDActionCfg

# track + task constants (:27-32)
CORNER_IN_RADIUS, CORNER_OUT_RADIUS = 0.3, 2.0   # termination bounds
LINE_RADIUS, STRAIGHT = 0.8, 0.8                  # centre line: spawn + cross-track reward
SLIP_THRESHOLD = 0.55                             # rad
MAX_SPEED = 3.0                                   # m/s


@configclass
class DriftTerrainImporterCfg(TerrainImporterCfg):
    """flat carpet plane, friction combined by multiplication (:39-51)"""
    height = 0.0
    terrain_type = "plane"
    physics_material = RigidBodyMaterialCfg(friction_combine_mode="multiply", restitution_combine_mode="multiply",
                                            static_friction=1.1, dynamic_friction=1.0)


@configclass
class MushrDriftSceneCfg(InteractiveSceneCfg):
    terrain = DriftTerrainImporterCfg()
    robot = MUSHR_SUS_2WD_CFG.replace(prim_path="{ENV_REGEX_NS}/Robot")


@configclass
class DriftEventsCfg:
    reset_root_state = EventTerm(                                                   # :82-93
        func=mdp.reset_root_state_along_track, mode="reset",
        params=dict(track_radius=LINE_RADIUS, track_straight_dist=STRAIGHT, num_points=20, pos_noise=0.5, yaw_noise=1.0,
                    asset_cfg=SceneEntityCfg("robot")))


@configclass
class DriftEventsRandomCfg(DriftEventsCfg):
    change_wheel_friction = EventTerm(                                              # :98-109
        func=mdp.randomize_rigid_body_material, mode="startup",
        params=dict(static_friction_range=(0.3, 0.5), dynamic_friction_range=(0.3, 0.5), restitution_range=(0.0, 0.0),
                    num_buckets=20, asset_cfg=SceneEntityCfg("robot", body_names=".*wheel_link"), make_consistent=True))
    randomize_gains = EventTerm(                                                    # :111-119
        func=mdp.randomize_actuator_gains, mode="startup",
        params=dict(asset_cfg=SceneEntityCfg("robot", joint_names=[".*back.*throttle"]),
                    damping_distribution_params=(10.0, 50.0), operation="abs"))
    push_robots_hf = EventTerm(                                                     # :121-132 small frequent pushes
        func=mdp.push_by_setting_velocity, mode="interval", interval_range_s=(0.1, 0.4),
        params=dict(velocity_range={"x": (-0.1, 0.1), "y": (-0.03, 0.03), "yaw": (-0.3, 0.3)}))
    push_robots_lf = EventTerm(                                                     # :134-143 rare yaw kicks
        func=mdp.push_by_setting_velocity, mode="interval", interval_range_s=(0.8, 1.2),
        params=dict(velocity_range={"yaw": (-0.6, 0.6)}))
    add_base_mass = EventTerm(                                                      # :145-154
        func=mdp.randomize_rigid_body_mass, mode="startup",
        params=dict(asset_cfg=SceneEntityCfg("robot", body_names=["base_link"]), mass_distribution_params=(0.3, 0.5),
                    operation="add", distribution="uniform"))


@configclass
class DriftRewardsCfg:
    """:246-299"""
    side_slip = RewTerm(func=mdp.side_slip, weight=10.0,
                        params=dict(min_thresh=0.25, max_thresh=SLIP_THRESHOLD, min_vel_x=1.0))
    vel = RewTerm(func=mdp.vel_dist, weight=-5.0, params=dict(speed_target=MAX_SPEED))
    progress = RewTerm(func=mdp.track_progress_rate, weight=40.0)
    tlgr = RewTerm(func=mdp.turn_left_go_right, weight=0.0, params=dict(ang_vel_thresh=1.0))
    turn_energy = RewTerm(func=mdp.energy_through_turn, weight=20.0, params=dict(straight=STRAIGHT))
    cross_track = RewTerm(func=mdp.cross_track_dist, weight=-50.0,
                          params=dict(straight=STRAIGHT, track_radius=LINE_RADIUS, p=1, offset=-1.0))
    term_pens = RewTerm(func=mdp.rewards.is_terminated_term, weight=-5000.0, params=dict(term_keys=["out_of_bounds"]))


def _ramp(term, increase, every, stop):
    return CurrTerm(func=mdp.increase_reward_weight_over_time,
                    params=dict(reward_term_name=term, increase=increase, episodes_per_increase=every, max_increases=stop))


@configclass
class DriftCurriculumCfg:
    """:309-337"""
    more_slip = _ramp("side_slip", 20.0, 20, 10)
    more_tlgr = _ramp("tlgr", 10.0, 20, 5)
    more_term_pens = _ramp("term_pens", -1000.0, 50, 5)


@configclass
class DriftTerminationsCfg:
    """:351-362"""
    time_out = DoneTerm(func=mdp.time_out, time_out=True)
    out_of_bounds = DoneTerm(func=mdp.cart_off_track,
                             params=dict(straight=STRAIGHT, corner_in_radius=CORNER_IN_RADIUS,
                                         corner_out_radius=CORNER_OUT_RADIUS))


@configclass
class MushrDriftRLEnvCfg(ManagerBasedRLEnvCfg):
    """:369-404"""
    seed: int = 42
    num_envs: int = 1024
    env_spacing: float = 0.0
    observations: BlindObsCfg = BlindObsCfg()
    actions: MushrRWDActionCfg = MushrRWDActionCfg()
    rewards: DriftRewardsCfg = DriftRewardsCfg()
    events: DriftEventsCfg = DriftEventsRandomCfg()
    terminations: DriftTerminationsCfg = DriftTerminationsCfg()
    curriculum: DriftCurriculumCfg = DriftCurriculumCfg()

    def __post_init__(self):
        self.viewer.eye, self.viewer.lookat = [4.0, -4.0, 4.0], [0.0, 0.0, 0.0]
        self.sim.dt = 0.005            # 200 Hz physics
        self.decimation = 4            # 50 Hz control
        self.sim.render_interval = 20
        self.episode_length_s = 5
        self.actions.throttle_steer.scale = (MAX_SPEED, 0.488)
        self.observations.policy.enable_corruption = True
        self.scene = MushrDriftSceneCfg(num_envs=self.num_envs, env_spacing=self.env_spacing)


@configclass
class MushrDriftPlayEnvCfg(MushrDriftRLEnvCfg):
    """evaluation variant: no rewards / terminations / curriculum, noise-free spawn (:411-430)"""
    events: DriftEventsCfg = DriftEventsRandomCfg(
        reset_root_state=EventTerm(func=mdp.reset_root_state_along_track, mode="reset",
                                   params=dict(pos_noise=0.0, yaw_noise=0.0)))
    rewards = None
    terminations = None
    curriculum = None
