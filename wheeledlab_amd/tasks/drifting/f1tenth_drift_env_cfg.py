"""F1Tenth drift variant (reference: wheeledlab_tasks/drifting/f1tenth_drift_env_cfg.py:42-161,
wheeledlab_assets/f1tenth.py:9-27, common/actions.py:50-71): same task as the MuSHR drift env with a 4WD action term,
wheelbase 0.365 / track 0.284, the F1Tenth actuator constants, and all four throttle gains randomised."""
from ...assets import F1TENTH_CFG
from ...envs import mdp
from ...envs.configclass import configclass
from ...envs.managers_cfg import EventTermCfg as EventTerm
from ...envs.managers_cfg import RewardTermCfg as RewTerm
from ...envs.managers_cfg import SceneEntityCfg
from ..common import F1Tenth4WDActionCfg
from .mushr_drift_env_cfg import DriftEventsRandomCfg, DriftRewardsCfg, MushrDriftRLEnvCfg, MushrDriftSceneCfg, MAX_SPEED


@configclass
class F1TenthDriftSceneCfg(MushrDriftSceneCfg):
    robot = F1TENTH_CFG.replace(prim_path="{ENV_REGEX_NS}/Robot")


@configclass
class F1TenthDriftEventsRandomCfg(DriftEventsRandomCfg):
    randomize_gains = EventTerm(                                                       # :57-65 all four wheel motors
        func=mdp.randomize_actuator_gains, mode="startup",
        params=dict(asset_cfg=SceneEntityCfg("robot", joint_names=["wheel_(back|front)_.*"]),
                    damping_distribution_params=(10.0, 50.0), operation="abs"))
    change_wheel_friction = EventTerm(                                                 # :67-78
        func=mdp.randomize_rigid_body_material, mode="startup",
        params=dict(static_friction_range=(0.3, 0.5), dynamic_friction_range=(0.3, 0.5), restitution_range=(0.0, 0.0),
                    num_buckets=20, asset_cfg=SceneEntityCfg("robot", body_names="wheel.*"), make_consistent=True))


@configclass
class F1TenthDriftRewardsCfg(DriftRewardsCfg):
    # turn_left_go_right_f1 (:94-109) differs from the MuSHR term only in the steering joint names
    tlgr = RewTerm(func=mdp.turn_left_go_right, params=dict(ang_vel_thresh=1.0), weight=0.0)


@configclass
class F1TenthDriftRLEnvCfg(MushrDriftRLEnvCfg):
    """:132-161"""
    num_envs: int = 256
    actions: F1Tenth4WDActionCfg = F1Tenth4WDActionCfg()
    rewards: F1TenthDriftRewardsCfg = F1TenthDriftRewardsCfg()
    events: F1TenthDriftEventsRandomCfg = F1TenthDriftEventsRandomCfg()

    def __post_init__(self):
        MushrDriftRLEnvCfg.__post_init__(self)
        self.actions.throttle_steer.scale = (MAX_SPEED, 0.488)
        self.scene = F1TenthDriftSceneCfg(num_envs=self.num_envs, env_spacing=self.env_spacing)
