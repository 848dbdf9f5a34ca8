"""Shared observation / action configs of the tasks (reference: wheeledlab_tasks/common/{observations,actions}.py)."""
from ...envs import mdp
from ...envs.actions import RCCar4WDActionCfg, RCCarRWDActionCfg
from ...envs.configclass import configclass
from ...envs.managers_cfg import AdditiveGaussianNoiseCfg as Gnoise
from ...envs.managers_cfg import ObservationGroupCfg as ObsGroup
from ...envs.managers_cfg import ObservationTermCfg as ObsTerm

_MUSHR_GEOM = dict(base_length=0.325, base_width=0.2, wheel_radius=0.05, scale=(3.0, 0.488), no_reverse=True,
                   bounding_strategy="clip", asset_name="robot")
_STEER_JOINTS = ["front_left_wheel_steer", "front_right_wheel_steer"]
_REAR = ["back_left_wheel_throttle", "back_right_wheel_throttle"]
_FRONT = ["front_left_wheel_throttle", "front_right_wheel_throttle"]


@configclass
class MushrRWDActionCfg:
    throttle_steer = RCCarRWDActionCfg(wheel_joint_names=_REAR, steering_joint_names=_STEER_JOINTS, **_MUSHR_GEOM)


@configclass
class Mushr4WDActionCfg:
    throttle_steer = RCCar4WDActionCfg(wheel_joint_names=_REAR + _FRONT, steering_joint_names=_STEER_JOINTS, **_MUSHR_GEOM)


@configclass
class F1Tenth4WDActionCfg:
    throttle_steer = RCCar4WDActionCfg(
        wheel_joint_names=["wheel_back_left", "wheel_back_right", "wheel_front_left", "wheel_front_right"],
        steering_joint_names=["rotator_left", "rotator_right"],
        **{**_MUSHR_GEOM, "base_length": 0.365, "base_width": 0.284})


@configclass
class BlindObsCfg:
    """14-dim proprioceptive observation with the empirically chosen Gaussian noise levels
    (common/observations.py:19-56); corruption is off until a task enables it."""

    @configclass
    class PolicyCfg(ObsGroup):
        root_pos_w_term = ObsTerm(func=mdp.root_pos_w, noise=Gnoise(mean=0.0, std=0.1))
        root_euler_xyz_term = ObsTerm(func=mdp.root_euler_xyz, noise=Gnoise(mean=0.0, std=0.1))
        base_lin_vel_term = ObsTerm(func=mdp.base_lin_vel, noise=Gnoise(mean=0.0, std=0.5))
        base_ang_vel_term = ObsTerm(func=mdp.base_ang_vel, noise=Gnoise(std=0.4))
        last_action_term = ObsTerm(func=mdp.last_action, clip=(-1.0, 1.0))

        def __post_init__(self):
            self.concatenate_terms = True
            self.enable_corruption = False

    policy: PolicyCfg = PolicyCfg()
