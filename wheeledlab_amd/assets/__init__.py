"""Robot asset configs: actuator constants of the reference's `wheeledlab_assets` (hound.py:4-52, mushr.py:19-60).
The USD meshes are not part of this implementation (and are missing from the reference snapshot): geometry / mass
live in `wheeledlab_amd.params.mushr_vehicle`."""
from ..envs.managers_cfg import ArticulationCfg, DCMotorCfg, ImplicitActuatorCfg
from ..envs.scene import MUSHR_JOINT_NAMES

_STEER = ImplicitActuatorCfg(joint_names_expr=["front_left_wheel_steer", "front_right_wheel_steer"], velocity_limit=10.0,
                             effort_limit=3.2, stiffness=100.0, damping=10.0, friction=0.0)
_THROTTLE = DCMotorCfg(joint_names_expr=[".*throttle"], saturation_effort=1.05, effort_limit=0.25, velocity_limit=450.0,
                       stiffness=0, damping=1000.0, friction=0.0)
_SUSPENSION = ImplicitActuatorCfg(joint_names_expr=[".*_suspension"], effort_limit=None, velocity_limit=None,
                                  stiffness=1e8, damping=0.0, friction=0.5)

HOUND_ACTUATOR_CFG = {"steering_joints": _STEER, "throttle_joints": _THROTTLE}
HOUND_SUS_ACTUATOR_CFG = {**HOUND_ACTUATOR_CFG, "suspension": _SUSPENSION}
HOUND_SUS_2WD_ACTUATOR_CFG = {
    "steering_joints": _STEER,
    "suspension": _SUSPENSION,
    "throttle_joints": _THROTTLE.replace(joint_names_expr=["back_.*throttle"], effort_limit=0.5),
    "passive_joints": ImplicitActuatorCfg(joint_names_expr=["front_.*throttle"], effort_limit=None, velocity_limit=None,
                                          stiffness=0.0, damping=0.0, friction=0.0),
}

MUSHR_CFG = ArticulationCfg(usd_path="Robots/UWPRL/mushr_nano.usd", joint_names=MUSHR_JOINT_NAMES[:6],
                            actuators=HOUND_ACTUATOR_CFG)
MUSHR_SUS_CFG = MUSHR_CFG.replace(usd_path="Robots/UWRLL/mushr_nano_v2.usd", joint_names=list(MUSHR_JOINT_NAMES),
                                  actuators=HOUND_SUS_ACTUATOR_CFG)
MUSHR_SUS_2WD_CFG = MUSHR_SUS_CFG.replace(actuators=HOUND_SUS_2WD_ACTUATOR_CFG)

# F1Tenth (wheeledlab_assets/f1tenth.py:9-64): 4WD, all throttle joints driven
F1TENTH_JOINT_NAMES = ["rotator_left", "rotator_right", "wheel_back_left", "wheel_back_right", "wheel_front_left",
                       "wheel_front_right"]
F1TENTH_4WD_ACTUATOR_CFG = {
    "steering_joints": ImplicitActuatorCfg(joint_names_expr=["rotator_(left|right)"], velocity_limit=10.0, effort_limit=2.5,
                                           stiffness=120.0, damping=8.0, friction=0.0),
    "throttle_joints": DCMotorCfg(joint_names_expr=[".*wheel_(back|front)_.*"], saturation_effort=1.0, effort_limit=0.25,
                                  velocity_limit=400.0, stiffness=0, damping=1100.0, friction=0.0),
}
F1TENTH_CFG = ArticulationCfg(usd_path="Robots/F1TENTH/f1tenth.usd", joint_names=F1TENTH_JOINT_NAMES,
                              actuators=F1TENTH_4WD_ACTUATOR_CFG)
