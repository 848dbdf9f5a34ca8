"""Manager term config classes with the field names of `isaaclab.managers` that the reference's task configs use
(RewTerm / DoneTerm / ObsTerm / EventTerm / CurrTerm / SceneEntityCfg, noise and actuator / material configs)."""
from __future__ import annotations

from .configclass import MISSING, configclass


class SceneEntityCfg:
    def __init__(self, name: str, joint_names=None, body_names=None, joint_ids=slice(None), body_ids=slice(None)):
        self.name, self.joint_names, self.body_names = name, joint_names, body_names
        self.joint_ids, self.body_ids = joint_ids, body_ids

    def resolve(self, scene):
        if self.joint_names is not None:
            self.joint_ids = scene[self.name].find_joints(self.joint_names)[0]
        if self.body_names is not None:
            self.body_ids = scene[self.name].find_bodies(self.body_names)[0]
        return self


@configclass
class ManagerTermBaseCfg:
    func = MISSING
    params: dict = {}


@configclass
class RewardTermCfg(ManagerTermBaseCfg):
    weight: float = MISSING


@configclass
class TerminationTermCfg(ManagerTermBaseCfg):
    time_out: bool = False


@configclass
class CurriculumTermCfg(ManagerTermBaseCfg):
    pass


@configclass
class EventTermCfg(ManagerTermBaseCfg):
    mode: str = MISSING
    interval_range_s = None
    is_global_time: bool = False
    min_step_count_between_reset: int = 0


@configclass
class NoiseCfg:
    pass


@configclass
class AdditiveGaussianNoiseCfg(NoiseCfg):
    mean: float = 0.0
    std: float = 1.0


@configclass
class AdditiveUniformNoiseCfg(NoiseCfg):
    n_min: float = -1.0
    n_max: float = 1.0


GaussianNoiseCfg, UniformNoiseCfg = AdditiveGaussianNoiseCfg, AdditiveUniformNoiseCfg


@configclass
class ObservationTermCfg(ManagerTermBaseCfg):
    noise = None
    clip = None
    scale = None


@configclass
class ObservationGroupCfg:
    concatenate_terms: bool = True
    enable_corruption: bool = False


@configclass
class ActionTermCfg:
    class_type = MISSING
    asset_name: str = "robot"


class ManagerTermBase:
    """class-type terms: constructed with (cfg, env), called with (env, env_ids, **params)
    (reference: drifting/mdp/events.py:10,15,102)"""

    def __init__(self, cfg, env):
        self.cfg, self._env = cfg, env

    @property
    def device(self):
        return self._env.device

    @property
    def num_envs(self):
        return self._env.num_envs

    def reset(self, env_ids=None):
        pass


# ---- scene / sim / asset configs (values only; there is no USD stage here) -------------------------------------

@configclass
class RigidBodyMaterialCfg:
    static_friction: float = 0.5
    dynamic_friction: float = 0.5
    restitution: float = 0.0
    friction_combine_mode: str = "average"
    restitution_combine_mode: str = "average"


@configclass
class TerrainImporterCfg:
    prim_path: str = "/World/ground"
    terrain_type: str = "plane"
    height: float = 0.0
    collision_group: int = -1
    physics_material: RigidBodyMaterialCfg = RigidBodyMaterialCfg()
    debug_vis: bool = False


@configclass
class ImplicitActuatorCfg:
    joint_names_expr: list = []
    effort_limit = None
    velocity_limit = None
    stiffness = None
    damping = None
    friction: float = 0.0


@configclass
class DCMotorCfg(ImplicitActuatorCfg):
    saturation_effort: float = MISSING


@configclass
class ArticulationCfg:
    prim_path: str = ""
    usd_path: str = ""
    init_pos: tuple = (0.0, 0.0, 0.0)
    joint_names: list = []
    actuators: dict = {}
    solver_position_iteration_count: int = 4
    solver_velocity_iteration_count: int = 0


@configclass
class InteractiveSceneCfg:
    num_envs: int = MISSING
    env_spacing: float = 0.0


@configclass
class SimulationCfg:
    dt: float = 1.0 / 60.0
    render_interval: int = 1
    device: str = "cuda:0"
    gravity: tuple = (0.0, 0.0, -9.81)


@configclass
class ViewerCfg:
    eye: list = [7.5, 7.5, 7.5]
    lookat: list = [0.0, 0.0, 0.0]


@configclass
class ManagerBasedRLEnvCfg:
    """fields of isaaclab.envs.ManagerBasedRLEnvCfg the reference sets (e.g. mushr_drift_env_cfg.py:369-404)"""
    seed = None
    sim: SimulationCfg = SimulationCfg()
    viewer: ViewerCfg = ViewerCfg()
    decimation: int = MISSING
    episode_length_s: float = MISSING
    scene = MISSING
    observations = MISSING
    actions = MISSING
    rewards = MISSING
    terminations = MISSING
    events = None
    curriculum = None
    commands = None
    is_finite_horizon: bool = False
    rerender_on_reset: bool = False
    # additions of this implementation
    sync_episode_log: bool = False   # True: host-sync every step and emit extras["log"] only on reset steps (IsaacLab)
    metrics_slots: int = 512         # per-step metric ring length (>= the consumer's logging window)
