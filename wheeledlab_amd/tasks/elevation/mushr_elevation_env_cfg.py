"""Elevation task configuration -- same config surface (class / term names, values) as the reference's
wheeledlab_tasks/elevation/mushr_elevation_env_cfg.py; line citations inline.  Terms are the kernel-backed ones of
`wheeledlab_amd.envs.mdp`.  The terrain is a heightfield (the reference's USD mesh is missing): pass your own
`(height, x0, y0, cell)` in `scene.terrain.heightfield`, or leave None for the synthetic one."""
from ...assets import MUSHR_SUS_CFG
from ...envs import mdp
from ...envs.configclass import configclass
from ...envs.managers_cfg import CurriculumTermCfg as CurrTerm
from ...envs.managers_cfg import EventTermCfg as EventTerm
from ...envs.managers_cfg import (InteractiveSceneCfg, ManagerBasedRLEnvCfg, RigidBodyMaterialCfg, SceneEntityCfg,
                                  TerrainImporterCfg)
from ...envs.managers_cfg import ObservationGroupCfg as ObsGroup
from ...envs.managers_cfg import ObservationTermCfg as ObsTerm
from ...envs.managers_cfg import RewardTermCfg as RewTerm
from ...envs.managers_cfg import TerminationTermCfg as DoneTerm
from ...envs.sensors_cfg import GridPatternCfg, RayCasterCfg, UniformPose2dCommandCfg
from ..common import Mushr4WDActionCfg


@configclass
class ElevationObsCfg:
    """689-dim observation (:57-88): goal(2) | euler(3) | v_b(3) | w_b(3) | last action(2) | 26 x 26 height map"""

    @configclass
    class ConcatObs(ObsGroup):
        goal_relative_xyz = ObsTerm(func=mdp.goal_relative_xyz)
        world_euler_xyz = ObsTerm(func=mdp.root_euler_xyz)
        base_lin_vel = ObsTerm(func=mdp.base_lin_vel, clip=(-10.0, 10.0))
        base_ang_vel = ObsTerm(func=mdp.base_ang_vel, clip=(-10.0, 10.0))
        last_action = ObsTerm(func=mdp.last_action, clip=(-1.0, 1.0))
        elevation_map = ObsTerm(func=mdp.world_height_map, clip=(-10.0, 10.0),
                                params=dict(sensor_cfg=SceneEntityCfg("height_scanner"), offset=0.084, plane_init_value=0.19))

        def __post_init__(self):
            self.enable_corruption = False
            self.concatenate_terms = True

    policy: ConcatObs = ConcatObs()


@configclass
class ElevationTerrainImporterCfg(TerrainImporterCfg):
    """:94-108"""
    height = 0.25
    terrain_type = "heightfield"
    heightfield = None                 # (height[ny][nx] float32, x0, y0, cell); None -> wheeledlab_amd.terrain.synthetic_heightfield
    physics_material = RigidBodyMaterialCfg(friction_combine_mode="multiply", restitution_combine_mode="multiply",
                                            static_friction=1.0, dynamic_friction=1.0)


@configclass
class ElevationSceneCfg(InteractiveSceneCfg):
    terrain = ElevationTerrainImporterCfg()
    robot = MUSHR_SUS_CFG.replace(prim_path="{ENV_REGEX_NS}/Robot")
    height_scanner = RayCasterCfg(prim_path="{ENV_REGEX_NS}/Robot/mushr_nano/base_link", offset_pos=(0.0, 0.0, 20.0),
                                  attach_yaw_only=True, pattern_cfg=GridPatternCfg(size=[2.5, 2.5], resolution=0.1),
                                  mesh_prim_paths=["/World/elevation/terrain"])                     # :132-142


@configclass
class ElevationRewardsCfg:
    """:283-305"""
    vel_towards_goal = RewTerm(func=mdp.goal_progress_rate, weight=200.0)
    height_z = RewTerm(func=mdp.higher_elevation, weight=5000.0)
    falling_penalty = RewTerm(func=mdp.is_falling_penalty, weight=0.0)
    termination_penalty = RewTerm(func=mdp.rewards.is_terminated_term, params=dict(term_keys="stuck"), weight=-200.0)


def _ramp(term, increase, every, stop):
    return CurrTerm(func=mdp.increase_reward_weight_over_time,
                    params=dict(reward_term_name=term, increase=increase, episodes_per_increase=every, max_increases=stop))


@configclass
class ElevationCurriculumCfg:
    """:311-333"""
    more_goal = _ramp("vel_towards_goal", 5.0, 50, 5)
    more_falling_pen = _ramp("falling_penalty", 1.0, 50, 10)


@configclass
class ElevationTerminationsCfg:
    """:349-376"""
    time_out = DoneTerm(func=mdp.time_out, time_out=True)
    cart_out_of_bounds = DoneTerm(func=mdp.root_height_below_minimum, params=dict(minimum_height=0.15))
    stuck = DoneTerm(func=mdp.stuck, params=dict(min_vel=0.02, wheel_spin_thr=5.0))
    rollover = DoneTerm(func=mdp.upright_bool, params=dict(thresh_deg=60.0))
    at_goal = DoneTerm(func=mdp.close_to_goal, params=dict(dist=0.5))


@configclass
class ElevationSceneEventsCfg:
    """:382-419"""
    change_wheel_friction = EventTerm(
        func=mdp.randomize_rigid_body_material, mode="startup",
        params=dict(static_friction_range=(2.0, 2.0), dynamic_friction_range=(1.0, 1.0), restitution_range=(0.0, 0.0),
                    num_buckets=5, asset_cfg=SceneEntityCfg("robot", body_names=".*wheel_.*link")))
    add_base_mass = EventTerm(
        func=mdp.randomize_rigid_body_mass, mode="startup",
        params=dict(asset_cfg=SceneEntityCfg("robot", body_names=["base_link"]), mass_distribution_params=(0.2, 0.5),
                    operation="add"))
    set_goal = EventTerm(
        func=mdp.reset_root_state_uniform, mode="reset",
        params=dict(pose_range={"x": (-19.0, 19.0), "y": (-19.0, 19.0), "yaw": (-3.14, 3.14)},
                    velocity_range={"x": (0.1, 0.2), "y": (0.1, 0.2)}))


@configclass
class ElevationCommandCfg:
    """:421-435"""
    goal_pose = UniformPose2dCommandCfg(
        asset_name="robot", simple_heading=True, resampling_time_range=(10.0, 10.0), debug_vis=True,
        ranges=UniformPose2dCommandCfg.Ranges(pos_x=(-19.0, 19.0), pos_y=(-19.0, 19.0), heading=(-3.14, 3.14)))


@configclass
class MushrElevationRLEnvCfg(ManagerBasedRLEnvCfg):
    """:437-469"""
    wl_task = "elevation"
    seed: int = 42
    num_envs: int = 512
    env_spacing: float = 0.0
    observations: ElevationObsCfg = ElevationObsCfg()
    actions: Mushr4WDActionCfg = Mushr4WDActionCfg()
    events: ElevationSceneEventsCfg = ElevationSceneEventsCfg()
    curriculum: ElevationCurriculumCfg = ElevationCurriculumCfg()
    rewards: ElevationRewardsCfg = ElevationRewardsCfg()
    terminations: ElevationTerminationsCfg = ElevationTerminationsCfg()
    commands: ElevationCommandCfg = ElevationCommandCfg()

    def __post_init__(self):
        self.viewer.eye, self.viewer.lookat = [20.0, -20.0, 20.0], [0.0, 0.0, 0.0]
        self.sim.dt = 0.01        # 100 Hz physics
        self.decimation = 10      # 10 Hz control
        self.actions.throttle_steer.scale = (3.0, 0.488)
        self.sim.render_interval = self.decimation
        self.episode_length_s = 20
        self.scene = ElevationSceneCfg(num_envs=self.num_envs, env_spacing=self.env_spacing)


@configclass
class MushrElevationPlayEnvCfg(MushrElevationRLEnvCfg):
    """no terminations (:472-475)"""
    terminations = None
