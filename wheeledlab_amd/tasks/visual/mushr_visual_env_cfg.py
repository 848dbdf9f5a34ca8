"""Visual task configuration -- same config surface as the reference's
wheeledlab_tasks/visual/mushr_visual_env_cfg.py (citations inline).  Unlike the reference, importing this module has
no side effects: the traversability map is generated when the env is constructed, from
`scene.terrain` parameters, not at class-definition time into a USD file."""
from ...assets import MUSHR_SUS_CFG
from ...envs import mdp
from ...envs.configclass import configclass
from ...envs.managers_cfg import AdditiveUniformNoiseCfg as Unoise
from ...envs.managers_cfg import EventTermCfg as EventTerm
from ...envs.managers_cfg import (InteractiveSceneCfg, ManagerBasedRLEnvCfg, RigidBodyMaterialCfg, SceneEntityCfg,
                                  TerrainImporterCfg)
from ...envs.managers_cfg import ObservationGroupCfg as ObsGroup
from ...envs.managers_cfg import ObservationTermCfg as ObsTerm
from ...envs.managers_cfg import RewardTermCfg as RewTerm
from ...envs.managers_cfg import TerminationTermCfg as DoneTerm
from ...envs.sensors_cfg import PinholeCameraCfg, TiledCameraCfg
from ..common import Mushr4WDActionCfg


@configclass
class VisualObsCfg:
    """3208-dim observation (:37-58): camera(3200) | v_b(3) | w_b(3) | last action(2); noise declared, corruption off"""

    @configclass
    class PolicyCfg(ObsGroup):
        camera = ObsTerm(func=mdp.camera_data_rgb_flattened_aug, params=dict(sensor_cfg=SceneEntityCfg("camera")))
        base_lin_vel = ObsTerm(func=mdp.base_lin_vel, noise=Unoise(n_min=-0.1, n_max=0.1))
        base_ang_vel = ObsTerm(func=mdp.base_ang_vel, noise=Unoise(n_min=-0.1, n_max=0.1))
        last_action = ObsTerm(func=mdp.last_action, clip=(-1.0, 1.0), noise=Unoise(n_min=-0.1, n_max=0.1))

        def __post_init__(self):
            self.enable_corruption = False
            self.concatenate_terms = True

    policy: PolicyCfg = PolicyCfg()


@configclass
class VisualTerrainImporterCfg(TerrainImporterCfg):
    """procedural black/white traversability plane (:67-135)"""
    terrain_type = "traversability_plane"
    row_spacing, col_spacing = 0.5, 0.5
    num_rows, num_cols = 500, 500
    env_num_rows, env_num_cols = 100, 100
    group_num_rows, group_num_cols = 50, 50
    num_walkers = 1
    color_sampling = False
    traversability_hashmap = None      # bool [rows, cols]; None -> generated at env construction from cfg.seed
    physics_material = RigidBodyMaterialCfg(friction_combine_mode="multiply", restitution_combine_mode="multiply",
                                            static_friction=2.0, dynamic_friction=2.0)

    @property
    def width(self):
        return self.num_rows * self.row_spacing

    @property
    def height_m(self):
        return self.num_cols * self.col_spacing

    def get_map_id(self, x, y):
        """the cfg class's OWN cell lookup (reference :201-208) -- not the one the reward terms use
        (visual/utils/traversability_utils.py: round, [y, x] order): here the index is floor((x + width / 2 - spacing / 2) / spacing),
        clamped to the map, and the map is then indexed [x_idx, y_idx].  Torch tensors in, long tensors out."""
        import torch
        x_idx = torch.floor((x + self.width / 2 - self.row_spacing / 2) / self.row_spacing).long()
        y_idx = torch.floor((y + self.height_m / 2 - self.col_spacing / 2) / self.col_spacing).long()
        return torch.clamp(x_idx, 0, self.num_rows - 1), torch.clamp(y_idx, 0, self.num_cols - 1)

    def get_traversability(self, poses):
        """traversability of poses [N, >= 2] (x, y) through get_map_id (reference :190-196); needs `traversability_hashmap`"""
        import torch
        if self.traversability_hashmap is None:
            raise ValueError("traversability_hashmap is generated when the env is constructed; set it (or build the env) first")
        x_idx, y_idx = self.get_map_id(poses[:, 0], poses[:, 1])
        return torch.as_tensor(self.traversability_hashmap).to(x_idx.device)[x_idx, y_idx]


@configclass
class MushrVisualSceneCfg(InteractiveSceneCfg):
    """:210-252"""
    terrain = VisualTerrainImporterCfg()
    robot = MUSHR_SUS_CFG.replace(prim_path="{ENV_REGEX_NS}/Robot")
    camera = TiledCameraCfg(prim_path="{ENV_REGEX_NS}/Robot/mushr_nano/camera_link/camera", update_period=0.1, height=60,
                            width=80, data_types=["rgb"], offset_pos=(0.08, 0.0, 0.0), offset_convention="ros",
                            spawn=PinholeCameraCfg(focal_length=1.9299999475479126, horizontal_aperture=3.8959999084472656,
                                                   vertical_aperture=2.453000068664551, clipping_range=(0.01, 1e2)))


@configclass
class VisualEventsCfg:
    reset_root_state = EventTerm(func=mdp.reset_root_state, mode="reset")                      # :257-262


@configclass
class VisualEventsRandomCfg(VisualEventsCfg):
    """:264-299"""
    change_wheel_friction = EventTerm(
        func=mdp.randomize_rigid_body_material, mode="startup",
        params=dict(static_friction_range=(0.4, 0.6), dynamic_friction_range=(0.4, 0.6), restitution_range=(0.0, 0.0),
                    num_buckets=10, asset_cfg=SceneEntityCfg("robot", body_names=".*wheel_.*link"), make_consistent=False))
    add_base_mass = EventTerm(
        func=mdp.randomize_rigid_body_mass, mode="startup",
        params=dict(asset_cfg=SceneEntityCfg("robot", body_names=["base_link"]), mass_distribution_params=(1.0, 3.0),
                    operation="abs"))
    add_wheel_mass = EventTerm(
        func=mdp.randomize_rigid_body_mass, mode="startup",
        params=dict(asset_cfg=SceneEntityCfg("robot", body_names=".*wheel_.*link"), mass_distribution_params=(0.01, 0.3),
                    operation="abs"))


@configclass
class VisualRewardsCfg:
    """:373-385"""
    traversablility = RewTerm(func=mdp.traversable_reward, weight=5.0)
    vel_rew = RewTerm(func=mdp.forward_vel, weight=7.0)


@configclass
class VisualTerminationsCfg:
    """:405-409"""
    time_out = DoneTerm(func=mdp.time_out, time_out=True)
    out_range = DoneTerm(func=mdp.out_of_map)


@configclass
class MushrVisualRLEnvCfg(ManagerBasedRLEnvCfg):
    """:411-444"""
    wl_task = "visual"
    seed: int = 42
    num_envs: int = 1024
    env_spacing: float = 0.0
    events: VisualEventsCfg = VisualEventsCfg()
    actions: Mushr4WDActionCfg = Mushr4WDActionCfg()
    observations: VisualObsCfg = VisualObsCfg()
    rewards: VisualRewardsCfg = VisualRewardsCfg()
    terminations: VisualTerminationsCfg = VisualTerminationsCfg()
    augment_camera: bool = True        # sample (brightness, contrast, blur sigma) per step like torchvision's transforms

    def __post_init__(self):
        self.viewer.eye, self.viewer.lookat = [40.0, 0.0, 45.0], [0.0, 0.0, -3.0]
        self.sim.dt = 0.02
        self.decimation = 10
        self.episode_length_s = 10
        self.scene = MushrVisualSceneCfg(num_envs=self.num_envs, env_spacing=self.env_spacing)


@configclass
class MushrVisualRLRandomEnvCfg(MushrVisualRLEnvCfg):
    events: VisualEventsCfg = VisualEventsRandomCfg()


@configclass
class MushrVisualPlayEnvCfg(MushrVisualRLEnvCfg):
    """no rewards / terminations (:450-470)"""
    rewards = None
    terminations = None
