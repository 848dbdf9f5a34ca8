"""Visual-DEPTH task -- an EXTENSION, not a reference id (BASELINE.json configs[4]: "Visual task, 4096 envs, depth raycast against
heightfield").  The reference defines the observation functions `camera_data_depth` / `raycast_depth`
(wheeledlab_tasks/visual/mdp_sensors/observations.py:89-95) and a camera with a `distance_to_image_plane` capable spawn
(visual/mushr_visual_env_cfg.py:230-246) but never wires them into an observation group; its visual task drives on a flat
black/white plane.  This config wires them: the visual task's actions / rewards / terminations / reset (same classes as
tasks/visual), a HEIGHTFIELD terrain under the traversability map, and `raycast_depth` as the policy's image term --
observation [4808] = distance_to_image_plane 60 x 80 | base_lin_vel | base_ang_vel | last_action."""
from ...envs import mdp
from ...envs.configclass import configclass
from ...envs.managers_cfg import AdditiveUniformNoiseCfg as Unoise
from ...envs.managers_cfg import ObservationGroupCfg as ObsGroup
from ...envs.managers_cfg import ObservationTermCfg as ObsTerm
from ...envs.managers_cfg import SceneEntityCfg
from ...envs.sensors_cfg import PinholeCameraCfg, TiledCameraCfg
from ..visual.mushr_visual_env_cfg import MushrVisualRLEnvCfg, MushrVisualSceneCfg, VisualTerrainImporterCfg


@configclass
class VisualDepthObsCfg:
    """camera depth (4800) | v_b (3) | w_b (3) | last action (2); corruption off, as the visual task's group (:37-58)"""

    @configclass
    class PolicyCfg(ObsGroup):
        depth = ObsTerm(func=mdp.raycast_depth, params=dict(sensor_cfg=SceneEntityCfg("camera")))
        base_lin_vel = ObsTerm(func=mdp.base_lin_vel, noise=Unoise(n_min=-0.1, n_max=0.1))
        base_ang_vel = ObsTerm(func=mdp.base_ang_vel, noise=Unoise(n_min=-0.1, n_max=0.1))
        last_action = ObsTerm(func=mdp.last_action, clip=(-1.0, 1.0), noise=Unoise(n_min=-0.1, n_max=0.1))

        def __post_init__(self):
            self.enable_corruption = False
            self.concatenate_terms = True

    policy: PolicyCfg = PolicyCfg()


@configclass
class VisualDepthTerrainCfg(VisualTerrainImporterCfg):
    """the traversability map over a heightfield: 80 x 80 cells of 0.5 m = the 40 m square of the synthetic 800 x 800 field at
    0.05 m (wheeledlab_amd/terrain.py); `heightfield`: (height [ny, nx], x0, y0, cell) or None = that field"""
    terrain_type = "traversability_heightfield"
    num_rows, num_cols = 80, 80
    env_num_rows, env_num_cols = 40, 40
    group_num_rows, group_num_cols = 20, 20
    heightfield = None


@configclass
class MushrVisualDepthSceneCfg(MushrVisualSceneCfg):
    terrain = VisualDepthTerrainCfg()
    camera = TiledCameraCfg(prim_path="{ENV_REGEX_NS}/Robot/mushr_nano/camera_link/camera", update_period=0.1, height=60,
                            width=80, data_types=["distance_to_image_plane"], offset_pos=(0.08, 0.0, 0.0), offset_convention="ros",
                            spawn=PinholeCameraCfg(focal_length=1.9299999475479126, horizontal_aperture=3.8959999084472656,
                                                   vertical_aperture=2.453000068664551, clipping_range=(0.01, 20.0)))


@configclass
class MushrVisualDepthRLEnvCfg(MushrVisualRLEnvCfg):
    wl_task = "visual_depth"
    observations: VisualDepthObsCfg = VisualDepthObsCfg()
    augment_camera: bool = False

    def __post_init__(self):
        super().__post_init__()
        self.scene = MushrVisualDepthSceneCfg(num_envs=self.num_envs, env_spacing=self.env_spacing)


@configclass
class MushrVisualDepthPlayEnvCfg(MushrVisualDepthRLEnvCfg):
    rewards = None
    terminations = None
