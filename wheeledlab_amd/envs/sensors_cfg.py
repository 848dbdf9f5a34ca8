"""Sensor / command config classes named as in `isaaclab.sensors` / `isaaclab.envs.mdp.commands` (values only)."""
from .configclass import MISSING, configclass


@configclass
class GridPatternCfg:
    size: tuple = MISSING
    resolution: float = MISSING
    direction: tuple = (0.0, 0.0, -1.0)


@configclass
class RayCasterCfg:
    prim_path: str = ""
    offset_pos: tuple = (0.0, 0.0, 0.0)
    attach_yaw_only: bool = True
    pattern_cfg: GridPatternCfg = GridPatternCfg(size=(1.0, 1.0), resolution=0.1)
    mesh_prim_paths: list = []
    debug_vis: bool = False


@configclass
class PinholeCameraCfg:
    focal_length: float = 24.0
    horizontal_aperture: float = 20.955
    vertical_aperture: float = 15.29
    clipping_range: tuple = (0.01, 1e2)


@configclass
class TiledCameraCfg:
    prim_path: str = ""
    update_period: float = 0.0
    height: int = 60
    width: int = 80
    data_types: list = ["rgb"]
    spawn: PinholeCameraCfg = PinholeCameraCfg()
    offset_pos: tuple = (0.0, 0.0, 0.0)
    offset_rot: tuple = (1.0, 0.0, 0.0, 0.0)
    offset_convention: str = "ros"
    body_pos: tuple = (0.23, 0.0, 0.18)   # pose of the camera in the base frame (designed: camera_link is in the missing USD)
    # what a depth pixel beyond the far clipping plane reads: "max" (the far distance), "zero", or "none" (+inf) -- the switch of
    # IsaacLab's camera cfgs [IsaacLab-recalled; its own default is "none"]; "max" here keeps observation rows finite
    depth_clipping_behavior: str = "max"
    debug_vis: bool = False


@configclass
class UniformPose2dCommandRanges:
    pos_x: tuple = MISSING
    pos_y: tuple = MISSING
    heading: tuple = MISSING


@configclass
class UniformPose2dCommandCfg:
    asset_name: str = "robot"
    simple_heading: bool = True
    resampling_time_range: tuple = MISSING
    ranges: UniformPose2dCommandRanges = MISSING
    debug_vis: bool = False
    Ranges = UniformPose2dCommandRanges
