from .mushr_visual_depth_env_cfg import MushrVisualDepthPlayEnvCfg, MushrVisualDepthRLEnvCfg  # noqa: F401
