from .mushr_visual_env_cfg import MushrVisualPlayEnvCfg, MushrVisualRLEnvCfg, MushrVisualRLRandomEnvCfg  # noqa: F401
