from .mushr_elevation_env_cfg import MushrElevationPlayEnvCfg, MushrElevationRLEnvCfg  # noqa: F401
