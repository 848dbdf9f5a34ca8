from .mushr_drift_env_cfg import MushrDriftPlayEnvCfg, MushrDriftRLEnvCfg  # noqa: F401
