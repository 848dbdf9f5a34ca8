from .f1tenth_drift_env_cfg import F1TenthDriftRLEnvCfg  # noqa: F401
from .mushr_drift_env_cfg import MushrDriftPlayEnvCfg, MushrDriftRLEnvCfg  # noqa: F401
