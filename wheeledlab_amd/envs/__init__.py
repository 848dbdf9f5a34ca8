from .configclass import MISSING, configclass  # noqa: F401
from .manager_based_rl_env import ManagerBasedRLEnv  # noqa: F401
from .managers_cfg import ManagerBasedRLEnvCfg  # noqa: F401
