"""Task registrations -- same ids and kwargs keys as wheeledlab_tasks/__init__.py:14-63, entry point = this
package's ManagerBasedRLEnv."""
from ..registry import register
from .drifting import MushrDriftPlayEnvCfg, MushrDriftRLEnvCfg

_ENV = "wheeledlab_amd.envs:ManagerBasedRLEnv"

register(
    id="Isaac-MushrDriftRL-v0",
    entry_point=_ENV,
    disable_env_checker=True,
    kwargs={
        "env_cfg_entry_point": MushrDriftRLEnvCfg,
        "rsl_rl_cfg_entry_point": "wheeledlab_amd.tasks.drifting.agents:MushrPPORunnerCfg",
        "play_env_cfg_entry_point": MushrDriftPlayEnvCfg,
    },
)
