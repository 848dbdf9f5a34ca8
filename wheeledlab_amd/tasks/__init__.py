"""Task registrations -- same ids and kwargs keys as wheeledlab_tasks/__init__.py:14-63, entry point = this
package's ManagerBasedRLEnv."""
from ..registry import register
from .drifting import F1TenthDriftRLEnvCfg, MushrDriftPlayEnvCfg, MushrDriftRLEnvCfg
from .elevation import MushrElevationPlayEnvCfg, MushrElevationRLEnvCfg
from .visual import MushrVisualPlayEnvCfg, MushrVisualRLEnvCfg
from .visual_depth import MushrVisualDepthPlayEnvCfg, MushrVisualDepthRLEnvCfg

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

register(
    id="Isaac-MushrVisualRL-v0", entry_point=_ENV, disable_env_checker=True,
    kwargs={"env_cfg_entry_point": MushrVisualRLEnvCfg,
            "rsl_rl_cfg_entry_point": "wheeledlab_amd.tasks.visual.agents:MushrPPORunnerCfg",
            "play_env_cfg_entry_point": MushrVisualPlayEnvCfg},
)
register(
    id="Isaac-MushrElevationRL-v0", entry_point=_ENV, disable_env_checker=True,
    kwargs={"env_cfg_entry_point": MushrElevationRLEnvCfg,
            "rsl_rl_cfg_entry_point": "wheeledlab_amd.tasks.elevation.agents:MushrPPORunnerCfg",
            "play_env_cfg_entry_point": MushrElevationPlayEnvCfg},
)
register(
    id="Isaac-F1TenthDriftRL-v0", entry_point=_ENV, disable_env_checker=True,
    kwargs={"env_cfg_entry_point": F1TenthDriftRLEnvCfg,
            "rsl_rl_cfg_entry_point": "wheeledlab_amd.tasks.drifting.agents:F1TenthPPORunnerCfg"},
)
# EXTENSION id (not in the reference): the visual task on a heightfield with the depth image as observation -- BASELINE.json configs[4]
register(
    id="Isaac-MushrVisualDepthRL-v0", entry_point=_ENV, disable_env_checker=True,
    kwargs={"env_cfg_entry_point": MushrVisualDepthRLEnvCfg,
            "rsl_rl_cfg_entry_point": "wheeledlab_amd.tasks.visual.agents:MushrPPORunnerCfg",
            "play_env_cfg_entry_point": MushrVisualDepthPlayEnvCfg},
)
