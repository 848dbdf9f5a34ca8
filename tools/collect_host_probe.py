import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import wheeledlab_amd.tasks
from wheeledlab_amd import registry
from wheeledlab_amd.policy import RolloutStorage
from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper
from wheeledlab_amd.rl.ppo import ActorCritic
DEV="cuda:0"
for task in ("Isaac-MushrElevationRL-v0", "Isaac-MushrDriftRL-v0"):
    cfg = registry.parse_env_cfg(task, device=DEV, num_envs=64)
    env = registry.make(task, cfg=cfg); env.action_space.low, env.action_space.high = -1.0, 1.0
    w = RslRlVecEnvWrapper(ClipAction(env))
    D = w.num_obs
    ac = ActorCritic(D, D, 2, activation="relu").to(DEV); view = ac.fused()
    K = 128
    st = RolloutStorage(K, 64, D, 2, DEV)
    obs,_ = w.get_observations(); st.observations[0].copy_(obs)
    base = w.unwrapped
    with torch.inference_mode():
        for rep in range(3):
            for k in range(K): base.collect_step(view, st, k)
        torch.cuda.synchronize()
        t0=time.perf_counter()
        for rep in range(5):
            for k in range(K): base.collect_step(view, st, k)
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        print(task, "host us/step", round((t1-t0)/(5*K)*1e6,1), "incl. drain", round((t2-t0)/(5*K)*1e6,1))
        pr=cProfile.Profile(); pr.enable()
        for k in range(K): base.collect_step(view, st, k)
        pr.disable(); s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:2600])
