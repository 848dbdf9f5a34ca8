#!/usr/bin/env python3
"""Train a registered task with PPO on one MI355X -- the counterpart of the reference's
source/wheeledlab_rl/scripts/train_rl.py (env creation :70-93, runner :95, checkpoints :96-106, learn :114).

    python scripts/train_rl.py --task Isaac-MushrDriftRL-v0 --num_envs 4096 --max_iterations 100 --log_dir logs/drift

For the drift task the rollout of every iteration is one fused launch (actor MLP on the matrix pipe + env.step, see
wheeledlab_amd/csrc/wl_policy.hip); `--stepwise` forces the generic one-launch-per-env.step() path (the only path for
the elevation / visual tasks, whose observations are 689 / 3208 wide)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="Isaac-MushrDriftRL-v0")
    ap.add_argument("--num_envs", type=int, default=4096)
    ap.add_argument("--max_iterations", type=int, default=100)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--log_dir", default=None)
    ap.add_argument("--load_run", default=None, help="checkpoint (model_*.pt) to resume from")
    ap.add_argument("--set_env_step", type=int, default=0, help="common_step_counter to continue curriculums from")
    ap.add_argument("--stepwise", action="store_true", help="one launch per env.step() with a torch actor")
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args()

    import torch

    import wheeledlab_amd.tasks  # noqa: F401  (registers the task ids)
    from wheeledlab_amd import registry
    from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper
    from wheeledlab_amd.rl.ppo import OnPolicyRunner

    torch.manual_seed(args.seed)
    env_cfg = registry.parse_env_cfg(args.task, device=args.device, num_envs=args.num_envs)
    env_cfg.seed = args.seed
    agent_cfg = registry.load_cfg_from_registry(args.task, "rsl_rl_cfg_entry_point")
    env = registry.make(args.task, cfg=env_cfg)
    env.action_space.low, env.action_space.high = -1.0, 1.0
    env = RslRlVecEnvWrapper(ClipAction(env))
    runner = OnPolicyRunner(env, agent_cfg, log_dir=args.log_dir, device=args.device, fused=False if args.stepwise else None)
    if args.load_run:
        runner.load(args.load_run)
    env.seed(args.seed)
    env.unwrapped.common_step_counter = args.set_env_step
    hist = runner.learn(args.max_iterations, verbose=not args.quiet)
    if args.log_dir:
        with open(os.path.join(args.log_dir, "history.json"), "w") as f:
            json.dump(hist, f)
    first, last = hist[0], hist[-1]
    print(json.dumps({"task": args.task, "fused_collection": runner.fused, "iterations": len(hist),
                      "mean_step_reward_first": first["mean_step_reward"], "mean_step_reward_last": last["mean_step_reward"],
                      "mean_reward_last": last["mean_reward"], "mean_episode_length_last": last["mean_episode_length"],
                      "fps_last": last["fps"], "collection_fps_last": last["collection_fps"]}))
    env.close()


if __name__ == "__main__":
    main()
