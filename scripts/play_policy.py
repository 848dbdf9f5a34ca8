#!/usr/bin/env python3
"""Roll out a trained checkpoint and (optionally) save the rollout in the reference's format
(source/wheeledlab_rl/scripts/play_policy.py:128-165: torch.save({'observations': [T,N,D], 'actions': [T,N,2]})).

    python scripts/play_policy.py --task Isaac-MushrDriftRL-v0 --checkpoint logs/drift/models/model_99.pt --steps 200 \\
        --save_data logs/drift/playback/run-rollouts.pt"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--task", default="Isaac-MushrDriftRL-v0")
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--num_envs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--save_data", default=None)
    args = ap.parse_args()

    import torch

    import wheeledlab_amd.tasks  # noqa: F401
    from wheeledlab_amd import registry
    from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper
    from wheeledlab_amd.rl.ppo import OnPolicyRunner

    env_cfg = registry.parse_env_cfg(args.task, device=args.device, num_envs=args.num_envs, play=True)
    agent_cfg = registry.load_cfg_from_registry(args.task, "rsl_rl_cfg_entry_point")
    env = registry.make(args.task, cfg=env_cfg)
    env.action_space.low, env.action_space.high = -1.0, 1.0
    env = RslRlVecEnvWrapper(ClipAction(env))
    runner = OnPolicyRunner(env, agent_cfg, device=args.device)
    runner.load(args.checkpoint, load_optimizer=False)
    policy = runner.get_inference_policy(device=env.unwrapped.device)
    data = {"observations": [], "actions": []}
    obs, _ = env.get_observations()
    total, speed, yaw_rate = 0.0, 0.0, 0.0
    for _ in range(args.steps):
        with torch.inference_mode():
            actions = policy(obs)
            obs, rew, _, _ = env.step(actions)
        total += float(rew.mean())
        if obs.shape[1] == 14:           # drift observation: body-frame velocity at [6:9], angular velocity at [9:12]
            speed += float(obs[:, 6:8].norm(dim=-1).mean())
            yaw_rate += float(obs[:, 11].mean())
        data["observations"].append(obs.clone())
        data["actions"].append(actions.clone())
    # the play cfgs carry no reward terms (as the reference's: mushr_drift_env_cfg.py:425-427), so the reward is 0 there
    print(f"{args.steps} steps x {args.num_envs} envs: mean reward / step {total / args.steps:.4f}, "
          f"mean planar speed {speed / args.steps:.3f} m/s, mean yaw rate {yaw_rate / args.steps:.3f} rad/s")
    if args.save_data:
        os.makedirs(os.path.dirname(os.path.abspath(args.save_data)), exist_ok=True)
        torch.save({k: torch.stack(v, 0) for k, v in data.items()}, args.save_data)
        print("[INFO] Saved episode data to:", args.save_data)
    env.close()


if __name__ == "__main__":
    main()
