#!/usr/bin/env python3
"""Train a registered run config with PPO on one MI355X -- the counterpart of the reference's
source/wheeledlab_rl/scripts/train_rl.py (run config :8,:34, env creation :70-93, runner :95, checkpoints :96-106,
learn :114), with the same command line:

    python scripts/train_rl.py -r RSS_DRIFT_CONFIG env_setup.num_envs=4096 train.num_iterations=150 \\
        env.rewards.side_slip.weight=20 agent.algorithm.learning_rate=3e-4 train.log.run_name=drift-a

`-r` names a run config (wheeledlab_amd/configs/runs: RSS_DRIFT_CONFIG, RSS_ELEV_CONFIG, RSS_VISUAL_CONFIG,
F1TENTH_DRIFT_CONFIG); the remaining arguments are Hydra-style `key=value` overrides of `env_setup`, `train`, `env`
(the task's env cfg) and `agent` (its rsl_rl cfg).  For the drift tasks the rollout of every iteration is one fused launch
(actor MLP on the matrix pipe + env.step, csrc/wl_policy.hip) and the PPO update runs in csrc/wl_ppo.hip;
`--stepwise` forces the generic one-launch-per-env.step() collector (the only one for the elevation / visual tasks,
whose observations are 689 / 3208 wide)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def checkpoint_path(logs_dir: str, load_run: str, load_run_checkpoint: int = 0) -> str:
    """train_rl.py:99-106 of the reference (IsaacLab get_checkpoint_path): `load_run` names a run directory under the logs
    directory (a regex, the last match in sorted order wins), the checkpoint is models/model_<load_run_checkpoint>.pt or,
    for 0, the highest-numbered model_*.pt; a path to a checkpoint file is taken as it is."""
    import re
    if os.path.isfile(load_run):
        return load_run
    runs = sorted(d for d in (os.listdir(logs_dir) if os.path.isdir(logs_dir) else []) if re.fullmatch(load_run, d)
                  and os.path.isdir(os.path.join(logs_dir, d)))
    if not runs:
        raise FileNotFoundError(f"no run matching '{load_run}' under {logs_dir}")
    models = os.path.join(logs_dir, runs[-1], "models")
    number = str(load_run_checkpoint) if load_run_checkpoint > 0 else r"\d+"
    pat = re.compile(r"model_(" + number + r")\.pt")
    files = [(int(m.group(1)), f) for f in (os.listdir(models) if os.path.isdir(models) else []) if (m := pat.fullmatch(f))]
    if not files:
        raise FileNotFoundError(f"no checkpoint matching '{pat.pattern}' in {models}")
    return os.path.join(models, max(files)[1])


def main():
    ap = argparse.ArgumentParser(description="Train an RL agent on the MI355X-native WheeledLab envs.")
    ap.add_argument("-r", "--run-config-name", default="RSS_DRIFT_CONFIG")
    ap.add_argument("--stepwise", action="store_true", help="one launch per env.step() with a torch actor")
    ap.add_argument("--torch-policy", action="store_true", help="per-step collection with the policy step in torch eager")
    ap.add_argument("--torch-learner", action="store_true", help="PPO.update in torch (autograd + Adam) instead of the HIP learner")
    ap.add_argument("--history-out", default=None, help="write the per-iteration history (JSON) here")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("overrides", nargs="*", help="Hydra-style key=value overrides")
    args = ap.parse_args()

    import torch
    import yaml

    from wheeledlab_amd import registry
    from wheeledlab_amd.configs.runs import resolve_run
    from wheeledlab_amd.rl import ClipAction, RslRlVecEnvWrapper
    from wheeledlab_amd.rl.ppo import OnPolicyRunner

    from wheeledlab_amd import dist as D

    run_cfg = resolve_run(args.run_config_name, args.overrides)
    env_cfg, agent_cfg, train_cfg, log_cfg, env_setup = run_cfg.env, run_cfg.agent, run_cfg.train, run_cfg.train.log, run_cfg.env_setup
    # one process per GPU (python -m torch.distributed.run --nproc-per-node N scripts/train_rl.py ...): every rank trains on
    # its own num_envs envs (global ids rank * num_envs ...), gradients are averaged per minibatch step, rank 0 logs
    rank, local_rank, world = D.init_from_env()
    if world > 1:
        dev = f"cuda:{local_rank}" if torch.cuda.is_available() else "cpu"
        env_cfg.sim.device, train_cfg.device = dev, dev
    log_dir = None if (log_cfg.no_log or rank != 0) else log_cfg.run_log_dir
    if log_dir:
        os.makedirs(log_cfg.model_save_path, exist_ok=True)
        with open(os.path.join(log_dir, "run_config.yaml"), "w") as f:     # train_rl.py:62-64
            yaml.safe_dump(json.loads(json.dumps(run_cfg.to_dict(), default=str)), f)

    torch.manual_seed(train_cfg.seed)
    env = registry.make(env_setup.task_name, cfg=env_cfg)
    env.action_space.low, env.action_space.high = -1.0, 1.0
    env = RslRlVecEnvWrapper(ClipAction(env))
    if not log_cfg.no_checkpoints:
        agent_cfg.save_interval = min(agent_cfg.save_interval, log_cfg.checkpoint_every)
    if args.torch_learner:
        agent_cfg.algorithm.fused_update = False
    runner = OnPolicyRunner(env, agent_cfg, log_dir=None if log_cfg.no_checkpoints else log_dir, device=train_cfg.device,
                            fused=False if args.stepwise else None, kernel_policy=False if args.torch_policy else None)
    if world > 1:
        # the parameters above were initialised from the SAME seed on every rank (they must start identical); whatever is
        # sampled from torch's global RNG from here on (the torch policy path's exploration noise) must differ per shard
        torch.manual_seed(train_cfg.seed + 1000003 * rank)
    if train_cfg.load_run is not None:
        resume_path = checkpoint_path(log_cfg.logs_dir, train_cfg.load_run, train_cfg.load_run_checkpoint)
        if rank == 0 and not args.quiet:
            print(f"[INFO]: Loading model checkpoint from: {resume_path}")
        runner.load(resume_path)
    env.seed(agent_cfg.seed)
    env.unwrapped.common_step_counter = train_cfg.set_env_step       # for continuing curriculums (train_rl.py:113)
    hist = runner.learn(train_cfg.num_iterations, verbose=not args.quiet)
    if log_dir:
        with open(os.path.join(log_dir, "history.json"), "w") as f:
            json.dump(hist, f)
    if args.history_out and rank == 0:
        with open(args.history_out, "w") as f:
            json.dump(hist, f)
    first, last = hist[0], hist[-1]
    # replicated state must not have drifted apart (identical averaged gradients -> identical steps on every rank)
    flat_params = torch.cat([p.detach().reshape(-1) for p in runner.actor_critic.parameters()])
    in_sync = D.ranks_agree(flat_params)
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        print(json.dumps({"run_config": args.run_config_name, "task": env_setup.task_name, "num_envs": env_setup.num_envs,
                          "fused_collection": runner.fused, "fused_learner": runner.alg.fused_update, "kernel_policy": runner.kernel_policy, "iterations": len(hist),
                          "mean_step_reward_first": first["mean_step_reward"], "mean_step_reward_last": last["mean_step_reward"],
                          "mean_reward_last": last["mean_reward"], "mean_episode_length_last": last["mean_episode_length"],
                          "n_gpus": world, "ranks_in_sync": in_sync, "fps_last": last["fps"], "collection_fps_last": last["collection_fps"], "log_dir": log_dir}))
    env.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
