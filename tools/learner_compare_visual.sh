# tools/learner_compare.sh for the visual agent (3208 inputs, 1024 envs x 40 iterations), default seed and seed 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/learner_compare
mkdir -p $O
for seed in default 1; do
  S=""; [ $seed != default ] && S="train.seed=$seed agent.seed=$seed"
  timeout 300 python $R/scripts/train_rl.py -r RSS_VISUAL_CONFIG env_setup.num_envs=1024 train.num_iterations=40 train.log.no_log=true $S --quiet --history-out $O/visual_${seed}_hip.json > /dev/null 2>&1
  timeout 600 python $R/scripts/train_rl.py -r RSS_VISUAL_CONFIG env_setup.num_envs=1024 train.num_iterations=40 train.log.no_log=true $S --quiet --torch-learner --history-out $O/visual_${seed}_torch.json > /dev/null 2>&1
done
