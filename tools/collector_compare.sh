# elevation, 4096 envs x 40 iterations: the persistent collector (default) against per-step collection, two seeds
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/learner_compare
mkdir -p $O
for seed in default 1; do
  S=""; [ $seed != default ] && S="train.seed=$seed agent.seed=$seed"
  timeout 300 python $R/scripts/train_rl.py -r RSS_ELEV_CONFIG env_setup.num_envs=4096 train.num_iterations=40 train.log.no_log=true $S --quiet --history-out $O/elev_${seed}_persistent.json > /dev/null 2>&1
  WL_ELEV_PERSISTENT_COLLECT=0 timeout 300 python $R/scripts/train_rl.py -r RSS_ELEV_CONFIG env_setup.num_envs=4096 train.num_iterations=40 train.log.no_log=true $S --quiet --history-out $O/elev_${seed}_stepwise.json > /dev/null 2>&1
done
