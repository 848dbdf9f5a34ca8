# The HIP learner (split-bf16 first layer and 64 x 64 products) against the torch learner (f32 autograd + Adam) on the same
# task, seed and collection path: per-iteration histories -> gpurun_out/learner_compare/*.json
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/learner_compare
mkdir -p $O
for cfg in RSS_ELEV_CONFIG:4096:40 RSS_DRIFT_CONFIG:4096:60; do
  IFS=: read run n it <<< "$cfg"
  python $R/scripts/train_rl.py -r $run env_setup.num_envs=$n train.num_iterations=$it train.log.no_log=true --quiet --history-out $O/${run}_hip.json > /dev/null 2>&1
  python $R/scripts/train_rl.py -r $run env_setup.num_envs=$n train.num_iterations=$it train.log.no_log=true --quiet --torch-learner --history-out $O/${run}_torch.json > /dev/null 2>&1
done
