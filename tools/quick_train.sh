# short A/B after a learner / collector change: the wide-learner tests + 8-iteration runs of the elevation and visual configs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ppo_wide.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > $O/tests.txt
for cfg in RSS_ELEV_CONFIG:4096 RSS_VISUAL_CONFIG:1024; do
  timeout 200 python scripts/train_rl.py -r ${cfg%%:*} env_setup.num_envs=${cfg##*:} train.num_iterations=8 train.log.no_log=true --quiet 2>/dev/null | tail -1 > $O/train_${cfg%%:*}.json
done
