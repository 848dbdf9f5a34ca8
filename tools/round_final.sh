# The end-of-round GPU pass, ONCE per round (through gpurun): the -m gpu suite + smoke, bench.py twice (the driver's K = 20 command
# and the default K), the rocprofv3 kernel statistics of the driver's command + the PMC passes (tools/profile_pmc.sh), 8-iteration
# runs of the five run configs.     usage: bash tools/round_final.sh <tag>      outputs: gpurun_out/<tag>/  (then tools/round_collect.sh)
cd $GRAFT_REPO_ROOT
TAG=${1:?usage: round_final.sh <tag>}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_s20_detail.json > $O/bench_s20.json 2> $O/bench_s20.err; tail -c 300 $O/bench_s20.err
timeout 600 python bench.py --detail-out $O/bench_detail.json > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
for run in RSS_DRIFT_CONFIG:4096 RSS_ELEV_CONFIG:4096 RSS_VISUAL_CONFIG:1024 F1TENTH_DRIFT_CONFIG:4096 VISUAL_DEPTH_CONFIG:512; do
  r=${run%%:*}; n=${run#*:}
  timeout 300 python scripts/train_rl.py -r $r env_setup.num_envs=$n train.num_iterations=8 train.log.no_log=true --quiet --history-out $O/train_${r}.json > $O/train_${r}.log 2>&1
  echo "$r rc $?"
done
bash tools/profile_pmc.sh $TAG > $O/profile_pmc.log 2>&1; tail -c 600 $O/profile_pmc.log
tail -c 3000 $O/bench_s20.json
