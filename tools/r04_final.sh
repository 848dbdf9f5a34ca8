# round 4: bench lines + short training runs (through gpurun)     outputs: gpurun_out/r04f/
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err
for run in RSS_DRIFT_CONFIG:4096 RSS_ELEV_CONFIG:4096 RSS_VISUAL_CONFIG:1024 F1TENTH_DRIFT_CONFIG:4096 VISUAL_DEPTH_CONFIG:512; do
  r=${run%%:*}; n=${run#*:}
  timeout 300 python scripts/train_rl.py -r $r env_setup.num_envs=$n train.num_iterations=8 train.log.no_log=true --quiet --history-out $O/train_${r}.json > $O/train_${r}.log 2>&1
  echo "$r rc $?"; tail -2 $O/train_${r}.log | cut -c1-300
done
head -c 1500 $O/bench.json
