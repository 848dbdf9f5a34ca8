# End-of-round GPU pass (round 2): GPU tests, smoke, bench (plain and under rocprofv3 --stats), wide-learner kernel statistics,
# short end-to-end training runs of the three RSS configs.  Outputs under gpurun_out/final/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > $O/gpu_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-sweep > $O/bench_s20.json 2>> $O/bench.err
for cfg in RSS_DRIFT_CONFIG:4096 RSS_ELEV_CONFIG:4096 RSS_VISUAL_CONFIG:1024; do
  timeout 300 python scripts/train_rl.py -r ${cfg%%:*} env_setup.num_envs=${cfg##*:} train.num_iterations=8 train.log.no_log=true --quiet 2>/dev/null | tail -1 > $O/train_${cfg%%:*}.json
done
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python $R/bench.py --no-sweep --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/bench_stats -name '*kernel_stats.csv' | head -1) $O/bench_kernel_stats.csv
cd $R && bash tools/r02_wide_profile.sh
cp $R/gpurun_out/wide_689_kernel_stats.csv $R/gpurun_out/wide_3208_kernel_stats.csv $O/
