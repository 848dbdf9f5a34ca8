#!/bin/bash
# kernel statistics of the wide PPO step (elevation / visual agent sizes): rocprofv3 --kernel-trace --stats of tools/ppo_wide_probe.py
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for cfg in "689 524288" "3208 131072"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_wide_$1 -o wide -- python $GRAFT_REPO_ROOT/tools/ppo_wide_probe.py $1 $2 > $OUT/prof_wide_$1.log 2>&1
  f=$(find $OUT/prof_wide_$1 -name '*kernel_stats.csv' | head -1)
  cp "$f" $OUT/wide_$1_kernel_stats.csv
done
