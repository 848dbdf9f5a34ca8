# Profiling pass (run through gpurun): per-task HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters for the
# drift / elevation / visual step kernels, the depth ray-cast and the visual-depth task at the BASELINE size and large N, the
# calibration of the traffic counters on known bytes, and kernel statistics of bench.py itself (the driver's command).
# Every rocprofv3 --pmc run is its own pass (no trace domains next to --pmc) and runs under `timeout`.
#   usage: tools/profile_pmc.sh <tag>      (output gpurun_out/<tag>/, digest <tag>_pmc.json + <tag>_bench_kernel_stats.csv there)
#   WL_PMC_WORK="task:envs:K ..." overrides the workload list, WL_FLAGS forces a kernel form (WlEnvBuffers.flags), WL_PMC_NO_STATS=1
#   skips the kernel statistics of bench.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:?usage: profile_pmc.sh <tag>}
O=$R/gpurun_out/$TAG
mkdir -p $O
pm() { d=$1; shift; c=$1; shift; timeout ${WL_PMC_TIMEOUT:-240} rocprofv3 --output-format csv --pmc $c -d $O/$d -- "$@" > $O/$d.log 2>&1; }
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
SQB="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
WORK=${WL_PMC_WORK:-"drift:4096:64 drift:65536:16 drift:1048576:8 drift:4194304:4 elev:4096:32 elev:262144:4 visual:4096:16 visual:262144:2 depth:4096:8 visual_depth:4096:8"}
for w in $WORK; do
  IFS=: read task n k <<< "$w"
  pm FETCH_${task}_$n FETCH_SIZE python $R/tools/pmc_run.py $task $n $k
  pm WRITE_${task}_$n WRITE_SIZE python $R/tools/pmc_run.py $task $n $k
  pm sqa_${task}_$n "$SQA" python $R/tools/pmc_run.py $task $n $k
  pm sqb_${task}_$n "$SQB" python $R/tools/pmc_run.py $task $n $k
done
# calibration of the two traffic counters on known bytes (built here: the binary is not in the tree)
if timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/microbench/pmc_calib.hip -o $O/pmc_calib > $O/pmc_calib_build.log 2>&1; then
  pm calib_FETCH FETCH_SIZE $O/pmc_calib
  pm calib_WRITE WRITE_SIZE $O/pmc_calib
fi
if [ -z "$WL_PMC_NO_STATS" ]; then
  # the SAME command the driver runs (sweep children and CPU baseline included: their kernels land in their own process's summary)
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $O/bench_under_rocprof_detail.json > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
fi
cd $R && python tools/pmc_report.py $O $TAG
