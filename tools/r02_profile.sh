# Round-2 profiling pass (run through gpurun): PMC traffic + its calibration on known bytes, SQ counters, kernel statistics.
# Every rocprofv3 --pmc run is its own pass (no trace domains next to --pmc) and runs under `timeout`.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
pm() { d=$1; shift; c=$1; shift; timeout 240 rocprofv3 --output-format csv --pmc $c -d $O/$d -- "$@" > $O/$d.log 2>&1; }
for cfg in "4096 128 4" "4194304 4 2"; do set -- $cfg
  pm traffic_FETCH_$1 FETCH_SIZE python $R/tools/pmc_run.py $1 $2 $3
  pm traffic_WRITE_$1 WRITE_SIZE python $R/tools/pmc_run.py $1 $2 $3
  pm sq_a_$1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" python $R/tools/pmc_run.py $1 $2 $3
  pm sq_b_$1 "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" python $R/tools/pmc_run.py $1 $2 $3
done
# calibration of FETCH_SIZE / WRITE_SIZE on KNOWN bytes in this kernel's access width (4-byte buffer loads / stores per lane,
# 34 rows in / 30 rows out per env) and on a 16-byte-per-lane copy: tools/microbench/layout_bw
pm calib_FETCH FETCH_SIZE $R/tools/microbench/layout_bw
pm calib_WRITE WRITE_SIZE $R/tools/microbench/layout_bw
# kernel statistics of the bench command itself
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -- python $R/bench.py --no-sweep --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tasks_stats -- python $R/tools/elev_probe.py > $O/tasks_probe.json 2>&1
cd $R && python tools/r02_pmc_report.py $O
