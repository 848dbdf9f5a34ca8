# HBM traffic of the fused drift step per launch: separate FETCH_SIZE / WRITE_SIZE passes (gpurun rule: no trace domains
# next to --pmc), workload tools/pmc_run.py (= bench.py's launch).  Corrections as MI355X_MICROARCH.md prescribes are
# applied by the reader (counters in KiB; FETCH_SIZE x2 on gfx950).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "4096 128" "4194304 4"; do set -- $cfg
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 240 rocprofv3 --output-format csv --pmc $c -d $R/gpurun_out/pmc_traffic/${c}_$1 -- python $R/tools/pmc_run.py $1 $2 > $R/gpurun_out/pmc_traffic_${c}_$1.log 2>&1
  done
done
cd $R && python tools/pmc_summarize.py gpurun_out/pmc_traffic/FETCH_SIZE_4096 gpurun_out/pmc_traffic/WRITE_SIZE_4096 gpurun_out/pmc_traffic/FETCH_SIZE_4194304 gpurun_out/pmc_traffic/WRITE_SIZE_4194304 | tee gpurun_out/pmc_traffic/summary.json
