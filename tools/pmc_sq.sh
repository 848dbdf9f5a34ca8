cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "4096 128" "4194304 4"; do set -- $cfg
  timeout 240 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $R/gpurun_out/pmc_sq/n$1 -- python $R/tools/pmc_run.py $1 $2 > $R/gpurun_out/pmc_sq_n$1.log 2>&1
  timeout 240 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $R/gpurun_out/pmc_sq/n$1 -- python $R/tools/pmc_run.py $1 $2 >> $R/gpurun_out/pmc_sq_n$1.log 2>&1
done
cd $R && python tools/pmc_sq_summary.py gpurun_out/pmc_sq n4096 n4194304 | tee gpurun_out/pmc_sq/summary.json
