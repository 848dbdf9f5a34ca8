# round 4: counters of the two height-scan forms at 262 144 envs (lane-form step + scan), separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04scan; mkdir -p $O
timeout 120 rocprofv3 -L > $O/counters_list.txt 2>&1
pm() { d=$1; shift; c=$1; shift; timeout 300 rocprofv3 --output-format csv --pmc $c -d $O/$d -- "$@" > $O/$d.log 2>&1; }
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
SQB="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
TA1="TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum"
TC1="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"
for form in gather:8 lds:4; do
  name=${form%%:*}; fl=${form#*:}
  export WL_FLAGS=$fl
  pm FETCH_$name FETCH_SIZE python $R/tools/pmc_run.py elev 262144 4
  pm WRITE_$name WRITE_SIZE python $R/tools/pmc_run.py elev 262144 4
  pm sqa_$name "$SQA" python $R/tools/pmc_run.py elev 262144 4
  pm sqb_$name "$SQB" python $R/tools/pmc_run.py elev 262144 4
  pm ta_$name "$TA1" python $R/tools/pmc_run.py elev 262144 4
  pm tc_$name "$TC1" python $R/tools/pmc_run.py elev 262144 4
done
unset WL_FLAGS
ls $O; tail -3 $O/ta_gather.log $O/tc_gather.log
find $O -name "*.csv" | head -30
