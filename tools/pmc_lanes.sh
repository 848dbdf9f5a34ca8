# SQ counter passes of the fused drift step at one env count for the given kernel forms (WlEnvBuffers.lanes)
# usage: pmc_lanes.sh <n_envs> <steps> <tag> <lanes...>   -> gpurun_out/pmc_<tag>/summary.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=$1; K=$2; TAG=$3; shift 3
for L in "$@"; do
  D=$R/gpurun_out/pmc_$TAG/l$L
  timeout 240 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $D -- python $R/tools/pmc_run.py $N $K $L > $R/gpurun_out/pmc_${TAG}_l$L.log 2>&1
  timeout 240 rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM -d $D -- python $R/tools/pmc_run.py $N $K $L >> $R/gpurun_out/pmc_${TAG}_l$L.log 2>&1
  timeout 240 rocprofv3 --output-format csv --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_TRANS SQ_INST_CYCLES_VMEM SQ_WAVE32_INSTS SQ_INSTS_VALU_MFMA_MOPS_F32 -d $D -- python $R/tools/pmc_run.py $N $K $L >> $R/gpurun_out/pmc_${TAG}_l$L.log 2>&1
done
cd $R && python tools/pmc_sq_summary.py gpurun_out/pmc_$TAG $(for L in "$@"; do echo l$L; done) | tee gpurun_out/pmc_$TAG/summary.json
