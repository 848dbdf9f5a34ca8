# SQ counter passes on the visual camera kernel (augmented and plain) at 4096 envs -> gpurun_out/pmc_obs/*.csv + summary
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_obs
mkdir -p $O
pm() { d=$1; shift; c=$1; shift; timeout 240 rocprofv3 --output-format csv --pmc $c -d $O/$d -- "$@" > $O/$d.log 2>&1; }
for mode in aug plain; do
  extra=""; [ $mode = aug ] && extra="aug"
  pm ${mode}_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES" python $R/tools/obs_run.py visual 4096 20 $extra
  pm ${mode}_b "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" python $R/tools/obs_run.py visual 4096 20 $extra
  pm ${mode}_c "SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum" python $R/tools/obs_run.py visual 4096 20 $extra
done
python - <<'PY'
import csv,glob,os,collections
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_obs'
for d in sorted(glob.glob(O+'/*_[abc]')):
    fs=glob.glob(d+'/*/*counter_collection.csv')
    if not fs: print(d,'no csv'); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'visual_obs_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(os.path.basename(d), {k: round(sum(v)/len(v),1) for k,v in agg.items()})
PY
