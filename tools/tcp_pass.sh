cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e/tcp; mkdir -p $O
timeout 60 rocprofv3 -L > $O/avail.txt 2>&1
grep -o "\bTCP_[A-Z0-9_]*\|\bTA_[A-Z0-9_]*\|\bTD_[A-Z0-9_]*\|\bTCC_[A-Z_]*HIT[A-Z_]*\|\bTCC_[A-Z_]*MISS[A-Z_]*\|\bTCC_REQ[A-Z_]*" $O/avail.txt | sort -u > $O/names.txt
wc -l $O/names.txt
pm() { d=$1; shift; c=$1; shift; timeout 120 rocprofv3 --output-format csv --pmc $c -d $O/$d -- "$@" > $O/$d.log 2>&1; echo "$d rc=$?"; }
export WL_LIB=$R/gpurun_variants/lib_base.so
pm p1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" python $R/tools/pmc_run.py elev 4096 32
pm p2 "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" python $R/tools/pmc_run.py elev 4096 32
pm p3 "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" python $R/tools/pmc_run.py elev 4096 32
pm p4 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" python $R/tools/pmc_run.py elev 4096 32
pm p5 "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" python $R/tools/pmc_run.py elev 4096 32
pm p6 "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum" python $R/tools/pmc_run.py elev 4096 32
pm p7 "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" python $R/tools/pmc_run.py elev 4096 32
for d in p1 p2 p3 p4 p5 p6 p7; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if 'elev_step_scan' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in acc.items():
    v = v[len(v)//4:]
    print(k, 'per launch', sum(v)/len(v), 'n', len(v))
PY
done
