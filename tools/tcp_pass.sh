# Texture-unit / L1 counters of one workload's kernels (round 6: how the fused elevation launch's scan phase was found to be bound by
# the texture unit's address rate).  usage (through gpurun): bash tools/tcp_pass.sh <tag> "task:envs:K ..."   e.g. "elev:4096:32 depth:4096:8"
# Each rocprofv3 --pmc run is its own pass, under `timeout`; per kernel and counter the average over the later half of the dispatches.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-tcp}; O=$R/gpurun_out/$TAG; mkdir -p $O
WORK=${2:-"elev:4096:32"}
pm() { d=$1; shift; c=$1; shift; timeout 150 rocprofv3 --output-format csv --pmc $c -d $O/$d -- "$@" > $O/$d.log 2>&1; }
for w in $WORK; do
  IFS=: read task n k <<< "$w"
  pm ${task}_${n}_a "TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" python $R/tools/pmc_run.py $task $n $k
  pm ${task}_${n}_b "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" python $R/tools/pmc_run.py $task $n $k
  pm ${task}_${n}_c "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_PENDING_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" python $R/tools/pmc_run.py $task $n $k
  python3 - $O ${task}_${n} <<'PY'
import csv, sys, collections, glob, json
O, key = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/{key}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in acc.items():
    if not any(s in k for s in ("elev_", "visual_", "drift_", "depth")):
        continue
    out[k] = {c: round(sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1), 1) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps({key: out}))
PY
done
