# one SQ pass (+ SQB) of a pmc_run.py workload: wave life and stall shares of its kernels    usage: r04_sq_quick.sh <task> <n> <K> [flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sqq; rm -rf $O; mkdir -p $O
export WL_FLAGS=${4:-0}
SQA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES"
SQB="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
timeout 150 rocprofv3 --output-format csv --pmc $SQA -d $O/sqa -- python $R/tools/pmc_run.py $1 $2 $3 > $O/sqa.log 2>&1
timeout 150 rocprofv3 --output-format csv --pmc $SQB -d $O/sqb -- python $R/tools/pmc_run.py $1 $2 $3 > $O/sqb.log 2>&1
python - <<PY
import csv, glob, re
from collections import defaultdict
def load(d):
    per=defaultdict(lambda: defaultdict(lambda: defaultdict(float))); dur=defaultdict(dict)
    for path in glob.glob(f"$O/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            m=re.match(r"(?:void )?(?:\(anonymous namespace\)::)?(\w+)", r["Kernel_Name"]); k=m.group(1)
            i=int(r["Dispatch_Id"]); per[k][i][r["Counter_Name"]]+=float(r["Counter_Value"]); dur[k][i]=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    out={}
    for k in per:
        ids=sorted(per[k]); ids=ids[1:] if len(ids)>=3 else ids
        names=set().union(*[per[k][i].keys() for i in ids])
        out[k]=({c:sum(per[k][i][c] for i in ids)/len(ids) for c in names}, sum(dur[k][i] for i in ids)/len(ids))
    return out
a=load("sqa"); b=load("sqb")
for k,(A,dur) in a.items():
    if not any(s in k for s in ("elev","drift","visual","depth")): continue
    B=b.get(k,({},0))[0]; wc=A["SQ_WAVE_CYCLES"]; w=A["SQ_WAVES"]
    print(k, "dur_us", round(dur/1e3,1), "waves", int(w), "wave_life_cycles", round(wc*4/w), "valu/wave", round(A["SQ_INSTS_VALU"]/w,1), "vmem_rd/wave", round(B.get("SQ_INSTS_VMEM_RD",0)/w,1), "lds/wave", round(B.get("SQ_INSTS_LDS",0)/w,1),
          "wait_any", round(A["SQ_WAIT_ANY"]/wc,3), "wait_inst", round(A["SQ_WAIT_INST_ANY"]/wc,3), "active", round(A["SQ_ACTIVE_INST_ANY"]/wc,3), "act_vmem", round(B.get("SQ_ACTIVE_INST_VMEM",0)/wc,3), "act_lds", round(B.get("SQ_ACTIVE_INST_LDS",0)/wc,3),
          "busy_cycles", round(A.get("SQ_BUSY_CYCLES",0)))
PY
