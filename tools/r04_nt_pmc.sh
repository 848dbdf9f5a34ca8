# round 4: FETCH_SIZE / WRITE_SIZE of the fused elevation step and the depth render at 4096 envs, per variant library
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04nt; mkdir -p $O
timeout 200 python $R/tools/r04_nt_probe.py > $O/probe.jsonl 2> $O/probe.err; cat $O/probe.jsonl
pm() { d=$1; shift; c=$1; shift; timeout 120 rocprofv3 --output-format csv --pmc $c -d $O/$d -- "$@" > $O/$d.log 2>&1; }
for v in "$@"; do
  lib=${v%%:*}; task=${v#*:}
  export WL_LIB=$R/gpurun_variants/lib_$lib.so
  pm FETCH_${lib}_$task FETCH_SIZE python $R/tools/pmc_run.py $task 4096 6
  pm WRITE_${lib}_$task WRITE_SIZE python $R/tools/pmc_run.py $task 4096 6
done
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "r04nt")
for d in sorted(glob.glob(O + "/*_*_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        if "elev_step_scan" in k or "depth_tile" in k:
            print(os.path.basename(d), k[:50], c, "n=%d" % len(v), "mean=%.1f" % (sum(v) / len(v)))
PY
