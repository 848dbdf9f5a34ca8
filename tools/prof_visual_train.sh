cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vis -- python $R/scripts/train_rl.py -r RSS_VISUAL_CONFIG env_setup.num_envs=1024 train.num_iterations=3 train.log.no_log=true --quiet > $R/gpurun_out/prof_vis.log 2>&1
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/prof_vis/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:16]: print(r['Name'][:100], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
