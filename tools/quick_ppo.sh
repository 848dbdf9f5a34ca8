# the drift learner after a change: its tests + kernel statistics of a short training run
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_fused_ppo.py tests/test_gpu_ppo_wide.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" > $O/tests.txt
timeout 200 python scripts/train_rl.py -r RSS_DRIFT_CONFIG env_setup.num_envs=4096 train.num_iterations=8 train.log.no_log=true --quiet 2>/dev/null | tail -1 > $O/train_RSS_DRIFT_CONFIG.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_drift -- python $R/scripts/train_rl.py -r RSS_DRIFT_CONFIG env_setup.num_envs=4096 train.num_iterations=4 train.log.no_log=true --quiet > /dev/null 2>&1
python - <<'PY'
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/quick/prof_drift/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
out=open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/quick/prof_drift.txt','w')
print('total ms', tot/1e6, file=out)
for r in rows[:18]: print(r['Name'][:90], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2), round(float(r['AverageNs'])/1e3,1), file=out)
PY
