R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > $R/gpurun_out/bench_v18.json 2> $R/gpurun_out/bench_v18.err; tail -c 300 $R/gpurun_out/bench_v18.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v18 -- python $R/bench.py --no-sweep --no-cpu-baseline > $R/gpurun_out/prof_v18.log 2>&1
ls $R/gpurun_out/prof_v18/*/ | head
