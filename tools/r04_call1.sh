# round 4, GPU call 1: the whole -m gpu suite with the new form / true-size tests, flag-driven A/B of the scan and drift forms,
# depth kernel variants
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 600 python tools/r04_probe.py all > $O/probe.jsonl 2> $O/probe.err; tail -3 $O/probe.err
timeout 600 python tools/depth_probe.py 4096 > $O/depth_probe.jsonl 2> $O/depth_probe.err; tail -3 $O/depth_probe.err
cat $O/probe.jsonl $O/depth_probe.jsonl
