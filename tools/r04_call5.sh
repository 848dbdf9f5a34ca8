cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 300 python tools/r04_rng_probe.py > $O/probe.jsonl 2> $O/probe.err; tail -3 $O/probe.err; cat $O/probe.jsonl
