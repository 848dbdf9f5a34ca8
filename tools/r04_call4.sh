cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c4; mkdir -p $O
timeout 300 python tools/r04_fused_probe.py > $O/probe.jsonl 2> $O/probe.err; tail -3 $O/probe.err; cat $O/probe.jsonl
timeout 900 python -m pytest tests -m gpu -q -x -k "elev or forms or training or plugin or surface" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -12 $O/pytest.log
