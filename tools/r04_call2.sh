cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -k "elev or forms or training or plugin or surface or sharding" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout 600 python tools/r04_probe.py elev > $O/probe.jsonl 2> $O/probe.err; tail -3 $O/probe.err
cat $O/probe.jsonl
