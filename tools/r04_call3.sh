cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c3; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -k "depth or visual or surface or plugin" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 300 python tools/depth_probe.py 4096 > $O/depth_probe.jsonl 2> $O/depth_probe.err; cat $O/depth_probe.jsonl
timeout 300 python tools/depth_probe.py 65536 >> $O/depth_probe.jsonl 2>> $O/depth_probe.err; tail -1 $O/depth_probe.jsonl
