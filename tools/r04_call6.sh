cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c6; mkdir -p $O
timeout 200 python tools/r04_vis_nt_probe.py > $O/vis.jsonl 2> $O/vis.err; tail -2 $O/vis.err; cat $O/vis.jsonl
timeout 300 python tools/r04_stream_probe.py > $O/stream.jsonl 2> $O/stream.err; grep elev $O/stream.jsonl
timeout 900 python -m pytest tests -m gpu -q -x -k "elev or forms or depth or visual or surface or plugin or training or bench" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -6 $O/pytest.log
