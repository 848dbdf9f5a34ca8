# the whole -m gpu suite + smoke (what the driver runs at round end)      usage (through gpurun): bash tools/gpu_tests.sh [pytest args]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/tests
timeout 1200 python -m pytest tests -m gpu -q "$@" > gpurun_out/tests/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/tests/pytest.log
tail -40 gpurun_out/tests/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
