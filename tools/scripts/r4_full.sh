cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 3000 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/r4a/pytest_full.log
cat gpurun_out/r4a/pytest_full.log
