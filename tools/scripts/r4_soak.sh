cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{ for i in 1 2 3; do timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -1; done; } > gpurun_out/r4a/soak.log 2>&1
cat gpurun_out/r4a/soak.log
