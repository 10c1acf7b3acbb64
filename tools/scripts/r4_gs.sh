cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "graphed_step_surfaces or hipgraph" 2>&1 | tail -15 > gpurun_out/r4a/gs.log
cat gpurun_out/r4a/gs.log
