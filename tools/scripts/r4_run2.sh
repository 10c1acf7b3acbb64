cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 2000 python -m pytest tests/test_gpu_parity.py -q -k "single_launch or misaligned" 2>&1 | tail -25 > gpurun_out/r4a/pytest2.log
cat gpurun_out/r4a/pytest2.log
