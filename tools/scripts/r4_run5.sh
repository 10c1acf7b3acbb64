cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "non_default or short_conv or rmsnorm or fused_kernels" 2>&1 | tail -8 > gpurun_out/r4a/pytest5.log
cat gpurun_out/r4a/pytest5.log
