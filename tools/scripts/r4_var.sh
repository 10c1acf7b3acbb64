cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "varlen or gdn_error or head_count or short_conv" 2>&1 | tail -15 > gpurun_out/r4a/var.log
cat gpurun_out/r4a/var.log
