cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "fp16_activations or gdn_golden or gdn_vs_oracle or gdn_error or recurrent or decode_step" 2>&1 | tail -12 > gpurun_out/r4a/f16.log
cat gpurun_out/r4a/f16.log
