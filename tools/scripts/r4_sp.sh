cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "sequence_parallel or bench_two_ranks or carried_state" 2>&1 | tail -3 > gpurun_out/r4a/sp.log
cat gpurun_out/r4a/sp.log
