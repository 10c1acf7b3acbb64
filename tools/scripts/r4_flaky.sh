cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{
for i in 1 2 3 4 5; do
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "single_launch or sequence_parallel or bench_two_ranks" 2>&1 | tail -1
done
} > gpurun_out/r4a/flaky.log 2>&1
cat gpurun_out/r4a/flaky.log
