cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "runs_out or co_running or two_processes or real_occupancy" --durations=10 2>&1 | tail -25
time python tools/gdn_sync_stress.py --T 256,1000,4300 --iters 6 --co-stream
python tools/gdn_sync_stress.py --T 1000,4300,256 --iters 8 --barrier-file /tmp/go --nprocs 2 --seed 0 &
python tools/gdn_sync_stress.py --T 1000,4300,256 --iters 8 --barrier-file /tmp/go --nprocs 2 --seed 1 &
wait
} > gpurun_out/r4a/stress.log 2>&1
cat gpurun_out/r4a/stress.log
