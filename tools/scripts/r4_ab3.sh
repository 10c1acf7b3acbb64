cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "single_launch or gdn_long_call" 2>&1 | tail -3
for i in 1 2; do
for lib in "" ab/libivl_r3.so; do
echo "== lib=$lib"
python tools/kernel_bench.py --only "gdn_chunk_fused" ${lib:+--lib $lib} 2>&1 | grep "fused"
done
done
python tools/ab_ncw.py 4096 1 0 2>&1 | grep "^T="
python tools/ab_ncw.py 1024 1 0 2>&1 | grep "^T="
python tools/ab_ncw.py 2048 2 0 2>&1 | grep "^T="
} > gpurun_out/r4a/ab3.log 2>&1
cat gpurun_out/r4a/ab3.log
