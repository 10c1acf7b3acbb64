cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{
for i in 1 2 3; do
for lib in "" ab/libivl_r3.so; do
echo "== lib=$lib"
python tools/kernel_bench.py --only gdn_chunk_fused ${lib:+--lib $lib} 2>&1 | grep "gdn_chunk_fused("
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "single_launch" 2>&1 | tail -2
} > gpurun_out/r4a/ab.log 2>&1
cat gpurun_out/r4a/ab.log
