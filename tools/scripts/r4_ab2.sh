cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{
for i in 1 2; do
for lib in "" ab/libivl_r3.so; do
echo "== lib=$lib"
python tools/kernel_bench.py --only "gdn_chunk" ${lib:+--lib $lib} 2>&1 | grep "@B=8"
done
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "gdn" 2>&1 | tail -3
} > gpurun_out/r4a/ab2.log 2>&1
cat gpurun_out/r4a/ab2.log
