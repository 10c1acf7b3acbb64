cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{
for i in 1 2 3; do
for lib in "" ab/libivl_r3.so; do
echo "== lib=$lib"
python tools/kernel_bench.py --only "gdn_decode_step" ${lib:+--lib $lib} 2>&1 | grep "gdn_decode"
done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "decode_step" 2>&1 | tail -2
} > gpurun_out/r4a/dec2.log 2>&1
cat gpurun_out/r4a/dec2.log
