set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "single_launch or fused_front_end or bit_stable or gdn_long_call or hipgraph or misaligned" 2>&1 | tail -15 > gpurun_out/r4a/pytest.log
cat gpurun_out/r4a/pytest.log
for i in 1 2; do
python tools/kernel_bench.py --only gdn_chunk 2>&1 | tail -8
python tools/kernel_bench.py --only gdn_chunk --lib ab/libivl_r3.so 2>&1 | tail -8
done > gpurun_out/r4a/kb.log 2>&1
cat gpurun_out/r4a/kb.log
