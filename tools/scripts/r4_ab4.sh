cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{
for i in 1 2 3; do
for lib in "" ab/libivl_ns4.so ab/libivl_ns5.so ab/libivl_ns6.so ab/libivl_r3.so; do
echo "== lib=$lib"
python tools/kernel_bench.py --only "gdn_chunk_fused@T" ${lib:+--lib $lib} 2>&1 | grep "fused"
done
done
} > gpurun_out/r4a/ab4.log 2>&1
cat gpurun_out/r4a/ab4.log
