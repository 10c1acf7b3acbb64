cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
IVL_TRACE_FUSED=1 python tools/trace_gdn.py 256 2>&1 | grep -v Warn | tail -16 > gpurun_out/r4a/trace256.log
cat gpurun_out/r4a/trace256.log
