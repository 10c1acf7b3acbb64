cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
python tools/proto/run_proto.py > gpurun_out/r4a/proto.log 2>&1
cat gpurun_out/r4a/proto.log
