cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{ for w in 4096 2048 1024; do echo "window $w"; python tools/kernel_bench.py --only swa_decode --window $w 2>&1 | grep swa_decode; done; } > gpurun_out/r4a/dec_window.log 2>&1
cat gpurun_out/r4a/dec_window.log
