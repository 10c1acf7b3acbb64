cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
{ for tb in "4096 1" "256 8" "4096 4" "4096 8" "1024 3"; do for n in 2 4; do python tools/ab_ncw.py $tb $n; done; done; } 2>&1 | grep "^T=" > gpurun_out/r4a/ncw3.log
cat gpurun_out/r4a/ncw3.log
