cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
for n in 3 4; do
  rm -f /tmp/go$n.*
  for i in $(seq 1 $n); do
    python tools/gdn_sync_stress.py --T 1000,4300,256 --iters 150 --barrier-file /tmp/go$n --nprocs $n --seed $i &
  done
  wait
done > gpurun_out/r4a/multi.log 2>&1
grep GDN_SYNC_STRESS gpurun_out/r4a/multi.log
