mkdir -p gpurun_out
python tools/kernel_bench.py > gpurun_out/kb_base.log 2>&1
for s in 1 2 3 4 5 6; do IVL_DEBUG_PREP_STOP=$s IVL_DEBUG_SKIP_SCAN=1 python tools/kernel_bench.py 2>&1 | grep gdn_chunk | sed "s/^/prep_stop=$s scan=off /" ; done > gpurun_out/kb_ladder.log
IVL_DEBUG_SKIP_SCAN=1 python tools/kernel_bench.py 2>&1 | grep gdn_chunk | sed "s/^/prep_full scan=off /" >> gpurun_out/kb_ladder.log
IVL_DEBUG_PREP_STOP=1 python tools/kernel_bench.py 2>&1 | grep gdn_chunk | sed "s/^/prep_stop=1 scan=on /" >> gpurun_out/kb_ladder.log
cat gpurun_out/kb_base.log gpurun_out/kb_ladder.log
