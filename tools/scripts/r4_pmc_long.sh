cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
CTRS="SQ_WAVES,SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_MFMA;SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SALU,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_VALU_MFMA_BUSY_CYCLES;FETCH_SIZE;WRITE_SIZE,TCC_HIT_sum,TCC_MISS_sum"
bash tools/pmc.sh r04_long_after "$CTRS" gdn_chunk -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py --only "gdn_chunk_fused@T=4096" > gpurun_out/profiles/r04_long_after.txt 2>&1
mv gpurun_out/profiles/r04_long_after_pmc.json gpurun_out/profiles/r04_pmc_gdn_T4096_after.json
bash tools/pmc.sh r04_long_before "$CTRS" gdn_chunk -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py --only "gdn_chunk_fused@T=4096" --lib $GRAFT_REPO_ROOT/ab/libivl_r3.so > gpurun_out/profiles/r04_long_before.txt 2>&1
mv gpurun_out/profiles/r04_long_before_pmc.json gpurun_out/profiles/r04_pmc_gdn_T4096_before_r3lib.json
tail -40 gpurun_out/profiles/r04_long_after.txt
