#!/bin/bash
# PMC passes (kernel-trace only, one group per run) of the two forms of the 4096-token call over a full ring: swa_prefill_kernel (128-row)
# and swa_linearize_kernel + swa_ring256_kernel (256-row).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
G1="SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_LDS,SQ_INSTS_SALU"
G2="SQ_VALU_MFMA_BUSY_CYCLES,SQ_INSTS_MFMA,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_LDS,SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_INST_LEVEL_VMEM"
G3="TCC_HIT_sum,TCC_MISS_sum,TCP_PENDING_STALL_CYCLES_sum"
G4="FETCH_SIZE"
G5="WRITE_SIZE"
G6="GRBM_GUI_ACTIVE"
bash $R/tools/pmc.sh r06_swa_T4096 "$G1;$G2;$G3;$G4;$G5;$G6" "swa_" -- python $R/tools/kernel_bench.py --only "T=4096(full ring" > $R/gpurun_out/profiles/r06_swa_T4096_pmc.txt 2>&1
tail -80 $R/gpurun_out/profiles/r06_swa_T4096_pmc.txt
