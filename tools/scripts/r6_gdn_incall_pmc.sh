#!/bin/bash
# SQ / TCP / TCC counters of the long-call GDN launch, hot vs behind the real in-projection GEMM (one process each).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles
G1="SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_LDS,SQ_INSTS_SALU"
G2="SQ_ACTIVE_INST_VALU,SQ_ACTIVE_INST_LDS,SQ_ACTIVE_INST_VMEM,SQ_ACTIVE_INST_SCA,SQ_WAIT_INST_LDS,SQ_INST_CYCLES_VMEM,SQ_ACTIVE_INST_MISC,SQ_VALU_MFMA_BUSY_CYCLES"
G3="SQ_INSTS_VMEM_RD,SQ_INSTS_VMEM_WR,SQ_INSTS_SMEM,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_MFMA,SQ_INST_LEVEL_VMEM,SQ_INST_LEVEL_LDS"
G4="TCP_PENDING_STALL_CYCLES_sum,TCP_TCC_READ_REQ_sum,TCP_TCC_READ_REQ_LATENCY_sum,TCP_GATE_EN1_sum"
G5="TCC_HIT_sum,TCC_MISS_sum,TCC_EA0_RDREQ_sum,TCC_EA0_RDREQ_32B_sum"
G6="FETCH_SIZE"
G7="WRITE_SIZE"
for st in hot gemm; do
  bash $R/tools/pmc.sh r06_gdn_incall_$st "$G1;$G2;$G3;$G4;$G5;$G6;$G7" "gdn_chunk_single_kernel" -- python $R/tools/gdn_incall.py 4096 - $st > $R/gpurun_out/profiles/r06_gdn_incall_$st.txt 2>&1
done
python $R/tools/gdn_incall.py 4096 > $R/gpurun_out/profiles/r06_gdn_incall_times.txt 2>&1
