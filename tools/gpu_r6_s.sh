#!/bin/bash
# where do the on-the-fly split kernels spend their time?  SQ counters for two ResBlock shapes (forward / dgrad / wgrad kernels)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6s; mkdir -p $O
for shape in "RB1(32) k11" "RB1(128) k11 d1"; do
  tag=$(echo "$shape" | tr -c 'A-Za-z0-9' '_')
  bash tools/conv_pmc.sh "$shape" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE > $O/sq1_$tag.txt 2>&1
  bash tools/conv_pmc.sh "$shape" SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA > $O/sq2_$tag.txt 2>&1
  bash tools/conv_pmc.sh "$shape" SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU TCP_PENDING_STALL_CYCLES > $O/sq3_$tag.txt 2>&1
  bash tools/conv_pmc.sh "$shape" FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum > $O/mem_$tag.txt 2>&1
done
tail -n +1 $O/sq1_*.txt | head -120
