#!/bin/bash
# rocprofv3 PMC passes over the five hot kernels at their 14B / 832x480 shapes (scripts/one_kernel.py): MFMA busy, wave
# cycles / stalls, LDS, L2 hit rate and the memory-side traffic (FETCH_SIZE / WRITE_SIZE in their own passes).  Counter passes
# carry only --kernel-trace.  Output: gpurun_out/pmc_hot/summary.txt (copy to profiles/).
export TMPDIR=/tmp
R=$PWD
RAW=/tmp/pmc_hot
OUT=$R/gpurun_out/pmc_hot
rm -rf $RAW; mkdir -p $RAW $OUT
cd /tmp
for k in ${KERNELS:-gemm attn conv layernorm rope}; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $RAW/${k}_s$i -o r -- python $R/scripts/one_kernel.py $k > $RAW/${k}_s$i.log 2>&1
  done
done
cd $R
python scripts/pmc_table.py $RAW | tee $OUT/summary.txt
