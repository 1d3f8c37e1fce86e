#!/bin/bash
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_attn
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_UNALIGNED_STALL" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/s$i -o r -- python $R/scripts/one_attn.py > $OUT/s$i.log 2>&1
done
cd $R
python scripts/rocpd_summary.py --pmc-only $OUT 2>/dev/null | tee $OUT/summary.txt
