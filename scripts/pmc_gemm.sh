#!/bin/bash
# PMC passes over one GEMM launch set: MFMA busy, wave cycles / stalls, LDS, L2 hit rate.
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_gemm
mkdir -p $OUT
cd /tmp
for cfg in "$@"; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
             "TCP_TCC_READ_REQ_sum TCC_REQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $OUT/cfg${cfg}_s$i -o r -- python $R/scripts/one_gemm.py $cfg > $OUT/cfg${cfg}_s$i.log 2>&1
  done
done
cd $R
python scripts/rocpd_summary.py --pmc-only $OUT 2>/dev/null | tee $OUT/summary.txt | head -80
