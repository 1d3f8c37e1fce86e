#!/bin/bash
# PMC passes (MFMA busy, wave stalls, LDS, clock) over the FFN-in GEMM for several tile configs: usage pmc_gemm_cfgs.sh cfg ...
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03
mkdir -p $OUT
cd /tmp
for cfg in "$@"; do
  RAW=/tmp/pmc_g$cfg
  rm -rf $RAW; mkdir -p $RAW
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    RTV_GEMM_CFG=$cfg rocprofv3 --pmc $set --kernel-trace -d $RAW/gemm_s$i -o r -- python $R/scripts/one_kernel.py gemm > $RAW/gemm_s$i.log 2>&1
  done
  # duration from an un-instrumented pass
  RTV_GEMM_CFG=$cfg rocprofv3 --kernel-trace -d $RAW/gemm_s9 -o r -- python $R/scripts/one_kernel.py gemm > $RAW/gemm_s9.log 2>&1
  echo "== tile cfg $cfg"
  python $R/scripts/pmc_table.py $RAW | grep -v "no data"
done | tee $OUT/pmc_gemm_cfgs.txt
