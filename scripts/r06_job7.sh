#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job7
mkdir -p $O
bash scripts/pmc_hot_kernels.sh > $O/pmc_hot.log 2>&1
cp gpurun_out/pmc_hot/summary.txt $O/pmc_hot_kernels.txt
cat $O/pmc_hot_kernels.txt
