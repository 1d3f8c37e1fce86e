#!/bin/bash
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job11
mkdir -p $O
PMC=0 bash scripts/profile_bench.sh 14b_gaps --steps 2 --warmup 3 --no-cpu-baseline --profile-classes none > $O/profile.log 2>&1
python scripts/kernel_gaps.py /tmp/prof_14b_gaps/trace > $O/kernel_gaps.txt 2>&1
cat $O/kernel_gaps.txt
