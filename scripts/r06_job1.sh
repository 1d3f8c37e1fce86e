#!/bin/bash
# r06 GPU job 1: the configurations VERDICT r05 named (32760-row window at 14B width, config 5 literal) + the c = 18 bench point
# with its rocprofv3 attention line.  Run from the repo root on the GPU box; everything lands in gpurun_out/r06_job1/.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job1
mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_depth_gpu.py tests/test_vae_gpu.py -m gpu -x -q -s \
  -k "32760 or key_lengths or config5 or single_frame or dispatch_takes or golden_vs_reference" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
python bench.py --kv-cache-num-frames 18 --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_14b_c18.json 2> $O/bench_14b_c18.err
KERNELS=attn RTV_ATTN_LKV=32760 bash scripts/pmc_hot_kernels.sh > $O/pmc_attn_32760.log 2>&1
cp gpurun_out/pmc_hot/summary.txt $O/pmc_attn_32760_summary.txt
PMC=0 bash scripts/profile_bench.sh 14b_c18 --kv-cache-num-frames 18 --steps 1 --warmup 1 --no-cpu-baseline > $O/profile_c18.log 2>&1
cp gpurun_out/prof_14b_c18/kernel_stats.txt $O/kernel_stats_14b_c18.txt
python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_14b_default.json 2> $O/bench_14b_default.err
tail -3 $O/tests.log
