#!/bin/bash
# r06 GPU job 6: final-build records - full GPU suite, smoke, the driver-style bench line (with cpu_baseline), kernel stats
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job6
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=6 > $O/gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_14b_final.json 2> $O/bench_14b_final.err
PMC=0 bash scripts/profile_bench.sh 14b --steps 1 --warmup 1 --no-cpu-baseline > $O/profile.log 2>&1
cp gpurun_out/prof_14b/kernel_stats.txt $O/kernel_stats_14b_final.txt
python bench.py --steps 6 --warmup 3 --no-cpu-baseline --fp8 > $O/bench_14b_fp8.json 2> $O/bench_14b_fp8.err
tail -3 $O/gpu_suite.log; tail -2 $O/smoke.log; tail -4 $O/bench_14b_final.err; head -c 400 $O/bench_14b_final.json
