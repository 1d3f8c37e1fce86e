#!/bin/bash
# r06 GPU job 4: records for profiles/ on the current build - traffic JSON, rocprofv3 kernel stats of the bench command, the other
# operating points (fp8, c = 9, c = 18 steady state, 1.3B, config 5 = fp8 + c 9 + simulated 8-way CP), the simulated CP table,
# the full GPU suite and smoke().
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job4
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_suite.log 2>&1; echo "suite rc=$?" >> $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
bash scripts/gemm_traffic_table.sh > $O/gemm_traffic.log 2>&1
cp gpurun_out/gemm_traffic/traffic.json $O/traffic_14b.json; cp gpurun_out/gemm_traffic/table.txt $O/gemm_traffic_table.txt
PMC=0 bash scripts/profile_bench.sh 14b --steps 1 --warmup 1 --no-cpu-baseline > $O/profile.log 2>&1
cp gpurun_out/prof_14b/kernel_stats.txt $O/kernel_stats_14b.txt
B="--steps 6 --warmup 3 --no-cpu-baseline"
python bench.py $B > $O/bench_14b.json 2> $O/bench_14b.err
python bench.py $B --fp8 > $O/bench_14b_fp8.json 2> $O/bench_14b_fp8.err
python bench.py $B --kv-cache-num-frames 9 --warmup 4 > $O/bench_14b_c9.json 2> $O/bench_14b_c9.err
python bench.py --steps 3 --warmup 7 --no-cpu-baseline --kv-cache-num-frames 18 > $O/bench_14b_c18.json 2> $O/bench_14b_c18.err
python bench.py $B --model 1.3b > $O/bench_1p3b.json 2> $O/bench_1p3b.err
python bench.py --steps 3 --warmup 5 --no-cpu-baseline --fp8 --kv-cache-num-frames 9 --simulate-cp 8 --profile-classes all > $O/bench_config5_fp8_c9_simcp8.json 2> $O/bench_config5.err
sh scripts/simulate_cp_table.sh > $O/simulate_cp_table.log 2>&1
tail -3 $O/gpu_suite.log; tail -2 $O/smoke.log; cat $O/simulate_cp_table.log
