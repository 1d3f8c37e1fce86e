#!/bin/bash
# r06 GPU job 2: full GPU suite, the c = 18 bench point in steady state (7 warm-up blocks fill the 32760-row window), the per-shape
# GEMM traffic table, balanced-supertile A/B.
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job2
mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=15 > $O/gpu_suite.log 2>&1
echo "suite rc=$?" >> $O/gpu_suite.log
python bench.py --kv-cache-num-frames 18 --steps 3 --warmup 7 --no-cpu-baseline > $O/bench_14b_c18.json 2> $O/bench_14b_c18.err
bash scripts/gemm_traffic_table.sh > $O/gemm_traffic.log 2>&1
cp gpurun_out/gemm_traffic/table.txt $O/gemm_traffic_table.txt
TAG=bal RTV_LIB_PATH=$R/realtime_video_amd/librtv_hip_bal.so bash scripts/gemm_traffic_table.sh > $O/gemm_traffic_bal.log 2>&1
cp gpurun_out/gemm_traffic/table_bal.txt $O/gemm_traffic_table_bal.txt
for i in 1 2 3; do
  CP_M=4680 CP_TORCH=$([ $i = 1 ] && echo 1 || echo 0) python scripts/cp_gemm_shapes.py 0 >> $O/gemm_shapes_default.log 2>&1
  CP_M=4680 CP_TORCH=0 RTV_LIB_PATH=$R/realtime_video_amd/librtv_hip_bal.so python scripts/cp_gemm_shapes.py 0 >> $O/gemm_shapes_bal.log 2>&1
done
python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
RTV_LIB_PATH=$R/realtime_video_amd/librtv_hip_bal.so python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_bal.json 2> $O/bench_bal.err
python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_default2.json 2> $O/bench_default2.err
RTV_LIB_PATH=$R/realtime_video_amd/librtv_hip_bal.so python bench.py --steps 5 --warmup 3 --no-cpu-baseline > $O/bench_bal2.json 2> $O/bench_bal2.err
tail -4 $O/gpu_suite.log
