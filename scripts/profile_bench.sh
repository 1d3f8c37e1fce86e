#!/bin/bash
# rocprofv3 passes over the bench command (run on the GPU box, from the repo root):
#   1. --kernel-trace --stats           -> per-kernel time table
#   2. --pmc FETCH_SIZE   (own passes)  -> L2 memory-side read bytes per launch of the hot kernels
#   3. --pmc WRITE_SIZE   (own passes)  -> L2 memory-side write bytes per launch
# Counter passes carry only --kernel-trace (no sys/hip/hsa traces).  The raw databases stay in /tmp on the box;
# scripts/traffic_summary.py condenses them into gpurun_out/prof_<tag>/{kernel_stats.txt,traffic.json};
# copy those into profiles/ to commit them.
# usage: scripts/profile_bench.sh <tag> [bench args...]
export TMPDIR=/tmp
R=$PWD
TAG=${1:-14b}
shift
ARGS=${@:---steps 1 --warmup 1 --no-cpu-baseline}
OUT=/tmp/prof_$TAG            # raw rocpd databases (hundreds of MiB) stay on the box
SUM=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT $SUM
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o r -- python $R/bench.py $ARGS > $OUT/trace.log 2>&1
if [ "${PMC:-1}" = "0" ]; then          # PMC=0: kernel-time table only
  cd $R; python scripts/traffic_summary.py $OUT $SUM; exit 0
fi
# PMC passes run on the bare kernels at the bench's shapes (rocprofv3 --pmc over the whole multi-thousand-launch
# bench process segfaults inside the profiler on this image): the six projection GEMMs of one 14B DiT layer at
# M = 4680 rows (default tile config = what the bench uses; 100 launches per layer GEMM and pass) and the self-attention call
# of a denoising step.
i=0
for shape in "15360 5120" "5120 5120" "13824 5120" "5120 13824"; do
  i=$((i+1))
  n=100; [ "$shape" = "5120 5120" ] && n=300          # o-proj, cross-q, cross-o share a shape: weighted 3x
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch_g${i} -o r -- python $R/scripts/one_gemm.py 0 4680 $shape $n > $OUT/fetch_g$i.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write_g${i} -o r -- python $R/scripts/one_gemm.py 0 4680 $shape $n > $OUT/write_g$i.log 2>&1
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch_attn -o r -- python $R/scripts/one_attn.py 4680 9360 40 5 > $OUT/fetch_attn.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write_attn -o r -- python $R/scripts/one_attn.py 4680 9360 40 5 > $OUT/write_attn.log 2>&1
cd $R
python scripts/traffic_summary.py $OUT $SUM
