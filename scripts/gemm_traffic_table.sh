#!/bin/bash
# rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit-miss, each in its own pass, counter passes carry only --kernel-trace) over
# the four projection shapes of a 14B layer at M = 4680, then the model-vs-measured table of scripts/gemm_traffic_model.py.
# Output: gpurun_out/gemm_traffic/table.txt (copy to profiles/).   RTV_LIB_PATH selects an A/B library.
export TMPDIR=/tmp
R=$PWD
RAW=/tmp/gemm_traffic${TAG:+_$TAG}
OUT=$R/gpurun_out/gemm_traffic
rm -rf $RAW; mkdir -p $RAW $OUT
cd /tmp
i=0
for shape in "10240 5120 0" "5120 5120 1" "13824 5120 0" "5120 13824 1"; do
  i=$((i+1))
  set -- $shape
  for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum"; do
    set -- $shape; N=$1; K=$2; RES=$3
    set -- $pass; name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace -d $RAW/${name}_g$i -o r -- python $R/scripts/one_gemm.py 0 4680 $N $K 30 $RES > $RAW/${name}_g$i.log 2>&1
  done
done
cd $R
TRAFFIC_JSON=$OUT/traffic${TAG:+_$TAG}.json python scripts/gemm_traffic_model.py $RAW | tee $OUT/table${TAG:+_$TAG}.txt
