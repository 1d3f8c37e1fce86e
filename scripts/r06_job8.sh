#!/bin/bash
# bracket-sampling A/B: today's strides vs sparser ones vs no brackets (same box, alternating)
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job8
mkdir -p $O
B="--steps 8 --warmup 3 --no-cpu-baseline"
for i in 1 2; do
  python bench.py $B > $O/cur_$i.json 2>> $O/err.log
  python bench.py $B --profile-stride gemm=29,attn=11,layernorm=23,rope=17,conv=5 > $O/sparse_$i.json 2>> $O/err.log
  python bench.py $B --profile-classes none > $O/none_$i.json 2>> $O/err.log
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_job8/*.json")):
    j = json.load(open(f)); c = j["config"]
    print(f.split("/")[-1], "%.3f frames/s %.2f ms" % (j["value"], j["ms_per_step"]), {k: round(v, 1) for k, v in c["kernel_ms_per_block"].items()}, "frac %.4f" % j["roofline"]["frac"], "attn %.0f" % (j["roofline"]["attention_TFLOPs"] or 0), "bracketed", j["roofline"]["launches"])
PY
