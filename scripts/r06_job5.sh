#!/bin/bash
# r06 GPU job 5: the fresh one-frame encode on its last time tap only - parity (VAE + session tests), timing, bench A/B
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job5
mkdir -p $O
python -m pytest tests/test_vae_gpu.py tests/test_dit_gpu.py -m gpu -x -q -k "encod or vae or session or config1" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for s in 1 0 1 0; do
  echo "== fresh tap skip $s" >> $O/time_encode.log
  RTV_FRESH_TAP_SKIP=$s python - >> $O/time_encode.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from realtime_video_amd import _lib
import realtime_video_amd.vae_encoder
_lib.call("rtv_vae_set_fresh_tap_skip", int(os.environ["RTV_FRESH_TAP_SKIP"]))
exec(open("scripts/time_encode.py").read())
PY
done
for s in 1 0 1 0; do
  RTV_FRESH_TAP_SKIP=$s python bench.py --steps 6 --warmup 3 --no-cpu-baseline > $O/bench_skip${s}_$RANDOM.json 2>> $O/bench.err
done
tail -3 $O/tests.log; grep -E "==|encode 1 frame" $O/time_encode.log | awk 'NR%5==1 || NR%5==0'
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_job5/bench_skip*.json")):
    j = json.load(open(f)); c = j["config"]
    print(f.split("/")[-1], "%.3f frames/s %.1f ms" % (j["value"], j["ms_per_step"]), {k: round(v, 1) for k, v in c["kernel_ms_per_block"].items()})
PY
