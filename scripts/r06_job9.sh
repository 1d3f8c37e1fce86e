#!/bin/bash
# sliding concat windows in the VAE decoder: parity (whole VAE file + session / CP tests that decode), A/B against the previous build
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/r06_job9
mkdir -p $O
python -m pytest tests/test_vae_gpu.py tests/test_dit_gpu.py tests/test_context_parallel_gpu.py -m gpu -x -q -k "vae or decod or session or config1 or rank or two_process" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
B="--steps 8 --warmup 3 --no-cpu-baseline"
for i in 1 2; do
  python bench.py $B > $O/new_$i.json 2>> $O/err.log
  RTV_LIB_PATH=$R/scripts/micro/librtv_HEAD.so python bench.py $B > $O/old_$i.json 2>> $O/err.log
done
PMC=0 bash scripts/profile_bench.sh 14b_slide --steps 2 --warmup 2 --no-cpu-baseline > $O/profile.log 2>&1
grep -E "copyBuffer|conv_halo4p|kernel  " gpurun_out/prof_14b_slide/kernel_stats.txt > $O/kernel_stats_excerpt.txt
tail -3 $O/tests.log
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_job9/*_?.json")):
    j = json.load(open(f)); c = j["config"]
    print(f.split("/")[-1], "%.3f frames/s %.2f ms" % (j["value"], j["ms_per_step"]), {k: round(v, 1) for k, v in c["kernel_ms_per_block"].items()}, "mem %.1f GB" % c["max_memory_allocated_GB"], c["last_block_latents_checksum"]["sha256_bf16"])
PY
cat $O/kernel_stats_excerpt.txt
