export RTV_BENCH_SHARED_GPU=1
for args in "--gpus 1" "--gpus 1" "--gpus 2 --cp-exchange heads" "--gpus 4 --cp-exchange heads" "--gpus 8 --cp-exchange heads" "--gpus 8 --cp-exchange rows" "--gpus 1 --no-vae" "--gpus 8 --cp-exchange heads --no-vae" "--gpus 8 --cp-exchange rows --no-vae" "--gpus 4 --cp-exchange heads --no-vae"; do
  echo "== $args"
  timeout 300 python bench.py $args --model tiny --steps 1 --warmup 2 --no-cpu-baseline --cp-attn-splits 1 --profile-classes none 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['config']['last_block_latents_checksum'], round(d['ms_per_step'],1))"
done
