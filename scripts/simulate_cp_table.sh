#!/bin/sh
# Strong-scaling compute table on ONE GPU: all shards of an N-way context-parallel forward run in lockstep (bench.py
# --simulate-cp N); kernel ms per block summed over the shards, / N = one rank's compute.  usage: scripts/simulate_cp_table.sh
for n in 1 2 4 8; do
  if [ $n = 1 ]; then extra="--no-vae"; else extra="--simulate-cp $n"; fi
  python bench.py $extra --steps 3 --warmup 2 --profile-classes all --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
n = $n
r = json.loads(sys.stdin.read())
k = r['config']['kernel_ms_per_block']
tot = sum(v for c, v in k.items() if c != 'conv')
print(f'N={n}: gemm {k.get(\"gemm\", 0):7.1f}  attn {k.get(\"attn\", 0):7.1f}  ln+rope+misc {k.get(\"layernorm\", 0) + k.get(\"rope\", 0) + k.get(\"misc\", 0):6.1f}  sum/N {tot / n:7.1f} ms')
"
done
