#!/bin/bash
# same-box A/B of the V cache write: GEMM epilogue (1, default) vs copy by the RoPE / cache kernel (0); interleaved, GPU box
for m in 1 0 1 0; do RTV_DIRECT_V=$m python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); k=j['config']['kernel_ms_per_block']; print('direct_v=$m', round(j['value'],2), 'frames/s', round(j['ms_per_step'],1), 'ms per block', {a:round(b,1) for a,b in k.items()})"; done
