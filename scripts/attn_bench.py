"""Attention kernel timing on the DiT shapes (HIP events, 3 rounds of 10 launches, median).  usage: attn_bench.py
(RTV_LIB_PATH selects another build of the library for A/B runs, scripts/ab_build.sh)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

for name, lq, lkv, h, cb in [("14B c=3 denoise", 4680, 9360, 40, 0), ("14B c=3 recompute (block-causal)", 4680, 4680, 40, 4680),
                             ("14B c=9 denoise", 4680, 18720, 40, 0), ("14B max window", 4680, 32760, 40, 0),
                             ("cross-attention", 4680, 512, 40, 0), ("cp8 heads (5 heads)", 4680, 9360, 5, 0),
                             ("1.3B c=3", 4680, 9360, 12, 0)]:
    q = torch.randn(1, lq, h, 128, device="cuda").to(torch.bfloat16)
    k = torch.randn(1, lkv, h, 128, device="cuda").to(torch.bfloat16)
    v = torch.randn(1, lkv, h, 128, device="cuda").to(torch.bfloat16)
    o = torch.empty_like(q)
    fn = lambda: ops.attn_fwd(q, k, v, out=o, causal_block=cb)
    for _ in range(3):
        fn()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    print(f"attn {name:34s} {lq} x {lkv} x {h}: {ms * 1e3:8.0f} us  {4.0 * lq * lkv * h * 128 / ms / 1e9:7.0f} TF/s (dense-equivalent)", flush=True)
