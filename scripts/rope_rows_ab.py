"""RoPE / cache kernel, workgroup-per-row (0) vs two-waves-per-row (1), by row count (the token shards of context parallelism have
585 / 1170 / 2340 rows).  Interleaved, median; the data of the previous launch of the other variant sits in L2 / MALL (as it does in
the forward, where the QKV GEMM has just written qkv).   usage: rope_rows_ab.py [rounds]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import _lib, ops  # noqa: E402
from realtime_video_amd.rope import rope_cos_sin_table  # noqa: E402

lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 9
dev, d, H = "cuda", 5120, 40
tab = rope_cos_sin_table(128).to(dev)
w = torch.ones(d, device=dev, dtype=torch.bfloat16)
for M in (195, 585, 1170, 1560, 2340, 3120, 4680):
    qkv = torch.randn(M, 3 * d, device=dev).to(torch.bfloat16)
    arena = torch.zeros(9360, 2, H, 128, device=dev, dtype=torch.bfloat16)
    qo = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.qk_norm_rope_cache(qkv, arena[:, 0], arena[:, 1], 4680, H, w, w, tab, (3, 30, 52), 3, q_out=qo, row_offset=0)
    t = {0: [], 1: []}
    for _ in range(rounds):
        for mode in (0, 1):
            lib.rtv_rope_set_wave(mode)
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) / 20 * 1e3)
    a, b = statistics.median(t[0]), statistics.median(t[1])
    print(f"M={M:5d}: workgroup per row {a:6.1f} us   two waves per row {b:6.1f} us   ({'wave' if b < a else 'workgroup'} form wins by {abs(a - b) / max(a, b) * 100:.0f} %)")
lib.rtv_rope_set_wave(-1)
