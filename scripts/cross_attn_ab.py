"""Cross-attention shape (4680 queries x 512 text keys x 40 heads): the attention builds against each other, interleaved.
waves 8 = 256-row lockstep kernel (default for this shape), 4 = 128-row / 4-wave build (two workgroups per CU), 82 = four-phase."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(1)
for (Lq, Lkv, H) in [(4680, 512, 40), (585, 512, 40), (4680, 512, 12)]:
    q = torch.randn(1, Lq, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(1, Lkv, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(1, Lkv, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    o = torch.empty_like(q)
    t = {}
    outs = {}
    for w in (0, 8, 4, 82):
        ops.attn_set_waves(w)
        outs[w] = ops.attn_fwd(q, k, v).clone()
        t[w] = []
    for _ in range(9):
        for w in t:
            ops.attn_set_waves(w)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attn_fwd(q, k, v, out=o)
            e1.record()
            torch.cuda.synchronize()
            t[w].append(e0.elapsed_time(e1) / 10)
    flop = 4.0 * Lq * Lkv * 128 * H
    print(f"Lq {Lq} Lkv {Lkv} H {H}: " + "  ".join(f"waves {w}: {statistics.median(x) * 1e3:6.1f} us {flop / statistics.median(x) / 1e9:5.0f} TF/s "
                                                  f"(== default: {torch.equal(outs[w], outs[0])})" for w, x in t.items()), flush=True)
ops.attn_set_waves(0)
