"""The self-attention launch of ONE context-parallel rank (8 ranks: 5 heads x 4680 rows after the head exchange, or 40 heads x 585
rows with the row exchange) over the c = 3 / c = 9 key windows: one launch against the KV-split launch (rtv_attn_fwd_split),
interleaved.  Columns: (waves, splits); waves 0 = chosen by grid size, 4 / 8 = lockstep 128- / 256-row, 82 = four-phase."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(1)
VARIANTS = [(0, 1), (4, 1), (82, 1), (0, 2), (4, 2), (82, 2), (4, 3), (82, 3), (4, 4), (82, 4)]
for (Lq, Lkv, H) in [(4680, 9360, 5), (4680, 18720, 5), (585, 9360, 40), (2340, 9360, 10), (1170, 9360, 10)]:
    q = torch.randn(1, Lq, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(1, Lkv, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(1, Lkv, H, 128, generator=g).to(torch.bfloat16).to(DEV)
    o = torch.empty_like(q)
    ws = torch.empty(4 * H * Lq * 130, dtype=torch.float32, device=DEV)

    def run(w, s):
        ops.attn_set_waves(w)
        if s == 1:
            return ops.attn_fwd_win(q, k, v, (0, Lkv), out=o)
        return ops.attn_fwd_split(q, k, v, (0, Lkv), kv_splits=s, out=o, workspace=ws)

    base = run(0, 1).clone()
    t = {vs: [] for vs in VARIANTS}
    err = {vs: float((run(*vs).float() - base.float()).abs().max()) for vs in VARIANTS}
    for _ in range(7):
        for vs in VARIANTS:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(*vs)
            e0.record()
            for _ in range(10):
                run(*vs)
            e1.record()
            torch.cuda.synchronize()
            t[vs].append(e0.elapsed_time(e1) / 10)
    flop = 4.0 * Lq * Lkv * 128 * H
    print(f"Lq {Lq} Lkv {Lkv} H {H}:", flush=True)
    for vs, x in t.items():
        m = statistics.median(x)
        print(f"   waves {vs[0]:2d} splits {vs[1]}: {m * 1e3:7.1f} us {flop / m / 1e9:5.0f} TF/s   max|d| vs one launch {err[vs]:.2e}", flush=True)
ops.attn_set_waves(0)
