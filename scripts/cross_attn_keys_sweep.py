"""How the text cross-attention's time depends on its key count (4680 queries x 40 heads, the dup-key fold of the padded text rows):
64 keys = one 64-key tile, 65 = two (the bench's 64-token prompt + the one counted padding row), 128 = two full ones, ...
usage: cross_attn_keys_sweep.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

DEV = "cuda"
g = torch.Generator().manual_seed(1)
Lq, H = 4680, 40
q = torch.randn(1, Lq, H, 128, generator=g).to(torch.bfloat16).to(DEV)
kc = torch.randn(1, 512, H, 128, generator=g).to(torch.bfloat16).to(DEV)
vc = torch.randn(1, 512, H, 128, generator=g).to(torch.bfloat16).to(DEV)
o = torch.empty_like(q)
cases = [32, 33, 63, 64, 65, 96, 128, 129, 192, 256, 512]
times = {n: [] for n in cases}
for _ in range(7):
    for n in cases:
        k, v = kc[:, :n], vc[:, :n]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.attn_fwd_dup(q, k, v, n - 1, 512 - n + 1, out=o)
        e0.record()
        for _ in range(10):
            ops.attn_fwd_dup(q, k, v, n - 1, 512 - n + 1, out=o)
        e1.record()
        torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 10)
for n in cases:
    t = statistics.median(times[n]) * 1e3
    print(f"keys {n:4d} ({(n + 63) // 64} tiles): {t:6.1f} us   Q + O bytes at {2 * Lq * H * 128 * 2 / t / 1e6:5.2f} TB/s", flush=True)
