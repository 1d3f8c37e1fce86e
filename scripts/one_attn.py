"""Run the attention kernel a few times (for rocprofv3 --pmc passes) and print its time.
usage: one_attn.py [Lq Lkv H] [iters] [waves]      (waves: 0 auto, 4 or 8 - rtv_attn_set_waves)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

lq, lkv, h = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4680, 9360, 40)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
q = torch.randn(1, lq, h, 128, device="cuda").to(torch.bfloat16)
k = torch.randn(1, lkv, h, 128, device="cuda").to(torch.bfloat16)
v = torch.randn(1, lkv, h, 128, device="cuda").to(torch.bfloat16)
o = torch.empty_like(q)
waves = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ops.attn_set_waves(waves)
for _ in range(3):
    ops.attn_fwd(q, k, v, out=o)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.attn_fwd(q, k, v, out=o)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"attn Lq={lq} Lkv={lkv} H={h}: {ms:.3f} ms  {4.0 * lq * lkv * h * 128 / ms / 1e9:.1f} TF/s  waves={waves}")
