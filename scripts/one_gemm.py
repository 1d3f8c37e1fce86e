"""Run one GEMM shape a few times (for rocprofv3 --pmc passes). usage: one_gemm.py cfg [M N K] [iters] [res]
res = 1: the fused `x + y` residual epilogue (the o / cross-o / ffn-out projections read the residual stream and write it in place)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

cfg = int(sys.argv[1])
m, n, k = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (4680, 13824, 5120)
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
ops.ensure_gemm_workspace(torch.device("cuda"))   # split-K tail units, as in the model
a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
b = torch.randn(n, device="cuda").to(torch.bfloat16)
out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
res = len(sys.argv) > 6 and sys.argv[6] == "1"
for _ in range(iters):
    ops.gemm(a, w, bias=b, out=out, tile_cfg=cfg, residual=out if res else None)
torch.cuda.synchronize()
