"""GEMM time of the four projection shapes of the 1.3B layer (d 1536, ffn 8960) at M = 4680 per tile config.
usage: gemm_shapes_1_3b.py [cfg ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

cfgs = [int(c) for c in sys.argv[1:]] or [0, 1, 50, 4, 2]
ops.ensure_gemm_workspace(torch.device("cuda"))
for name, n, k in [("qkv", 4608, 1536), ("o/cq/co", 1536, 1536), ("ffn0", 8960, 1536), ("ffn2", 1536, 8960)]:
    m = 4680
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    line = f"{name:8s} N={n:5d} K={k:5d}:"
    for cfg in cfgs:
        for _ in range(5):
            ops.gemm(a, w, bias=b, out=out, tile_cfg=cfg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, w, bias=b, out=out, tile_cfg=cfg)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        line += f"  cfg{cfg}: {ms * 1e3:6.1f} us {2.0 * m * n * k / ms / 1e9:5.0f} TF/s"
    print(line, flush=True)
