"""GEMM time at the row counts of N-way context parallelism (4680 / N local rows) for the four projection shapes of the
14B layer, per tile config.  usage: cp_gemm_shapes.py [cfg ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

cfgs = [int(c) for c in sys.argv[1:]] or [0, 1, 50, 4]
ops.ensure_gemm_workspace(torch.device("cuda"))
shapes = [("qkv", 15360, 5120), ("o/cq/co", 5120, 5120), ("ffn0", 13824, 5120), ("ffn2", 5120, 13824)]
for m in [int(x) for x in os.environ.get("CP_M", "4680,2340,1170,585").split(",")]:
    for name, n, k in shapes:
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
        w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
        b = torch.randn(n, device="cuda").to(torch.bfloat16)
        out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        line = f"M={m:5d} {name:8s} N={n:5d} K={k:5d}:"
        for cfg in cfgs:
            for _ in range(3):
                ops.gemm(a, w, bias=b, out=out, tile_cfg=cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(a, w, bias=b, out=out, tile_cfg=cfg)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            line += f"  cfg{cfg}: {ms * 1e3:6.0f} us {2.0 * m * n * k / ms / 1e9:5.0f} TF/s"
        print(line, flush=True)
