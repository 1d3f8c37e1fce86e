"""Is a GEMM over M rows bit-identical with the same GEMM over a shard of those rows (default dispatch)?  The context-parallel
forward relies on it (bit-identical sharded == unsharded): prints, per shape, whether rows [r0, r0 + m) of the full result equal
the shard's result for m = M / 2, / 4, / 8.   usage: gemm_shard_identity.py [d ffn]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

d, ffn = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 2048)
M = 4680
ops.ensure_gemm_workspace(torch.device("cuda"))
g = torch.Generator(device="cuda").manual_seed(0)
for name, n, k, act in (("qkv", 3 * d, d, 0), ("q", d, d, 0), ("kv", 2 * d, d, 0), ("o", d, d, 0), ("ffn0", ffn, d, 1), ("ffn2", d, ffn, 0)):
    a = torch.randn(M, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
    full = ops.gemm(a, w, bias=b, act=act)
    line = f"{name:5s} N={n:5d} K={k:5d}:"
    for world in (2, 4, 8):
        m = M // world
        same = all(torch.equal(ops.gemm(a[r * m:(r + 1) * m].contiguous(), w, bias=b, act=act), full[r * m:(r + 1) * m]) for r in range(world))
        line += f"  /{world}: {'identical' if same else 'DIFFERENT'}"
    for cfg in (1, 4, 6):
        line += f"  cfg{cfg}==cfg0: {torch.equal(ops.gemm(a, w, bias=b, act=act, tile_cfg=cfg), full)}"
    print(line)
