"""gemm5 (160x256 tiles, one wave per SIMD: csrc/gemm5.hip) against the ping-pong kernel: tile config 19 (no split-K) must be
bit-identical with tile config 4 (gemm8, no split-K) for every epilogue kind; tile config 9 (split-K where tiles are few) within
fp32 re-association of it.  usage (GPU box): python scripts/gemm5_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

DEV = "cuda"
ops.ensure_gemm_workspace(torch.device(DEV))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(*shape, generator=g, device=DEV) * scale).to(torch.bfloat16)


bad = 0
for (M, N, K) in [(585, 5120, 5120), (585, 15360, 1024), (160, 256, 128), (161, 512, 256), (1, 264, 192), (700, 1536, 1536), (1170, 13824, 512),
                  (585, 5120, 13824), (2340, 5120, 1024), (37, 200, 320)]:
    a, w, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    r = rnd(M, N, seed=4)
    gate = rnd(3, N, seed=5)
    rpf = (M + 2) // 3
    for kind, kw in (("bias", dict(bias=b)), ("gelu", dict(bias=b, act=1)), ("gate+res", dict(bias=b, gate=gate, gate_stride=N, rows_per_frame=rpf, residual=r)),
                     ("plain", dict())):
        ref = ops.gemm(a, w, tile_cfg=4, **kw).clone()
        outs = [ops.gemm(a, w, tile_cfg=19, **kw).clone() for _ in range(3)]
        same = torch.equal(outs[0], ref)
        rep = torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        sp = ops.gemm(a, w, tile_cfg=9, **kw).clone()
        sp2 = ops.gemm(a, w, tile_cfg=9, **kw).clone()
        rel = float((sp.float() - ref.float()).norm() / (ref.float().norm() + 1e-30))
        print(f"M{M} N{N} K{K} {kind:9s}: cfg19 == cfg4 {same}  repeatable {rep}  max|diff| {float((outs[0].float() - ref.float()).abs().max()):.3e}  "
              f"cfg9 rel-L2 vs cfg4 {rel:.2e} repeatable {torch.equal(sp, sp2)}  nan {int(torch.isnan(outs[0].float()).sum())}", flush=True)
        bad += (not same) or (not rep) or rel > 2e-3 or not torch.equal(sp, sp2)
print("MISMATCHES:", bad)
