"""Is the GEMM sensitive to the operand row stride (L2 / HBM channel mapping of 256 rows x one K slab)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops
ops.ensure_gemm_workspace('cuda')

def padded(rows, k, pad, scale=1.0):
    t = torch.empty(rows, k + pad, device="cuda", dtype=torch.bfloat16)
    t[:, :k] = (torch.randn(rows, k, device="cuda") * scale).to(torch.bfloat16)
    return t[:, :k]

for (m, n, k) in ((4680, 13824, 5120), (4680, 5120, 13824), (4680, 15360, 5120)):
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for cfg in (5, 7):
        for pa, pw in ((0, 0), (64, 64), (128, 128), (192, 192), (256, 256), (64, 0), (0, 64), (1088, 1088)):
            a, w = padded(m, k, pa), padded(n, k, pw, k ** -0.5)
            for _ in range(3):
                ops.gemm(a, w, out=out, tile_cfg=cfg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm(a, w, out=out, tile_cfg=cfg)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print(f"M{m} N{n} K{k} cfg {cfg} pad A +{pa:4d} W +{pw:4d} elements: {ms:7.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s", flush=True)
