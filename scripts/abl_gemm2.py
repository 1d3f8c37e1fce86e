"""Does the GEMM speed up when every operand load hits in L1 (all rows alias one row: lda = ldw = 0)?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops
m, n, k = 4680, 13824, 5120
ops.ensure_gemm_workspace('cuda')
a_full = torch.randn(m, k, device="cuda").to(torch.bfloat16)
w_full = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
a_one = a_full[:1].expand(m, k)
w_one = w_full[:1].expand(n, k)
out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
for cfg in (4, 6):
    for name, a, w in (("real operands", a_full, w_full), ("all rows alias row 0 (L1 hits)", a_one, w_one),
                       ("A real, W aliased", a_full, w_one), ("A aliased, W real", a_one, w_full)):
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
        print(f"cfg {cfg} {name:32s} {ms:7.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s")
