"""Timing ablations of the gemm9 schedule (tile configs 71..77: bit0 no DMA, bit1 no LDS reads, bit2 no barriers)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops
m, n, k = 4680, 13824, 5120
a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
names = {5: "gemm8 + split-K (NL = 2)", 6: "full", 71: "no DMA", 72: "no LDS reads", 73: "no DMA, no reads", 74: "no barriers", 75: "no DMA, no barriers",
         76: "no reads, no barriers", 77: "MFMA + waits only"}
ops.ensure_gemm_workspace('cuda')
for cfg in (5, 6, 71, 72, 73, 74, 75, 76, 77):
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
    print(f"cfg {cfg:3d} {names[cfg]:24s} {ms:7.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s (no split-K: 1026 tiles on 256 CUs)")
