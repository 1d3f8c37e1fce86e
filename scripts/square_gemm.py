"""gemm8 / gemm9 on the square shapes the CDNA4 guide quotes for its 8-phase template (random operands)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops
ops.ensure_gemm_workspace('cuda')
for n in (4096, 8192):
    a = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(n, n, device="cuda") * 2 - 1).to(torch.bfloat16)
    out = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    for cfg in (50, 7, 1):
        for _ in range(3):
            ops.gemm(a, w, out=out, tile_cfg=cfg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 50 if n == 4096 else 20
        e0.record()
        for _ in range(iters):
            ops.gemm(a, w, out=out, tile_cfg=cfg)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{n}^3 cfg {cfg}: {ms:7.3f} ms  {2.0 * n ** 3 / ms / 1e9:7.1f} TF/s", flush=True)
    ms = None
    lin = torch.nn.functional.linear
    for _ in range(3):
        lin(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lin(a, w)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{n}^3 torch.linear (hipBLASLt): {ms:7.3f} ms  {2.0 * n ** 3 / ms / 1e9:7.1f} TF/s", flush=True)
