"""GEMM throughput with COLD weights (as in the DiT: 28 GB of weights stream through per forward, so W never sits in the
256 MiB Infinity Cache): rotate over a pool of weight matrices much larger than the cache; activations stay warm."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops
ops.ensure_gemm_workspace('cuda')
m = 4680
for name, n, k in (("qkv", 15360, 5120), ("o", 5120, 5120), ("ffn0", 13824, 5120), ("ffn2", 5120, 13824)):
    pool = max(2, int(1.5e9 // (n * k * 2)))
    ws = [(torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16) for _ in range(pool)]
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for cfg in (50, 52, 7, 1):
        for mode in ("warm", "cold"):
            iters = 24
            for i in range(4):
                ops.gemm(a, ws[i % pool if mode == "cold" else 0], out=out, tile_cfg=cfg)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                ops.gemm(a, ws[(i + 4) % pool if mode == "cold" else 0], out=out, tile_cfg=cfg)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print(f"{name:5s} cfg {cfg:2d} {mode}: {ms:7.3f} ms  {2.0 * m * n * k / ms / 1e9:7.1f} TF/s  (pool {pool} x {n*k*2/1e6:.0f} MB)", flush=True)
    del ws
