"""Does the row stride of the GEMM operands matter (HBM channel / L2 set aliasing of the 8-rows-per-piece LDS DMA)?  Same problem with
W (and A) stored at row strides K, K + 64, K + 128, K + 256 elements; interleaved."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from realtime_video_amd import ops
ops.ensure_gemm_workspace(torch.device("cuda"))
def timed(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
m = 4680
for name, n, k in [("qkv", 15360, 5120), ("o", 5120, 5120), ("ffn0", 13824, 5120), ("ffn2", 5120, 13824)]:
    b = torch.randn(n, device="cuda").to(torch.bfloat16); out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    fns = {}
    keep = []
    for pad_w, pad_a in [(0, 0), (64, 0), (128, 0), (256, 0), (64, 64), (128, 128)]:
        wbig = (torch.randn(n, k + pad_w, device="cuda") * k ** -0.5).to(torch.bfloat16); w = wbig[:, :k]
        abig = torch.randn(m, k + pad_a, device="cuda").to(torch.bfloat16); a = abig[:, :k]
        keep.append((wbig, abig))
        fns[f"ldw+{pad_w} lda+{pad_a}"] = (lambda a=a, w=w: ops.gemm(a, w, bias=b, out=out, tile_cfg=5))
    for f in fns.values():
        for _ in range(2): f()
    t = {kk: [] for kk in fns}
    for _ in range(5):
        for kk, f in fns.items(): t[kk].append(timed(f))
    print(name, "  ".join(f"[{kk}] {statistics.median(v)*1e3:.0f} us" for kk, v in t.items()), flush=True)
