"""Time the three HBM-bound row kernels of a DiT layer at the 14B shape (M = 4680 rows, d = 5120, 40 heads) - LayerNorm + AdaLN
modulation, RMSNorm, RMSNorm(q,k) + RoPE + KV-cache write - rotating over enough buffers (> 256 MB Infinity Cache) that every
launch streams from HBM, and print us / TB/s of algorithmic bytes / fraction of the 8 TB/s peak.
usage: row_kernels_bench.py [iters]        (RTV_LIB_PATH selects another build for A/B)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402
from realtime_video_amd.rope import rope_cos_sin_table  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = "cuda"
M, d, H = 4680, 5120, 40
NB = 8   # buffers per operand: 8 x 48 MB of x alone


def timed(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % len(fns)]()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


xs = [torch.randn(M, d, device=dev).to(torch.bfloat16) for _ in range(NB)]
outs = [torch.empty(M, d, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
sh = torch.randn(3, d, device=dev).to(torch.bfloat16)
sc = torch.randn(3, d, device=dev).to(torch.bfloat16)
w = torch.randn(d, device=dev).to(torch.bfloat16)
b = torch.randn(d, device=dev).to(torch.bfloat16)
res = {}
res["layernorm_modulate"] = (timed([lambda x=x, o=o: ops.layernorm_modulate(x, shift=sh, scale=sc, frame_stride=d, rows_per_frame=1560, out=o)
                                    for x, o in zip(xs, outs)]), 2.0 * M * d * 2)
res["layernorm_affine"] = (timed([lambda x=x, o=o: ops.layernorm_modulate(x, weight=w, bias=b, out=o) for x, o in zip(xs, outs)]),
                           2.0 * M * d * 2)
res["rmsnorm"] = (timed([lambda x=x, o=o: ops.rmsnorm(x, w, out=o) for x, o in zip(xs, outs)]), 2.0 * M * d * 2)
del xs, outs
qkvs = [torch.randn(M, 3 * d, device=dev).to(torch.bfloat16) for _ in range(4)]
arenas = [torch.zeros(2 * M, 2, H, 128, device=dev, dtype=torch.bfloat16) for _ in range(4)]
qos = [torch.empty(M, d, device=dev, dtype=torch.bfloat16) for _ in range(4)]
tab = rope_cos_sin_table(128).to(dev)
res["qk_norm_rope_cache"] = (timed([lambda q=q, a=a, o=o: ops.qk_norm_rope_cache(q, a[:, 0], a[:, 1], M, H, w, w, tab, (3, 30, 52), 3, q_out=o)
                                    for q, a, o in zip(qkvs, arenas, qos)]), 6.0 * M * d * 2)
for k, (us, byt) in res.items():
    print(f"{k:22s} {us:7.1f} us   {byt / us / 1e6:6.2f} TB/s   hbm_frac {byt / us / 1e6 / 8.0:.2f}")
