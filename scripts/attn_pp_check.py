"""Four-phase attention kernel vs the lockstep one: bit-equality on a shape sweep (the two kernels accumulate in the same
order), then interleaved timing on the DiT shapes.  usage: attn_pp_check.py [--no-time]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed, dtype=torch.bfloat16, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def both(fn):
    outs = []
    for w in (81, 82):
        ops.attn_set_waves(w)
        outs.append(fn())
    ops.attn_set_waves(0)
    return outs


QUICK = "--quick" in sys.argv
bad = 0
SHAPES = [(1, 600, 1024, 2, 0, torch.bfloat16), (1, 333, 1000, 3, 96, torch.bfloat16),
                                 (2, 300, 333, 2, 0, torch.float16), (1, 256, 64, 1, 0, torch.bfloat16),
                                 (1, 100, 70, 2, 0, torch.bfloat16), (1, 257, 129, 2, 0, torch.bfloat16),
                                 (1, 1040, 1040, 2, 520, torch.bfloat16), (1, 512, 191, 8, 0, torch.bfloat16),
                                 (1, 4680, 9360, 40, 0, torch.bfloat16), (1, 4680, 4680, 40, 1560, torch.bfloat16),
                                 (1, 4680, 512, 40, 0, torch.bfloat16)]
if QUICK:
    SHAPES = [SHAPES[1], SHAPES[2], SHAPES[5], SHAPES[8]]
for (B, Lq, Lkv, H, cb, dt) in SHAPES:
    q = rnd(B, Lq, H, 128, seed=1, dtype=dt)
    k = rnd(B, Lkv, H, 128, seed=2, dtype=dt) * torch.linspace(0.3, 3.0, Lkv, device=DEV).view(1, Lkv, 1, 1).to(dt)
    v = rnd(B, Lkv, H, 128, seed=3, dtype=dt)
    a, b = both(lambda: ops.attn_fwd(q, k, v, causal_block=cb, q_offset=(Lkv - Lq if cb and Lkv > Lq else 0)))
    eq = torch.equal(a, b)
    fin = bool(torch.isfinite(b.float()).all())
    print(f"B{B} Lq{Lq} Lkv{Lkv} H{H} causal{cb} {str(dt)[6:]}: equal={eq} finite={fin} maxdiff={(a.float()-b.float()).abs().max().item():.3e}", flush=True)
    bad += (not eq) or (not fin)
# two-segment windows (second range below the first one = a wrapped ring, and above it), strided cache views
cache_k = rnd(1, 3000, 4, 128, seed=5)
cache_v = rnd(1, 3000, 4, 128, seed=6)
q = rnd(1, 520, 4, 128, seed=7)
for seg0, seg1 in [((2000, 700), (100, 333)), ((100, 333), (2000, 700)), ((64, 64), (0, 64)), ((1000, 1), (10, 130)),
                   ((500, 1000), (0, 0))][:1 if QUICK else 5]:
    a, b = both(lambda: ops.attn_fwd_win(q, cache_k, cache_v, seg0, seg1))
    kk = torch.cat([cache_k[:, seg0[0]:seg0[0] + seg0[1]], cache_k[:, seg1[0]:seg1[0] + seg1[1]]], 1)
    vv = torch.cat([cache_v[:, seg0[0]:seg0[0] + seg0[1]], cache_v[:, seg1[0]:seg1[0] + seg1[1]]], 1)
    ops.attn_set_waves(81)
    c = ops.attn_fwd(q, kk.contiguous(), vv.contiguous())
    ops.attn_set_waves(0)
    eq = torch.equal(a, b) and torch.equal(b, c)
    print(f"window {seg0} + {seg1}: equal={eq}", flush=True)
    bad += not eq
print("MISMATCHES:", bad)
if "--no-time" in sys.argv:
    sys.exit(1 if bad else 0)
for name, lq, lkv, h, cb in [("14B c=3 denoise", 4680, 9360, 40, 0), ("14B c=3 recompute", 4680, 4680, 40, 1560),
                             ("14B c=9 denoise", 4680, 18720, 40, 0), ("14B max window", 4680, 32760, 40, 0),
                             ("cross-attention", 4680, 512, 40, 0), ("1.3B c=3", 4680, 9360, 12, 0)][:2 if QUICK else 6]:
    q, k, v = rnd(1, lq, h, 128, seed=1), rnd(1, lkv, h, 128, seed=2), rnd(1, lkv, h, 128, seed=3)
    o = torch.empty_like(q)
    res = {81: [], 82: []}
    for rnd_i in range(3):
        for w in (81, 82):
            ops.attn_set_waves(w)
            for _ in range(2):
                ops.attn_fwd(q, k, v, out=o, causal_block=cb)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attn_fwd(q, k, v, out=o, causal_block=cb)
            e1.record()
            torch.cuda.synchronize()
            res[w].append(e0.elapsed_time(e1) / 10)
    ops.attn_set_waves(0)
    fl = 4.0 * lq * lkv * h * 128
    a, b = statistics.median(res[81]), statistics.median(res[82])
    print(f"attn {name:20s} {lq} x {lkv} x {h}: lockstep {a*1e3:7.0f} us {fl/a/1e9:6.0f} TF/s | four-phase {b*1e3:7.0f} us {fl/b/1e9:6.0f} TF/s  ({a/b:.2f}x)", flush=True)
