"""A/B of the idle-wave loop of the four-phase attention kernel (waves whose query rows lie beyond Lq skip all work but the
barriers and their K/V DMA duty): bit-equality with the switch off, then interleaved timing on the DiT self-attention shapes.
usage: python scripts/attn_idle_ab.py [rounds]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import _lib, ops  # noqa: E402

DEV = "cuda"
lib = _lib.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 9


def rnd(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(torch.bfloat16).to(DEV)


def timed(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for (Lq, Lkv, H, cb) in [(4680, 9360, 40, 0), (4680, 4680, 40, 1560), (4680, 18720, 40, 0), (333, 1100, 3, 0), (4680 - 256 * 18 + 256 * 2, 2048, 8, 0)]:
    q, k, v = rnd(1, Lq, H, 128, seed=1), rnd(1, Lkv, H, 128, seed=2), rnd(1, Lkv, H, 128, seed=3)
    outs = {}
    for on in (1, 0):
        lib.rtv_attn_set_skip_idle(on)
        outs[on] = ops.attn_fwd(q, k, v, causal_block=cb).clone()
    t = {1: [], 0: []}
    for _ in range(rounds):
        for on in (1, 0):
            lib.rtv_attn_set_skip_idle(on)
            t[on].append(timed(lambda: ops.attn_fwd(q, k, v, causal_block=cb)))
    flop = 4.0 * Lq * Lkv * 128 * H * (0.5 + 0.5 / max(1, Lq // max(cb, 1)) if cb else 1.0)
    line = f"Lq {Lq} Lkv {Lkv} H {H} causal {cb}: equal={torch.equal(outs[1], outs[0])}"
    for on in (1, 0):
        ms = statistics.median(t[on])
        line += f"  {'idle loop' if on else 'all waves'} {ms * 1e3:7.1f} us {flop / ms / 1e9:6.0f} TF/s"
    print(line, flush=True)
lib.rtv_attn_set_skip_idle(1)
