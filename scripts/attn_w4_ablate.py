"""Timing experiments on the one-wave-per-SIMD attention kernel (lab build, garbage results): what each class of filler costs.
usage (GPU box): RTV_LIB_PATH=realtime_video_amd/librtv_hip_lab.so python scripts/attn_w4_ablate.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402

NAMES = {82: "four-phase kernel", 840: "w4 0", 841: "w4 schedule 1", 842: "w4 schedule 2", 843: "w4 schedule 3",
         940: "w4 RS", 941: "w4 RS schedule 1", 942: "w4 RS schedule 2", 943: "w4 RS schedule 3", 1040: "w4 DMA_IMM", 1140: "w4 RS + DMA_IMM",
         1141: "w4 RS + DMA_IMM schedule 1", 1142: "w4 RS + DMA_IMM schedule 2",
         1150: "w4 RS+IMM LAB 1: no softmax instructions", 1160: "w4 RS+IMM LAB 2: exponentials -> adds",
         1170: "w4 RS+IMM LAB 3: no fragment reads", 1180: "w4 RS+IMM LAB 4: no DMA", 1200: "w4 RS+IMM LAB 6: matrix instructions only"}
lq, lkv, h = 4680, 9360, 40
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = (torch.randn(1, n, h, 128, generator=g, device="cuda").to(torch.bfloat16) for n in (lq, lkv, lkv))
o = torch.empty_like(q)
modes = [int(a) for a in sys.argv[1:]] or [82, 840, 841, 842, 843, 940, 941, 942, 943, 1040, 1140, 1141, 1142, 1150, 1160, 1170, 1180, 1200, 840, 82]
for mode in modes:
    ops.attn_set_waves(mode)
    fn = lambda: ops.attn_fwd(q, k, v, out=o)
    for _ in range(3):
        fn()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ms = statistics.median(ts)
    print(f"{NAMES.get(mode, mode):40s} {lq} x {lkv} x {h}: {ms * 1e3:7.0f} us  {4.0 * lq * lkv * h * 128 / ms / 1e9:6.0f} TF/s", flush=True)
ops.attn_set_waves(0)

# ---- cycle probe (lab build): shader cycles and 100 MHz ticks of the tile loop of wave 0 of every workgroup
import ctypes  # noqa: E402
from realtime_video_amd import _lib  # noqa: E402
lib = _lib.load()
if hasattr(lib, "rtv_attn_w4_probe"):
    lib.rtv_attn_w4_probe.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(1024 * 3, dtype=torch.int64, device="cuda")
    for mode in [m for m in modes if m >= 840]:
        ops.attn_set_waves(mode)
        buf.zero_()
        lib.rtv_attn_w4_probe(ctypes.c_void_p(buf.data_ptr()))
        for _ in range(5):
            ops.attn_fwd(q, k, v, out=o)
        torch.cuda.synchronize()
        lib.rtv_attn_w4_probe(ctypes.c_void_p(0))
        b = buf.view(1024, 3)[:760].double().cpu()
        cyc, ticks, n = b[:, 0], b[:, 1], b[:, 2]
        mfma = 72 if (mode - 840) // 100 % 2 else 64
        print(f"probe {NAMES.get(mode, mode)!s:40s}: {float((cyc / n).median()):7.0f} cycles per tile = {float((cyc / n / mfma).median()):5.1f} per MFMA slot; "
              f"clock {float((cyc / ticks).median()) * 100:6.0f} MHz; tile {float((ticks / n).median()) * 10:6.0f} ns", flush=True)
    ops.attn_set_waves(0)
