"""Phase trace of the four-phase attention kernel (lab build scripts/micro/libattn_tr.so): cycles of KV tiles 20..23 of waves 0
and 7 - the four phases (LK, QK, LV + softmax, PV) and the barrier waits between them.  usage: attn_trace_pp.py [Lq Lkv H]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", os.environ.get("ATT_LAB_LIB", "libattn_tr.so")))
vp, ci, c64, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
lib.rtv_attn_fwd.argtypes = [vp, vp, vp, vp, ci, ci, ci, ci, ci, c64, c64, c64, c64, c64, c64, c64, c64, cf, ci, ci, ci, vp]
lib.rtv_attn_debug_trace.argtypes = [vp]
Lq, Lkv, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4680, 9360, 40)
q = torch.randn(1, Lq, H, 128, device="cuda").to(torch.bfloat16)
k = torch.randn(1, Lkv, H, 128, device="cuda").to(torch.bfloat16)
v = torch.randn(1, Lkv, H, 128, device="cuda").to(torch.bfloat16)
o = torch.empty_like(q)
s = vp(torch.cuda.current_stream().cuda_stream)


def go():
    st = lib.rtv_attn_fwd(vp(q.data_ptr()), vp(k.data_ptr()), vp(v.data_ptr()), vp(o.data_ptr()), 1, Lq, Lkv, H, 128, q.stride(0),
                          q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1), o.stride(0), o.stride(1),
                          128 ** -0.5, 0, 0, 0, s)
    assert st == 0


for _ in range(3):
    go()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    go()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"attention {Lq} x {Lkv} x {H}: {ms * 1e3:.0f} us, {4.0 * Lq * Lkv * H * 128 / ms / 1e9:.0f} TF/s (instrumented build, trace off)")
tr = torch.zeros(512 * 2 * 32, dtype=torch.int32, device="cuda")
lib.rtv_attn_debug_trace(vp(tr.data_ptr()))
go()
torch.cuda.synchronize()
lib.rtv_attn_debug_trace(vp(0))
t = tr.cpu().numpy().astype("uint32").astype("int64").reshape(512, 2, 4, 8)
names = ["LK: 16 K reads + K DMA issue", "barrier", "QK: 16 MFMA", "barrier", "LV: 32 V tr reads + DMA + softmax", "barrier",
         "PV: 16 MFMA", "barrier (to next LK)"]
for w in range(2):
    ok = (t[:, w, :, 0] != 0).all(-1)
    tw = t[ok, w]                                   # [blocks, tiles, 8]
    period = tw[:, 1:, 0] - tw[:, :-1, 0]
    print(f"wave {'0' if w == 0 else '7'}: tile period median {np.median(period):.0f} cycles ({ok.sum()} blocks; ideal 2 x 1024 MFMA cycles "
          "per SIMD for its two waves)")
    d = np.diff(tw, axis=-1)                        # 7 in-tile intervals
    last = tw[:, 1:, 0] - tw[:, :-1, 7]             # PV end -> next LK start
    for i, n in enumerate(names):
        x = d[:, :, i] if i < 7 else last
        print(f"   {n:36s} median {np.median(x):6.0f}  p10 {np.quantile(x, 0.1):6.0f}  p90 {np.quantile(x, 0.9):6.0f}")
