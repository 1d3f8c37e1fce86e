"""Run one hot kernel back to back for N seconds (for scripts/power_clock_sample.sh).  usage: burn_kernel.py gemm|gemm_fp8|attn|conv [seconds]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import _lib, ops  # noqa: E402

which, secs = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
dev = "cuda"
if which == "gemm":
    a = torch.randn(4680, 5120, device=dev).to(torch.bfloat16)
    w = (torch.randn(15360, 5120, device=dev) * 5120 ** -0.5).to(torch.bfloat16)
    out = torch.empty(4680, 15360, device=dev, dtype=torch.bfloat16)
    fn, flop = (lambda: ops.gemm(a, w, out=out)), 2.0 * 4680 * 15360 * 5120
elif which == "gemm_fp8":
    a = torch.randn(4680, 5120, device=dev).to(torch.bfloat16)
    w = (torch.randn(15360, 5120, device=dev) * 5120 ** -0.5)
    sw = float(w.abs().max()) / 448.0
    wq = (w / sw).clamp(-448, 448).to(torch.float8_e4m3fn)
    aq, sa = ops.quantize_fp8(a)
    out = torch.empty(4680, 15360, device=dev, dtype=torch.bfloat16)
    fn, flop = (lambda: ops.gemm_fp8(aq, sa, wq, sw, out=out)), 2.0 * 4680 * 15360 * 5120
elif which == "attn":
    q = torch.randn(1, 4680, 40, 128, device=dev).to(torch.bfloat16)
    k = torch.randn(1, 9360, 40, 128, device=dev).to(torch.bfloat16)
    v = torch.randn(1, 9360, 40, 128, device=dev).to(torch.bfloat16)
    o = torch.empty_like(q)
    fn, flop = (lambda: ops.attn_fwd(q, k, v, out=o)), 4.0 * 4680 * 9360 * 128 * 40
else:
    import realtime_video_amd.vae_decoder as vd
    T, H, W, C = 4, 480, 832, 96
    x = torch.randn(T + 2, H, W, C, device=dev).half()
    wt = vd.pack_conv_weight(torch.randn(C, C, 3, 3, 3) * (27 * C) ** -0.5).to(dev)
    bias = torch.zeros(C, device=dev).half()
    zeros = torch.zeros(64, device=dev).half()
    out = torch.empty(T, H, W, C, device=dev).half()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    fn = lambda: _lib.call("rtv_conv_cl", P(x), P(wt), P(bias), ctypes.c_void_p(0), C, P(out), C, T, H, W, C, C, 3, 3, 3, 0, 0,
                           P(zeros), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    flop = 2.0 * T * H * W * C * 27 * C
fn()
torch.cuda.synchronize()
t0, n = time.time(), 0
while time.time() - t0 < secs:
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    n += 200
dt = time.time() - t0
print(f"{which}: {n} launches in {dt:.1f} s = {dt / n * 1e6:.1f} us each, {flop * n / dt / 1e12:.0f} TF/s sustained")
