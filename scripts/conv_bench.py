"""A/B of the convolution kernels of the VAE decoder (csrc/vae_conv.hip): the halo-tile kernel in its two forms (two waves per
SIMD, round 3; one wave per SIMD, round 5) vs the implicit-GEMM gather kernel, on the decoder's dominant layers (3x3x3 causal
convs with the residual add, and the 3x3 convs behind a nearest-2x upsampling), interleaved rounds, median.  Also checks against
torch's fp32 conv3d on the small shapes and the two halo forms against each other (bit-identical).
usage: python scripts/conv_bench.py [rounds]"""
import ctypes
import os
import statistics
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import _lib  # noqa: E402
from realtime_video_amd.vae_decoder import pack_conv_weight  # noqa: E402

DEV = "cuda"
lib = _lib.load()
lib.rtv_conv_set_halo.argtypes = [ctypes.c_int]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def run(x, wp, b, res, out, T, H, W, Cin, Cout, zeros, ups=False):
    _lib.call("rtv_conv_cl", p(x), p(wp), p(b), p(res), Cout, p(out), Cout, T, H, W, Cin, Cout, 1 if ups else 3, 3, 3, 1 if ups else 0, 0,
              p(zeros), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def timed(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


shapes = [(96, 96, 4, 480, 832), (192, 192, 4, 240, 416), (96, 96, 4, 60, 832), (192, 96, 4, 240, 416), (96, 192, 2, 240, 416),
          (384, 384, 4, 120, 208), (384, 384, 2, 120, 208), (384, 384, 1, 60, 104), (192, 384, 4, 120, 208), (384, 384, 4, 15, 208),
          # nearest-2x + Conv2d 3x3 (Resample): output dims; no residual
          (384, 192, 2, 120, 208, "ups"), (384, 192, 4, 240, 416, "ups"), (192, 96, 4, 480, 832, "ups")]
MODES = (3, 2, 6, 0) + ((4, 5) if lib.rtv_lab_build() else ())      # rtv_conv_set_halo: two waves per SIMD, one, one + persistent, gather (, lab forms)
NAMES = {3: "halo", 2: "halo4", 6: "halo4p", 0: "gather", 4: "h4-noDMA", 5: "h4-noEpi"}
zeros = torch.zeros(64, dtype=torch.float16, device=DEV)
with torch.backends.cudnn.flags(enabled=False):
    for si, shp in enumerate(shapes):
        Cin, Cout, T, H, W = shp[:5]
        ups = len(shp) > 5
        g = torch.Generator().manual_seed(5)
        if ups:
            x = (torch.randn(T, H // 2, W // 2, Cin, generator=g) * 0.5).half().to(DEV)
            w = (torch.randn(Cout, Cin, 1, 3, 3, generator=g) * (9 * Cin) ** -0.5).half().to(DEV)
        else:
            x = (torch.randn(T + 2, H, W, Cin, generator=g) * 0.5).half().to(DEV)
            w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).half().to(DEV)
        b = (torch.randn(Cout, generator=g) * 0.1).half().to(DEV)
        res = None if ups else torch.randn(T, H, W, Cout, generator=g).half().to(DEV)
        wp = pack_conv_weight(w).to(DEV)
        outs = {}
        for halo in MODES:
            lib.rtv_conv_set_halo(halo)
            outs[halo] = torch.empty(T, H, W, Cout, dtype=torch.float16, device=DEV)
            run(x, wp, b, res, outs[halo], T, H, W, Cin, Cout, zeros, ups)
        torch.cuda.synchronize()
        line = f"{Cin:3d}->{Cout:3d} T{T} {H}x{W}{' ups' if ups else '    '}: "
        if not ups and (si == 0 or H * W <= 240 * 416):
            xin = x.permute(3, 0, 1, 2).unsqueeze(0).float()
            ref = F.conv3d(F.pad(xin, (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0].permute(1, 2, 3, 0)
            ref = ref.half().float() + res.float()
            for halo in (3, 0):
                d = (outs[halo].float() - ref).abs().max().item()
                line += f"max|{NAMES[halo]} - fp32| {d:.2e}  "
        line += f"halo, gather differ in {(outs[3] != outs[0]).float().mean().item() * 100:.3f} % of the outputs; "
        line += f"halo4 == halo4p == halo: {torch.equal(outs[2], outs[3]) and torch.equal(outs[6], outs[3])}; "
        t = {m: [] for m in MODES}
        for _ in range(rounds):
            for halo in MODES:
                lib.rtv_conv_set_halo(halo)
                t[halo].append(timed(lambda: run(x, wp, b, res, outs[halo], T, H, W, Cin, Cout, zeros, ups)))
        flop = 2.0 * T * H * W * Cout * (9 if ups else 27) * Cin
        for halo in MODES:
            ms = statistics.median(t[halo])
            line += f" {NAMES[halo]} {ms * 1e3:7.1f} us {flop / ms / 1e9:6.0f} TF/s"
        print(line, flush=True)
lib.rtv_conv_set_halo(1)
