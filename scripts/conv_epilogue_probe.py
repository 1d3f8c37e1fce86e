import ctypes, os, statistics, sys, torch
sys.path.insert(0, "/root/repo")
from realtime_video_amd import _lib
from realtime_video_amd.vae_decoder import pack_conv_weight
lib = _lib.load()
DEV = "cuda"
def p(t): return ctypes.c_void_p(t.data_ptr() if t is not None else 0)
def timed(fn, iters=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
zeros = torch.zeros(64, dtype=torch.float16, device=DEV)
for (Cin, Cout, T, H, W) in ((96, 96, 4, 480, 832), (192, 192, 4, 240, 416)):
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(T + 2, H, W, Cin, generator=g) * 0.5).half().to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).half().to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.1).half().to(DEV)
    res = torch.randn(T, H, W, Cout, generator=g).half().to(DEV)
    wp = pack_conv_weight(w).to(DEV)
    out = torch.empty(T, H, W, Cout, dtype=torch.float16, device=DEV)
    cases = {"bias+res": (b, res), "bias": (b, None), "none": (None, None)}
    modes = [3, 2] + ([5] if lib.rtv_lab_build() else [])
    t = {(m, c): [] for m in modes for c in cases}
    for _ in range(5):
        for m in modes:
            lib.rtv_conv_set_halo(m)
            for c, (bb, rr) in cases.items():
                fn = lambda: _lib.call("rtv_conv_cl", p(x), p(wp), p(bb), p(rr), Cout, p(out), Cout, T, H, W, Cin, Cout, 3, 3, 3, 0, 0, p(zeros),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                fn(); t[(m, c)].append(timed(fn))
    for m in modes:
        print(f"{Cin}->{Cout} mode {m}: " + "  ".join(f"{c} {statistics.median(t[(m, c)]) * 1e3:7.1f} us" for c in cases), flush=True)
lib.rtv_conv_set_halo(1)
