"""Per-workgroup timeline of the product GEMM (lab build scripts/micro/libgemm_tl.so, see build_gemm_timeline.sh): when does
every workgroup start / finish its K loop / finish, full tiles vs split-K units.  usage: gemm_timeline.py [M N K cfg]..."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "libgemm_tl.so"))
vp, ci, c64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.rtv_gemm.argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, ci, vp, ci, vp, ci, ci, ci, vp, ci, ci, ci, vp]
lib.rtv_gemm_workspace_bytes.restype = ctypes.c_size_t
lib.rtv_gemm_set_workspace.argtypes = [vp, ctypes.c_size_t]
lib.rtv_gemm_debug_timeline.argtypes = [vp]
lib.rtv_last_error.restype = ctypes.c_char_p

n = lib.rtv_gemm_workspace_bytes()
ws = torch.zeros(n + 256, dtype=torch.uint8, device="cuda")
off = (-ws.data_ptr()) % 256
assert lib.rtv_gemm_set_workspace(vp(ws.data_ptr() + off), n) == 0


def run(M, N, K, cfg):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    tl = torch.zeros(8192 * 4, dtype=torch.int64, device="cuda")
    s = vp(torch.cuda.current_stream().cuda_stream)

    def go():
        st = lib.rtv_gemm(vp(a.data_ptr()), K, vp(w.data_ptr()), K, vp(c.data_ptr()), N, M, N, K, vp(b.data_ptr()), 0, vp(0), 0, 0, 0,
                          vp(0), 0, 0, cfg, s)
        assert st == 0, lib.rtv_last_error()

    lib.rtv_gemm_debug_timeline(vp(0))
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    lib.rtv_gemm_debug_timeline(vp(tl.data_ptr()))
    go()
    torch.cuda.synchronize()
    lib.rtv_gemm_debug_timeline(vp(0))
    raw = tl.cpu().numpy().reshape(-1, 4)
    grid = int((raw[:, 0] != 0).sum() // 2) if (raw[:, 1] < 0).any() else 0
    if grid:   # LDS-epilogue build: [grid..2 grid) holds the K-loop end, column 1 the end of the register -> LDS pass
        ng = int(np.nonzero(raw[:, 2])[0].max()) + 1
        kend_rt = raw[ng:2 * ng, 0].copy()
        mid = raw[:ng, 1] & 0x7fffffffffffffff
        t0g = raw[:ng, 0]
        ok = raw[:ng, 2] != 0
        print(f"   epilogue split (us): K loop end -> LDS image written: median {np.median((mid - kend_rt)[ok]) / 100.0:.2f}, "
              f"image -> all stores issued: median {np.median((raw[:ng, 2] - mid)[ok]) / 100.0:.2f}")
        raw = raw[:ng].copy()
        raw[:, 1] = kend_rt
    t = raw[raw[:, 0] != 0]
    if len(t) == 0:
        print(f'== M {M} N {N} K {K} cfg {cfg}: {ms * 1e3:.0f} us - no timeline (problem ran on the 128x128 kernel)')
        return
    t0 = t[:, 0].min()
    start, kend, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0
    split = (t[:, 3] >> 32) > 0
    print(f"== M {M} N {N} K {K} cfg {cfg}: {ms * 1e3:.0f} us ({2.0 * M * N * K / ms / 1e9:.0f} TF/s); {len(t)} workgroups, "
          f"{int(split.sum())} split units; kernel span {end.max():.1f} us")
    for name, sel in (("full tiles", ~split), ("split units", split)):
        if sel.sum() == 0:
            continue
        d, kl, ep = (end - start)[sel], (kend - start)[sel], (end - kend)[sel]
        print(f"   {name:11s}: start {start[sel].min():7.1f} .. {start[sel].max():7.1f}  end {end[sel].min():7.1f} .. {end[sel].max():7.1f} | "
              f"duration median {np.median(d):6.1f} (p10 {np.quantile(d, 0.1):6.1f} p90 {np.quantile(d, 0.9):6.1f}) | "
              f"K loop {np.median(kl):6.1f} | after K loop median {np.median(ep):5.1f} p90 {np.quantile(ep, 0.9):5.1f} max {ep.max():5.1f}")
    # concurrency profile: number of workgroups alive per 20-us bucket
    edges = np.arange(0, end.max() + 20, 20)
    alive = [(int(((start < b1) & (end > b0)).sum())) for b0, b1 in zip(edges[:-1], edges[1:])]
    print("   alive per 20 us:", alive)


if __name__ == "__main__":
    args = [int(x) for x in sys.argv[1:]]
    cases = [tuple(args[i:i + 4]) for i in range(0, len(args), 4)] or [(4680, 15360, 5120, 0), (4680, 15360, 5120, 4),
                                                                     (2304, 3328, 5120, 0), (2304, 3328, 5120, 4),
                                                                     (4680, 5120, 5120, 0), (4680, 13824, 5120, 0)]
    for cse in cases:
        run(*cse)
