"""Driver of scripts/micro/libgemm_lab.so (the instrumented / experimental builds of the 256x256x64 ping-pong GEMM):
interleaved A/B timing of the schedule variants, phase traces (s_memtime) and the per-workgroup timeline.
usage: gemm_lab.py [bench] [trace]      -> prints a report, writes gpurun_out/gemm_lab_*.json"""
import ctypes
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from realtime_video_amd import ops  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "scripts", "micro", "libgemm_lab.so"))
vp = ctypes.c_void_p
lib.lab_gemm.argtypes = [ctypes.c_int, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int,
                         ctypes.POINTER(ctypes.c_int), vp]
lib.lab_gemm.restype = ctypes.c_int
OPT_NAMES = {0: "base", 8: "buffer-load", 100: "v2 (16-MFMA segments, LDS epilogue)", 102: "v2, old epilogue"}
lib.lab_gemm_v2_epi.argtypes = [vp, vp, vp, vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, vp]
lib.lab_gemm_v2_epi.restype = ctypes.c_int


def epilogue_parity():
    """v2 with every fused epilogue vs the product kernel (same K order -> bit-identical), ragged M and N edges included."""
    ok = True
    for (M, N, K) in [(4680, 5120, 5120), (585, 1536, 1024), (300, 264, 128), (4680, 13824, 512)]:
        g = torch.Generator(device="cuda").manual_seed(M + N)
        a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.bfloat16)
        b = torch.randn(N, device="cuda", generator=g).to(torch.bfloat16)
        gate = torch.randn(3, 6, N, device="cuda", generator=g).to(torch.bfloat16)
        res = torch.randn(M, N, device="cuda", generator=g).to(torch.bfloat16)
        rpf = (M + 2) // 3
        for act, use_gate, use_res in [(0, False, False), (1, False, False), (0, True, True), (2, False, True), (0, False, True)]:
            ref = ops.gemm(a, w, bias=b, act=act, gate=gate[0, 2] if use_gate else None, gate_stride=6 * N,
                           rows_per_frame=rpf if use_gate else 0, residual=res if use_res else None, tile_cfg=4)
            for old in (0, 1):
                c = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
                p = lambda t: vp(t.data_ptr()) if t is not None else vp(0)
                st = lib.lab_gemm_v2_epi(p(a), p(w), p(c), p(b), act, p(gate[0, 2]) if use_gate else vp(0), 6 * N,
                                         rpf if use_gate else 0, p(res) if use_res else vp(0), M, N, K, old,
                                         vp(torch.cuda.current_stream().cuda_stream))
                torch.cuda.synchronize()
                same = torch.equal(c, ref)
                ok &= same
                if not same:
                    d = (c.float() - ref.float()).abs()
                    print(f"!! v2 epilogue mismatch M{M} N{N} K{K} act{act} gate{use_gate} res{use_res} old_epi{old}: max {d.max().item()} "
                          f"bad {int((d > 0).sum())}")
    print("v2 epilogue parity vs product kernel (bit-exact):", "OK" if ok else "FAILED", flush=True)


def lab(opt, a, w, c, bias, stamps=None, detail=None, blocks=None, kt0=-1, traced=None):
    M, K = a.shape
    N = w.shape[0]
    tr = (ctypes.c_int * 4)(*(traced or [-1] * 4))
    p = lambda t: vp(t.data_ptr()) if t is not None else vp(0)
    st = lib.lab_gemm(opt, p(a), p(w), p(c), p(bias), M, N, K, p(stamps), p(detail), p(blocks), kt0, tr,
                      vp(torch.cuda.current_stream().cuda_stream))
    if st:
        raise RuntimeError(f"lab_gemm opt {opt} -> {st}")


def time_fn(fn, iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def bench():
    out = {}
    for name, M, N, K, nw in [("4096x16384x5120 (4 rounds, warm)", 4096, 16384, 5120, 1),
                              ("4096x16384x5120 (4 rounds, 8 weight sets = cold)", 4096, 16384, 5120, 8),
                              ("4096x4096x4096 (1 round)", 4096, 4096, 4096, 1),
                              ("8192x8192x8192 (4 rounds)", 8192, 8192, 8192, 1)]:
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16) for _ in range(nw)]
        b = torch.randn(N, device="cuda").to(torch.bfloat16)
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ref = torch.nn.functional.linear(a, ws[0], b)
        cnt = [0]

        def nextw():
            cnt[0] += 1
            return ws[cnt[0] % nw]

        fns = {"hipBLASLt (torch)": lambda: torch.nn.functional.linear(a, nextw(), b),
               "product cfg4 (gemm8, no split)": lambda: ops.gemm(a, nextw(), bias=b, out=c, tile_cfg=4),
               "product cfg50 (gemm8 + split)": lambda: ops.gemm(a, nextw(), bias=b, out=c, tile_cfg=50)}
        for opt, nm in OPT_NAMES.items():
            c.zero_()
            lab(opt, a, ws[0], c, b)
            err = (c.float() - ref.float()).abs().max().item()
            if err > 0.25:
                print(f"!! lab opt {opt} ({nm}) WRONG: max abs err {err}")
            fns[f"lab {opt:2d} {nm}"] = (lambda o: (lambda: lab(o, a, nextw(), c, b)))(opt)
        times = {k: [] for k in fns}
        for _ in range(2):
            for f in fns.values():
                f()
        for _ in range(5):
            for k, f in fns.items():
                times[k].append(time_fn(f, 10))
        print(f"== {name}")
        res = {}
        for k, t in times.items():
            med, mn = statistics.median(t), min(t)
            tf = 2.0 * M * N * K / med / 1e9
            res[k] = {"ms_median": med, "ms_min": mn, "TFLOPs_median": tf, "TFLOPs_best": 2.0 * M * N * K / mn / 1e9}
            print(f"  {k:36s} {med * 1e3:8.1f} us  {tf:7.1f} TF/s (best {2.0 * M * N * K / mn / 1e9:7.1f})", flush=True)
        out[name] = res
    return out


def trace(opt=1):
    M, N, K = 4096, 16384, 5120
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    grid = (M // 256) * (N // 256)
    traced = [0, 9, 300, 1000]
    res = {"opt": opt, "shape": [M, N, K], "runs": []}
    for kt0 in (6, 30, 31, 60):
        stamps = torch.zeros(4 * 512, dtype=torch.int32, device="cuda")
        detail = torch.zeros(4 * 2 * 32, dtype=torch.int32, device="cuda")
        blocks = torch.zeros(grid * 6, dtype=torch.int64, device="cuda")
        lab(opt, a, w, c, b, stamps, detail, blocks, kt0, traced)
        torch.cuda.synchronize()
        st = stamps.cpu().numpy().astype("uint32").reshape(4, 2, 256)
        dt = detail.cpu().numpy().astype("uint32").reshape(4, 2, 32)[:, :, :20].reshape(4, 2, 4, 5)
        bl = blocks.cpu().numpy().reshape(grid, 6)
        res["runs"].append({"kt0": kt0, "tile_stamps": st.tolist(), "detail": dt.tolist(), "blocks": bl.tolist()})
        nk = K // 64
        print(f"-- opt {opt} trace run, detail K-tile {kt0}")
        for s, blk in enumerate(traced):
            for g in range(2):
                d = [(int(st[s, g, i + 1]) - int(st[s, g, i])) & 0xffffffff for i in range(nk - 1)]
                print(f"   block {blk:4d} group {g}: cycles per K-tile  median {statistics.median(d):6.0f}  min {min(d)}  max {max(d)}  "
                      f"p90 {sorted(d)[int(0.9 * len(d))]}")
                x = dt[s, g].astype("int64")
                rows = []
                for ph in range(4):
                    t0, tv, tl, tb, tm = (int(v) for v in x[ph])
                    nxt = int(x[ph + 1][0]) if ph < 3 else None
                    seg = {"vmcnt_wait": (tv - t0) if tv else 0, "lds_wait": tl - (tv or t0), "barrier1": tb - tl, "mfma_seg": tm - tb,
                           "barrier2": (nxt - tm) if nxt is not None else None}
                    rows.append(seg)
                print("      " + " | ".join(f"p{i + 1} vm {r['vmcnt_wait']:4d} lds {r['lds_wait']:4d} b1 {r['barrier1']:4d} mfma {r['mfma_seg']:4d} "
                                            f"b2 {r['barrier2'] if r['barrier2'] is not None else -1:4d}" for i, r in enumerate(rows)))
        # block timeline: effective shader clock and per-CU gaps
        rt0, rt1, mt0, mt1, mt2 = bl[:, 0], bl[:, 1], bl[:, 2], bl[:, 3], bl[:, 4]
        dur_us = (rt1 - rt0) / 100.0
        clk = (mt2 - mt0) / ((rt1 - rt0) / 100.0) / 1e3
        print(f"   workgroup duration us: median {statistics.median(dur_us.tolist()):.1f} min {dur_us.min():.1f} max {dur_us.max():.1f}; "
              f"shader clock GHz median {statistics.median(clk.tolist()):.3f}; K-loop share of a workgroup "
              f"{statistics.median(((mt1 - mt0) / (mt2 - mt0)).tolist()):.3f}")
        cu = {}
        for i in range(grid):
            key = (int(bl[i, 5]) >> 32, int(bl[i, 5]) & 0xffffffff & 0x0000ff00 | (int(bl[i, 5]) & 0xe000))
            cu.setdefault(key, []).append((int(rt0[i]), int(rt1[i])))
        gaps = []
        for key, iv in cu.items():
            iv.sort()
            for (s0, e0), (s1, e1) in zip(iv, iv[1:]):
                gaps.append((s1 - e0) / 100.0)
        span = (rt1.max() - rt0.min()) / 100.0
        print(f"   kernel span {span:.1f} us; distinct (xcc, cu) keys {len(cu)}; blocks per key {statistics.median([len(v) for v in cu.values()])}; "
              f"gap between consecutive workgroups on a CU us: median {statistics.median(gaps) if gaps else -1:.2f} max {max(gaps) if gaps else -1:.2f}")
        print(f"   first start spread {((rt0 - rt0.min()) / 100.0)[:256].max():.2f} us", flush=True)
    return res


def trace_v2():
    M, N, K = 4096, 16384, 5120
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    grid = (M // 256) * (N // 256)
    traced = [0, 9, 300, 1000]
    import numpy as np
    for opt in (101, 103):
        for kt0 in (30, 60):
            stamps = torch.zeros(4 * 512, dtype=torch.int32, device="cuda")
            detail = torch.zeros(4 * 2 * 32, dtype=torch.int32, device="cuda")
            blocks = torch.zeros(grid * 6, dtype=torch.int64, device="cuda")
            lab(opt, a, w, c, b, stamps, detail, blocks, kt0, traced)
            torch.cuda.synchronize()
            dt = detail.cpu().numpy().astype("uint32").reshape(4, 2, 32).astype("int64")
            bl = blocks.cpu().numpy().reshape(grid, 6)
            print(f"-- v2 opt {opt} trace, detail K-tile {kt0}")
            for s_, blk in enumerate(traced):
                for g in range(2):
                    x = dt[s_, g]
                    per_tile = ((x[9] - x[8]) & 0xffffffff) / 64.0
                    ph = []
                    for i in range(2):
                        t0, tl, tb, tm = (int(v) for v in x[i * 4:i * 4 + 4])
                        nxt = int(x[4]) if i == 0 else int(x[10])
                        ph.append(f"{'XY'[i]}: lds+wait {tl - t0:4d} b1 {tb - tl:4d} mfma {tm - tb:4d} b2 {(nxt - tm) & 0xffffffff:4d}")
                    print(f"   block {blk:4d} group {g}: mean cycles per K-tile (tiles 8..72) {per_tile:7.1f} | " + " | ".join(ph))
            rt0, rt1, mt0, mt1, mt2 = bl[:, 0], bl[:, 1], bl[:, 2], bl[:, 3], bl[:, 4]
            dur = (rt1 - rt0) / 100.0
            clk = (mt2 - mt0) / dur / 1e3
            print(f"   workgroup us: median {np.median(dur):.1f} min {dur.min():.1f} max {dur.max():.1f}; clock GHz {np.median(clk):.3f}; "
                  f"epilogue cycles median {np.median(mt2 - mt1):.0f} (first round {np.median((mt2 - mt1)[:256]):.0f}, later "
                  f"{np.median((mt2 - mt1)[256:]):.0f}); kernel span {(rt1.max() - rt0.min()) / 100.0:.1f} us", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["bench", "trace"]
    if "parity" in what:
        epilogue_parity()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    if "bench" in what:
        json.dump(bench(), open(os.path.join(ROOT, "gpurun_out", "gemm_lab_bench.json"), "w"), indent=1)
    if "trace2" in what:
        trace_v2()
    if "trace" in what:
        r = [trace(1), trace(9)]
        json.dump(r, open(os.path.join(ROOT, "gpurun_out", "gemm_lab_trace.json"), "w"))
