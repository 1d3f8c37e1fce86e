"""Kernel micro-benchmarks on the MI355X (HIP-event timed): GEMM tile configs on the DiT shapes,
attention on the KV-cache shapes, fused elementwise kernels.  Writes gpurun_out/kbench.json."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops  # noqa: E402


def timeit(fn, iters=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    res = {"gemm": [], "attn": [], "elementwise": []}
    dev = "cuda"
    M = 4680
    shapes = [("qkv14", M, 15360, 5120), ("o14", M, 5120, 5120), ("ffn0_14", M, 13824, 5120),
              ("ffn2_14", M, 5120, 13824), ("qkv1.3", M, 4608, 1536), ("ffn0_1.3", M, 8960, 1536),
              ("ffn2_1.3", M, 1536, 8960), ("cp8_ffn0_14", 585, 13824, 5120)]
    cfgs = [int(c) for c in os.environ.get("KBENCH_CFGS", "1,4,5").split(",")]
    for name, m, n, k in shapes:
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
        b = torch.randn(n, device=dev).to(torch.bfloat16)
        out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
        for cfg in cfgs:
            ms = timeit(lambda: ops.gemm(a, w, bias=b, out=out, tile_cfg=cfg))
            tf = 2.0 * m * n * k / ms / 1e9
            res["gemm"].append({"shape": name, "M": m, "N": n, "K": k, "cfg": cfg, "ms": ms, "TFLOPs": tf})
            print(f"gemm {name:12s} cfg{cfg} {ms:8.3f} ms {tf:8.1f} TF/s", flush=True)
        # library reference for context (hipBLASLt through torch)
        ms = timeit(lambda: torch.nn.functional.linear(a, w, b))
        print(f"gemm {name:12s} torch {ms:8.3f} ms {2.0*m*n*k/ms/1e9:8.1f} TF/s", flush=True)
        res["gemm"].append({"shape": name, "cfg": "torch.linear", "ms": ms, "TFLOPs": 2.0 * m * n * k / ms / 1e9})
    for name, lq, lkv, h in [("14B_c3", 4680, 9360, 40), ("14B_c9", 4680, 18720, 40), ("14B_max", 4680, 32760, 40),
                             ("1.3B_c3", 4680, 9360, 12), ("cross14", 4680, 512, 40), ("cp8_14B_c3", 585, 9360, 40)]:
        q = torch.randn(1, lq, h, 128, device=dev).to(torch.bfloat16)
        k = torch.randn(1, lkv, h, 128, device=dev).to(torch.bfloat16)
        v = torch.randn(1, lkv, h, 128, device=dev).to(torch.bfloat16)
        o = torch.empty_like(q)
        ms = timeit(lambda: ops.attn_fwd(q, k, v, out=o), iters=5, warmup=2)
        tf = 4.0 * lq * lkv * h * 128 / ms / 1e9
        res["attn"].append({"shape": name, "Lq": lq, "Lkv": lkv, "H": h, "ms": ms, "TFLOPs": tf})
        print(f"attn {name:12s} {ms:8.3f} ms {tf:8.1f} TF/s", flush=True)
        try:
            qt, kt, vt = (t.transpose(1, 2) for t in (q, k, v))
            ms = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(qt, kt, vt), iters=3, warmup=1)
            print(f"attn {name:12s} torch-sdpa {ms:8.3f} ms {4.0*lq*lkv*h*128/ms/1e9:8.1f} TF/s", flush=True)
            res["attn"].append({"shape": name, "impl": "torch.sdpa", "ms": ms, "TFLOPs": 4.0 * lq * lkv * h * 128 / ms / 1e9})
        except Exception as ex:  # noqa: BLE001
            print("torch sdpa failed:", ex)
    d = 5120
    x = torch.randn(M, d, device=dev).to(torch.bfloat16)
    emod = torch.randn(3, 6, d, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    ms = timeit(lambda: ops.layernorm_modulate(x, 1e-6, shift=emod[0, 0], scale=emod[0, 1], frame_stride=6 * d,
                                               rows_per_frame=1560, out=out))
    res["elementwise"].append({"kernel": "layernorm_modulate", "ms": ms, "GBps": 2 * M * d * 2 / ms / 1e6})
    print(f"layernorm_modulate {ms:.4f} ms {2*M*d*2/ms/1e6:.0f} GB/s")
    from realtime_video_amd.rope import rope_cos_sin_table
    qkv = torch.randn(M, 3 * d, device=dev).to(torch.bfloat16)
    kc = torch.zeros(9360, 40, 128, device=dev, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    wq = torch.ones(d, device=dev, dtype=torch.bfloat16)
    cs = rope_cos_sin_table(128).to(dev)
    qo = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: ops.qk_norm_rope_cache(qkv, kc, vc, 4680, 40, wq, wq, cs, (3, 30, 52), 3, q_out=qo))
    res["elementwise"].append({"kernel": "qk_norm_rope_cache", "ms": ms, "GBps": 6 * M * d * 2 / ms / 1e6})
    print(f"qk_norm_rope_cache {ms:.4f} ms {6*M*d*2/ms/1e6:.0f} GB/s")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/kbench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
