"""fp8 (e4m3) GEMM vs the bf16 default on the 14B DiT shapes (incl. the dynamic activation quantisation pass)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import ops
ops.ensure_gemm_workspace('cuda')
m = 4680


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for name, n, k in (("qkv", 15360, 5120), ("o", 5120, 5120), ("ffn0", 13824, 5120), ("ffn2", 5120, 13824)):
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * k ** -0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    sw = w.float().abs().max() / 448.0
    wq = (w.float() / sw).clamp(-448, 448).to(torch.float8_e4m3fn)
    aq, sa = ops.quantize_fp8(a)
    fl = 2.0 * m * n * k
    t_bf16 = timeit(lambda: ops.gemm(a, w, bias=b, out=out))
    t_f8 = timeit(lambda: ops.gemm_fp8(aq, sa, wq, float(sw), bias=b, out=out))
    t_q = timeit(lambda: ops.quantize_fp8(a, out=aq))
    print(f"{name:5s} bf16 {t_bf16:6.3f} ms {fl/t_bf16/1e9:7.1f} TF/s | fp8 gemm {t_f8:6.3f} ms {fl/t_f8/1e9:7.1f} TF/s | quantize {t_q:6.3f} ms "
          f"| fp8 total {t_f8+t_q:6.3f} ms ({t_bf16/(t_f8+t_q):.2f}x)", flush=True)
