"""Launch ONE hot kernel a few times at its 14B / 832x480 shape (for rocprofv3 --pmc passes).
usage: one_kernel.py gemm|attn|conv|layernorm|rope [iters]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from realtime_video_amd import _lib, ops  # noqa: E402

which = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = "cuda"
M, d, H = 4680, 5120, 40
if which == "gemm":          # FFN-in: 4680 x 13824 x 5120, bias + GELU(tanh) epilogue (RTV_GEMM_MN=rows,cols: another problem)
    gm, gn = (int(x) for x in os.environ.get("RTV_GEMM_MN", f"{M},13824").split(","))
    a = torch.randn(gm, d, device=dev).to(torch.bfloat16)
    w = (torch.randn(gn, d, device=dev) * d ** -0.5).to(torch.bfloat16)
    b = torch.randn(gn, device=dev).to(torch.bfloat16)
    out = torch.empty(gm, gn, device=dev, dtype=torch.bfloat16)
    gemm_cfg = int(os.environ.get("RTV_GEMM_CFG", "0"))   # tile config under test (0 = default dispatch)
    ops.ensure_gemm_workspace(torch.device(dev))
    fn = lambda: ops.gemm(a, w, bias=b, act=ops.ACT_GELU_TANH, out=out, tile_cfg=gemm_cfg)
elif which == "attn":        # denoise-step self-attention: 4680 queries x 9360 cached keys x 40 heads
    # RTV_ATTN_LKV=32760: the whole cache window (max_attention_size, causal_model.py:192), K / V read in place from one layer's
    # slice of the K/V-interleaved arena [rows, 2, H, 128] like the pipeline's cache (pipeline._initialize_kv_cache)
    Lkv = int(os.environ.get("RTV_ATTN_LKV", 2 * M))
    q = torch.randn(1, M, H, 128, device=dev).to(torch.bfloat16)
    arena = torch.randn(1, Lkv, 2, H, 128, device=dev).to(torch.bfloat16)
    k, v = arena[:, :, 0], arena[:, :, 1]
    o = torch.empty_like(q)
    fn = lambda: ops.attn_fwd(q, k, v, out=o)
elif which == "layernorm":   # LN + per-frame AdaLN modulation over [4680, 5120]
    x = torch.randn(M, d, device=dev).to(torch.bfloat16)
    sh = torch.randn(3, d, device=dev).to(torch.bfloat16)
    sc = torch.randn(3, d, device=dev).to(torch.bfloat16)
    out = torch.empty_like(x)
    fn = lambda: ops.layernorm_modulate(x, shift=sh, scale=sc, frame_stride=d, rows_per_frame=1560, out=out)
elif which == "rope":        # RMSNorm(q,k) + RoPE + KV-cache write over the fused QKV output
    from realtime_video_amd.rope import rope_cos_sin_table
    qkv = torch.randn(M, 3 * d, device=dev).to(torch.bfloat16)
    arena = torch.zeros(2 * M, 2, H, 128, device=dev, dtype=torch.bfloat16)
    kc, vc = arena[:, 0], arena[:, 1]
    wq = torch.ones(d, device=dev, dtype=torch.bfloat16)
    wk = torch.ones(d, device=dev, dtype=torch.bfloat16)
    tab = rope_cos_sin_table(128).to(dev)
    qo = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.qk_norm_rope_cache(qkv, kc, vc, M, H, wq, wk, tab, (3, 30, 52), 3, q_out=qo)
elif which == "conv":        # VAE decoder 3x3x3 causal conv, 96 -> 96 channels, 4 frames at 480 x 832 (stage up3)
    import realtime_video_amd.vae_decoder as vd
    T, Hh, Ww, C = 4, 480, 832, 96
    x = torch.randn(T + 2, Hh, Ww, C, device=dev).half()
    wt = vd.pack_conv_weight(torch.randn(C, C, 3, 3, 3) * (27 * C) ** -0.5).to(dev)
    bias = torch.zeros(C, device=dev).half()
    zeros = torch.zeros(64, device=dev).half()
    out = torch.empty(T, Hh, Ww, C, device=dev).half()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    fn = lambda: _lib.call("rtv_conv_cl", P(x), P(wt), P(bias), ctypes.c_void_p(0), C, P(out), C, T, Hh, Ww, C, C, 3, 3, 3, 0, 0,
                           P(zeros), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
else:
    raise SystemExit(__doc__)
for _ in range(iters):
    fn()
torch.cuda.synchronize()
