"""Thin tensor-level wrappers over the C-ABI kernels (device pointers + sizes + current HIP stream).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic happens in
librtv_hip.so.  Every wrapper refuses non-GPU tensors (no CPU fallback).
"""
import ctypes
import math

import torch

from . import _lib

ACT_NONE, ACT_GELU_TANH, ACT_SILU = 0, 1, 2
DT_BF16, DT_F16 = 0, 1
PROF_CLASSES = {"gemm": 0, "attn": 1, "layernorm": 2, "rope": 3, "conv": 4, "misc": 5}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t):
    if t.dtype == torch.bfloat16:
        return DT_BF16
    if t.dtype == torch.float16:
        return DT_F16
    raise TypeError(f"expected bfloat16 or float16 tensor, got {t.dtype}")


def _gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("realtime_video_amd kernels need GPU tensors (no CPU fallback)")


# --------------------------------------------------------------------------------------- profiling
def prof_enable(on=True, classes=None):
    """Bracket kernel launches with hipEvents: all classes, or only `classes` (names from PROF_CLASSES)."""
    mask = 0
    if on:
        mask = 0x3F if classes is None else sum(1 << PROF_CLASSES[c] for c in classes)
    _lib.call("rtv_prof_enable", int(mask))


def prof_reset():
    _lib.call("rtv_prof_reset")


def prof_set_stride(cls, stride):
    """Bracket only every `stride`-th launch of a class (an event pair costs the stream a few microseconds; prof_read then
    carries the sampled launches and `seen_*` all of them)."""
    _lib.call("rtv_prof_set_stride", PROF_CLASSES[cls] if isinstance(cls, str) else int(cls), int(stride))


def prof_bracket_overhead(n=256):
    """ms an EMPTY event bracket reads on the current stream (rtv_prof_bracket_overhead): what every bracketed launch's time
    carries on top of its kernel.  Synchronises."""
    ms = ctypes.c_double(0)
    _lib.call("rtv_prof_bracket_overhead", int(n), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(ms))
    return ms.value


def dispatch_counts(reset=False):
    """{kernel variant name: launches since load / the last reset} for the variants that ran (rtv_dispatch_counts): which GEMM /
    attention / conv kernels the dispatch rules actually chose."""
    lib = _lib.load()
    lib.rtv_dispatch_counts.restype, lib.rtv_dispatch_counts.argtypes = ctypes.c_int, [ctypes.POINTER(ctypes.c_int64), ctypes.c_int]
    lib.rtv_dispatch_name.restype, lib.rtv_dispatch_name.argtypes = ctypes.c_char_p, [ctypes.c_int]
    n = lib.rtv_dispatch_counts(None, 0)
    buf = (ctypes.c_int64 * n)()
    lib.rtv_dispatch_counts(buf, n)
    out = {lib.rtv_dispatch_name(i).decode(): int(buf[i]) for i in range(n) if buf[i]}
    if reset:
        lib.rtv_dispatch_reset.restype, lib.rtv_dispatch_reset.argtypes = ctypes.c_int, []
        lib.rtv_dispatch_reset()
    return out


def prof_read(cls):
    """-> ms / launches / work of the BRACKETED launches, seen_launches / seen_work of all launches of the class, and `ms_class` =
    the bracketed time scaled to the whole class by work."""
    c = PROF_CLASSES[cls] if isinstance(cls, str) else int(cls)
    ms, n, work = ctypes.c_double(0), ctypes.c_int64(0), ctypes.c_double(0)
    _lib.call("rtv_prof_read", c, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(work))
    sn, swork = ctypes.c_int64(0), ctypes.c_double(0)
    _lib.call("rtv_prof_read_seen", c, ctypes.byref(sn), ctypes.byref(swork))
    scale = swork.value / work.value if work.value > 0 else 1.0
    return {"ms": ms.value, "launches": n.value, "work": work.value, "seen_launches": sn.value, "seen_work": swork.value,
            "ms_class": ms.value * scale}


# --------------------------------------------------------------------------------------- attention
def attn_set_waves(waves=0):
    """Workgroup shape / kernel of attn_fwd (rtv_attn_set_waves, include/rtv_hip_lab.h): 0 = by grid size and window; 4 / 8 = 128 /
    256 query rows; 81 / 82 = 256 rows on the lockstep / four-phase schedule; 840 + v = the one-wave-per-SIMD kernel, variant v."""
    _lib.call("rtv_attn_set_waves", int(waves))


def attn_fwd(q, k, v, out=None, scale=None, causal_block=0, q_offset=0):
    """softmax(scale q k^T) v.  q:[B,Lq,H,128], k/v:[B,Lkv,H,128] (strided views allowed as long as
    the last two dims are dense), returns [B,Lq,H,128] contiguous."""
    _gpu(q, k, v)
    B, Lq, H, D = q.shape
    Lkv = k.shape[1]
    if k.shape != v.shape or k.shape[0] != B or k.shape[2] != H or k.shape[3] != D:
        raise ValueError(f"attention shape mismatch q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    if not (q.dtype == k.dtype == v.dtype):
        raise TypeError("q, k, v must share a dtype")
    for t in (q, k, v):
        if t.stride(3) != 1 or t.stride(2) != D:
            raise ValueError("attention operands need dense [H, D] inner dims (BLHD layout)")
    if out is None:
        out = torch.empty((B, Lq, H, D), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    _lib.call("rtv_attn_fwd", _ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Lq, Lkv, H, D,
              q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
              out.stride(0), out.stride(1), float(scale), int(causal_block), int(q_offset), _dt(q), _stream())
    return out


def attn_fwd_dup(q, k, v, dup_key, dup_count, out=None, scale=None):
    """Dense attention in which key `dup_key` of the window stands for `dup_count` identical keys (rtv_attn_fwd_dup)."""
    _gpu(q, k, v)
    B, Lq, H, D = q.shape
    Lkv = k.shape[1]
    if k.shape != v.shape or k.shape[0] != B or k.shape[2] != H or k.shape[3] != D:
        raise ValueError(f"attention shape mismatch q{tuple(q.shape)} k{tuple(k.shape)} v{tuple(v.shape)}")
    for t in (q, k, v):
        if t.stride(3) != 1 or t.stride(2) != D:
            raise ValueError("attention operands need dense [H, D] inner dims (BLHD layout)")
    if out is None:
        out = torch.empty((B, Lq, H, D), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    _lib.call("rtv_attn_fwd_dup", _ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Lq, Lkv, H, D,
              q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
              out.stride(0), out.stride(1), float(scale), int(dup_key), int(dup_count), _dt(q), _stream())
    return out


def attn_fwd_win(q, k_cache, v_cache, seg0, seg1=(0, 0), out=None, scale=None):
    """Attention over a key window made of two row ranges of a cache (rtv_attn_fwd_win): k_cache / v_cache [B, rows, H, 128],
    seg = (first_row, n_rows).  How a ring-indexed rolling KV cache is attended without the reference's shift copy."""
    _gpu(q, k_cache, v_cache)
    B, Lq, H, D = q.shape
    (r0, n0), (r1, n1) = seg0, seg1
    rows = k_cache.shape[1]
    if min(r0, n0, r1, n1) < 0 or r0 + n0 > rows or (n1 and r1 + n1 > rows):
        raise ValueError("attn_fwd_win: segment outside the cache")
    for t in (q, k_cache, v_cache):
        if t.stride(3) != 1 or t.stride(2) != D:
            raise ValueError("attention operands need dense [H, D] inner dims (BLHD layout)")
    if out is None:
        out = torch.empty((B, Lq, H, D), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    k, v = k_cache[:, r0:], v_cache[:, r0:]
    _lib.call("rtv_attn_fwd_win", _ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Lq, n0, n1,
              (r1 - r0) if n1 else 0, H, D, q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
              out.stride(0), out.stride(1), float(scale), 0, 0, _dt(q), _stream())
    return out


def attn_fwd_split(q, k_cache, v_cache, seg0, seg1=(0, 0), kv_splits=2, out=None, scale=None, workspace=None, causal_block=0,
                   q_offset=0):
    """attn_fwd_win with the key window cut into `kv_splits` ranges, one workgroup per (head, query tile, range), merged by a
    second kernel (rtv_attn_fwd_split): for launches too small to fill the chip.  workspace: fp32 tensor of at least
    kv_splits * B * H * Lq * 130 elements (allocated when None)."""
    _gpu(q, k_cache, v_cache)
    B, Lq, H, D = q.shape
    (r0, n0), (r1, n1) = seg0, seg1
    rows = k_cache.shape[1]
    if min(r0, n0, r1, n1) < 0 or r0 + n0 > rows or (n1 and r1 + n1 > rows):
        raise ValueError("attn_fwd_split: segment outside the cache")
    for t in (q, k_cache, v_cache):
        if t.stride(3) != 1 or t.stride(2) != D:
            raise ValueError("attention operands need dense [H, D] inner dims (BLHD layout)")
    if out is None:
        out = torch.empty((B, Lq, H, D), dtype=q.dtype, device=q.device)
    if scale is None:
        scale = 1.0 / math.sqrt(D)
    if workspace is None:
        lib = _lib.load()
        lib.rtv_attn_split_workspace_bytes.restype = ctypes.c_size_t
        lib.rtv_attn_split_workspace_bytes.argtypes = [ctypes.c_int] * 4
        need = lib.rtv_attn_split_workspace_bytes(B, Lq, H, int(kv_splits))
        workspace = torch.empty(max(need, 16) // 4, dtype=torch.float32, device=q.device)
    k, v = k_cache[:, r0:], v_cache[:, r0:]
    _lib.call("rtv_attn_fwd_split", _ptr(q), _ptr(k), _ptr(v), _ptr(out), B, Lq, n0, n1,
              (r1 - r0) if n1 else 0, H, D, q.stride(0), q.stride(1), k.stride(0), k.stride(1), v.stride(0), v.stride(1),
              out.stride(0), out.stride(1), float(scale), int(causal_block), int(q_offset), int(kv_splits), _ptr(workspace),
              workspace.numel() * workspace.element_size(), _dt(q), _stream())
    return out


# --------------------------------------------------------------------------------------- GEMM
_gemm_ws = {}


def ensure_gemm_workspace(device, stream=None):
    """Attach a split-K workspace (fp32 partial tiles + arrival counters, 64 MiB) for GEMM launches on `stream` (default: the
    current stream) of `device`.  One workspace per (device, stream): two streams - or two devices - never share slabs or
    arrival counters (rtv_gemm_set_stream_workspace)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if stream is None:
        stream = torch.cuda.current_stream(idx).cuda_stream
    key = (idx, int(stream))
    if key in _gemm_ws:
        return
    lib = _lib.load()
    lib.rtv_gemm_workspace_bytes.restype = ctypes.c_size_t
    lib.rtv_gemm_workspace_bytes.argtypes = []
    n = lib.rtv_gemm_workspace_bytes()
    with torch.cuda.device(idx):
        ws = torch.zeros(n + 256, dtype=torch.uint8, device=torch.device("cuda", idx))
        off = (-ws.data_ptr()) % 256
        try:
            _lib.call("rtv_gemm_set_stream_workspace", ctypes.c_void_p(int(stream)), ctypes.c_void_p(ws.data_ptr() + off),
                      ctypes.c_size_t(n))
        except RuntimeError as e:
            if "too many workspaces" not in str(e):
                raise
            # the library's table (64 streams over all devices) is full: GEMMs on this stream run WITHOUT split-K (same
            # results up to fp32 association, a few per cent slower on partial tile rounds) instead of failing the forward
            import warnings
            warnings.warn("rtv: split-K workspace table full; GEMMs on this stream run unsplit "
                          "(release_gemm_workspace() frees the entry of a stream that is no longer used)")
            ws = None
    _gemm_ws[key] = ws


def release_gemm_workspace(device, stream):
    """Detach and free the split-K workspace of a stream that will not launch GEMMs any more (e.g. before it is destroyed).
    Waits for the stream first: a launch still in flight may be using the slabs."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    handle = int(stream.cuda_stream if hasattr(stream, "cuda_stream") else stream)
    ws = _gemm_ws.pop((idx, handle), None)
    if ws is None:
        return
    with torch.cuda.device(idx):
        if hasattr(stream, "synchronize"):
            stream.synchronize()
        else:
            torch.cuda.synchronize(idx)
        _lib.call("rtv_gemm_set_stream_workspace", ctypes.c_void_p(handle), ctypes.c_void_p(0), ctypes.c_size_t(0))


def gemm(a, w, bias=None, act=ACT_NONE, gate=None, gate_stride=0, rows_per_frame=0, residual=None,
         out=None, tile_cfg=0, row_offset=0):
    """out[M,N] = epi(a[M,K] @ w[N,K]^T) — nn.Linear with fused bias/activation/gate/residual."""
    _gpu(a, w, bias, gate, residual, out)
    if a.dim() != 2 or w.dim() != 2 or a.shape[1] != w.shape[1]:
        raise ValueError(f"gemm shape mismatch a{tuple(a.shape)} w{tuple(w.shape)}")
    if a.stride(1) != 1 or w.stride(1) != 1:
        raise ValueError("gemm operands must be K-contiguous")
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
    if tile_cfg in (0, 5, 7, 50):
        ensure_gemm_workspace(a.device)
    _lib.call("rtv_gemm", _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(out), out.stride(0), M, N, K,
              _ptr(bias), int(act), _ptr(gate), int(gate_stride), int(rows_per_frame), int(row_offset),
              _ptr(residual), residual.stride(0) if residual is not None else 0,
              _dt(a), int(tile_cfg), _stream())
    return out


# --------------------------------------------------------------------------------------- fp8 path
def quantize_fp8(x, out=None):
    """Dynamic per-tensor e4m3 quantisation of a bf16 matrix (torchao PerTensor semantics): returns (q, scale) with
    q = e4m3(clamp(x / scale, +-448)) as a torch.float8_e4m3fn tensor and scale = max|x| / 448 as a 1-element fp32 tensor."""
    _gpu(x)
    if x.dtype != torch.bfloat16 or x.dim() != 2 or x.stride(1) != 1:
        raise TypeError("quantize_fp8 expects a K-contiguous bf16 matrix")
    M, d = x.shape
    if out is None:
        out = torch.empty((M, d), dtype=torch.float8_e4m3fn, device=x.device)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    scratch = torch.empty(1, dtype=torch.int32, device=x.device)
    _lib.call("rtv_quantize_fp8", _ptr(x), ctypes.c_int64(x.stride(0)), M, d, _ptr(out), ctypes.c_int64(out.stride(0)),
              _ptr(scale), _ptr(scratch), _stream())
    return out, scale


def gemm_fp8(a_q, a_scale, w_q, w_scale, bias=None, act=ACT_NONE, gate=None, gate_stride=0, rows_per_frame=0,
             residual=None, out=None, row_offset=0):
    """out[M,N] bf16 = epi((a_q @ w_q^T) * a_scale * w_scale): e4m3 operands, fp32 accumulation, rtv_gemm's epilogue."""
    _gpu(a_q, w_q, a_scale, bias, gate, residual, out)
    if a_q.dtype != torch.float8_e4m3fn or w_q.dtype != torch.float8_e4m3fn:
        raise TypeError("gemm_fp8 expects float8_e4m3fn operands")
    M, K = a_q.shape
    N = w_q.shape[0]
    if w_q.shape[1] != K or a_q.stride(1) != 1 or w_q.stride(1) != 1:
        raise ValueError("gemm_fp8 shape / stride mismatch")
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=a_q.device)
    ensure_gemm_workspace(a_q.device)
    _lib.call("rtv_gemm_fp8", _ptr(a_q), a_q.stride(0), _ptr(w_q), w_q.stride(0), _ptr(a_scale), ctypes.c_float(float(w_scale)),
              _ptr(out), out.stride(0), M, N, K, _ptr(bias), int(act), _ptr(gate), int(gate_stride), int(rows_per_frame),
              int(row_offset), _ptr(residual), residual.stride(0) if residual is not None else 0, _stream())
    return out


# --------------------------------------------------------------------------------------- norms
def layernorm_modulate(x, eps=1e-6, shift=None, scale=None, frame_stride=0, rows_per_frame=0,
                       weight=None, bias=None, out=None, row_offset=0):
    """LN(x) [* (1+scale[f]) + shift[f]]  or affine LN.  x:[M,d] bf16."""
    _gpu(x, shift, scale, weight, bias)
    M, d = x.shape
    if out is None:
        out = torch.empty_like(x)
    _lib.call("rtv_layernorm_modulate", _ptr(x), _ptr(out), M, d, float(eps), _ptr(shift), _ptr(scale),
              int(frame_stride), int(rows_per_frame), int(row_offset), _ptr(weight), _ptr(bias), _stream())
    return out


def rmsnorm(x, weight, eps=1e-6, out=None):
    _gpu(x, weight)
    M, d = x.shape
    if out is None:
        out = torch.empty((M, d), dtype=x.dtype, device=x.device)
    _lib.call("rtv_rmsnorm", _ptr(x), x.stride(0), _ptr(out), out.stride(0), M, d, float(eps), _ptr(weight),
              _stream())
    return out


def qk_norm_rope_cache(qkv, k_cache, v_cache, cache_row0, num_heads, wq, wk, rope_cs, grid, start_frame,
                       eps=1e-6, q_out=None, row_offset=0, ring=(0, 0, 0)):
    """qkv:[M,3d]; k_cache/v_cache:[kv_size,H,hd] views (row stride = stride(0)); grid=(F,gh,gw).
    ring = (ring_lo, ring_size, ring_shift): logical row r >= ring_lo is stored at ring_lo + (r - ring_lo + ring_shift) % ring_size."""
    _gpu(qkv, k_cache, v_cache, wq, wk, rope_cs)
    M, d3 = qkv.shape
    d = d3 // 3
    F, gh, gw = grid
    if q_out is None:
        q_out = torch.empty((M, d), dtype=qkv.dtype, device=qkv.device)
    if cache_row0 + row_offset + M > k_cache.shape[0]:
        raise ValueError("KV-cache write out of range")
    _lib.call("rtv_qk_norm_rope_cache_ring", _ptr(qkv), _ptr(q_out), _ptr(k_cache), _ptr(v_cache),
              k_cache.stride(0), int(cache_row0), M, d, int(num_heads), float(eps), _ptr(wq), _ptr(wk),
              _ptr(rope_cs), int(F), int(gh), int(gw), int(start_frame), int(row_offset),
              int(ring[0]), int(ring[1]), int(ring[2]), _stream())
    return q_out


def modulation_table(modulation, e0, out=None):
    """modulation:[L,J,d], e0:[F,J0,d] -> [L,F,J,d] = bf16(modulation + e0)."""
    _gpu(modulation, e0)
    L, J, d = modulation.shape
    F, J0, _ = e0.shape
    if out is None:
        out = torch.empty((L, F, J, d), dtype=modulation.dtype, device=modulation.device)
    _lib.call("rtv_modulation_table", _ptr(modulation), _ptr(e0), _ptr(out), L, F, J, J0, d, _stream())
    return out


def sinusoidal_embedding(t, dim):
    _gpu(t)
    t = t.to(torch.float32).contiguous()
    out = torch.empty((t.numel(), dim), dtype=torch.bfloat16, device=t.device)
    _lib.call("rtv_sinusoidal_embedding", _ptr(t), _ptr(out), t.numel(), int(dim), _stream())
    return out


def patchify(x, gh, gw):
    """x:[C,F,2gh,2gw] -> [F*gh*gw, C*4]"""
    _gpu(x)
    C, F = x.shape[0], x.shape[1]
    x = x.contiguous()
    out = torch.empty((F * gh * gw, C * 4), dtype=x.dtype, device=x.device)
    _lib.call("rtv_patchify", _ptr(x), _ptr(out), C, F, gh, gw, _stream())
    return out


def unpatchify(rows, C, F, gh, gw):
    _gpu(rows)
    rows = rows.contiguous()
    out = torch.empty((C, F, 2 * gh, 2 * gw), dtype=rows.dtype, device=rows.device)
    _lib.call("rtv_unpatchify", _ptr(rows), _ptr(out), C, F, gh, gw, _stream())
    return out


_T_KIND = {torch.float32: 0, torch.float64: 1, torch.int64: 2}


def _frame_view(x, what):
    """[F, C, h, w] (any frame / channel strides, contiguous h*w plane) -> (tensor, frame stride, channel stride)."""
    if x.dtype != torch.bfloat16:
        raise NotImplementedError(f"scheduler_step: {what} must be bf16 (the reference's inference dtype), got {x.dtype}")
    F, C, h, w = x.shape
    if w > 1 and x.stride(3) != 1 or h > 1 and x.stride(2) != w:
        x = x.contiguous()
    return x, x.stride(0), x.stride(1)


def scheduler_step(timesteps, sigmas, flow=None, xt=None, t=None, x0=None, noise=None, t_next=None):
    """One launch of the flow-matching step arithmetic (include/rtv_hip.h: rtv_scheduler_step) on [F, C, h, w] latents.
    flow/xt/t -> x0 (wan_wrapper.py:181-205); x0 (computed or given)/noise/t_next -> noisy (scheduler.py:159-176).
    Returns (x0, noisy or None)."""
    src = flow if flow is not None else x0
    _gpu(src, timesteps, sigmas)
    F, C, h, w = src.shape
    if timesteps.dtype != torch.float32 or sigmas.dtype != torch.float32 or timesteps.numel() != sigmas.numel():
        raise ValueError("scheduler_step: timesteps / sigmas must be float32 tables of one length")
    tk = None
    for tt in (t, t_next):
        if tt is not None:
            _gpu(tt)
            if tt.dtype not in _T_KIND or tt.numel() != F or not tt.is_contiguous():
                raise ValueError(f"scheduler_step: timestep must be a contiguous [F] float32/float64/int64 tensor, got "
                                 f"{tt.dtype} {tuple(tt.shape)}")
            if tk is not None and tk != _T_KIND[tt.dtype]:
                raise ValueError("scheduler_step: t and t_next must share a dtype")
            tk = _T_KIND[tt.dtype]
    fs = fc = xs = xc = 0
    if flow is not None:
        _gpu(xt)
        if xt.shape != flow.shape:
            raise ValueError(f"scheduler_step: flow {tuple(flow.shape)} vs xt {tuple(xt.shape)}")
        flow, fs, fc = _frame_view(flow, "flow")
        xt, xs, xc = _frame_view(xt, "xt")
        x0 = torch.empty((F, C, h, w), dtype=torch.bfloat16, device=src.device)
    else:
        if x0.dtype != torch.bfloat16:
            raise NotImplementedError("scheduler_step: x0 must be bf16")
        x0 = x0.contiguous()
    noisy = None
    if t_next is not None:
        _gpu(noise)
        if noise.dtype != torch.bfloat16 or noise.shape != x0.shape:
            raise ValueError("scheduler_step: noise must be bf16 of the latent shape")
        noise = noise.contiguous()
        noisy = torch.empty_like(x0)
    _lib.call("rtv_scheduler_step", _ptr(flow) if flow is not None else None, fs, fc,
              _ptr(xt) if flow is not None else None, xs, xc, _ptr(t) if t is not None else None,
              _ptr(t_next) if t_next is not None else None, tk if tk is not None else 0,
              _ptr(timesteps), _ptr(sigmas), timesteps.numel(), _ptr(x0),
              _ptr(noise) if noise is not None else None, _ptr(noisy) if noisy is not None else None,
              F, C, h * w, _stream())
    return x0, noisy


def pixels_to_rgb8(pixels, out=None):
    """Decoder pixels float32 [..., 3, H, W] in [-1, 1] -> uint8 [..., H, W, 3] (rtv_pixels_to_rgb8): the reference's
    host-side normalisation + to_pil_image byte conversion (release_server.py:984, :972) done on the GPU."""
    _gpu(pixels)
    if pixels.dtype != torch.float32 or pixels.shape[-3] != 3 or not pixels.is_contiguous():
        raise ValueError("pixels_to_rgb8 expects contiguous float32 [..., 3, H, W]")
    H, W = pixels.shape[-2:]
    T = pixels.numel() // (3 * H * W)
    shape = tuple(pixels.shape[:-3]) + (H, W, 3)
    if out is None:
        out = torch.empty(shape, dtype=torch.uint8, device=pixels.device)
    elif out.shape != shape or out.dtype != torch.uint8 or not out.is_contiguous():
        raise ValueError("pixels_to_rgb8: out must be contiguous uint8 [..., H, W, 3]")
    _lib.call("rtv_pixels_to_rgb8", _ptr(pixels), _ptr(out), T, H, W, _stream())
    return out
