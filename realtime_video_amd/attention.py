"""Attention-backend plugin: drop-in for the reference's `wan.modules.attention.attention`.

Reference boundary: wan/modules/attention.py:150-165 (signature), :166-212 (Sage / FlashAttn / SDPA
dispatch) and the torch custom op `mylib::sageattn` (wan/modules/sage.py:12-19).  Here the backend is
the hand-written gfx950 kernel behind `rtv_attn_fwd` (include/rtv_hip.h); it is registered as the
torch custom op `rtv::attn_fwd` with a fake implementation so traced graphs survive, exactly like the
reference registers sageattention.
"""
import torch

from . import ops

__all__ = ["attention", "attn_op", "install", "RTV_ATTN_AVAILABLE"]

RTV_ATTN_AVAILABLE = True


@torch.library.custom_op("rtv::attn_fwd", mutates_args=(), device_types="cuda")
def attn_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, softmax_scale: float = -1.0,
            causal_block: int = 0, q_offset: int = 0) -> torch.Tensor:
    """q:[B,Lq,H,128], k/v:[B,Lkv,H,128] (BLHD; k/v may be strided KV-cache views) -> [B,Lq,H,128]."""
    return ops.attn_fwd(q, k, v, scale=None if softmax_scale <= 0 else softmax_scale,
                        causal_block=causal_block, q_offset=q_offset)


@attn_op.register_fake
def _attn_fake(q, k, v, softmax_scale=-1.0, causal_block=0, q_offset=0):
    return torch.empty(q.shape, device=q.device, dtype=q.dtype)


def attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None,
              causal=False, window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16,
              fa_version=None):
    """Same call contract as wan/modules/attention.py:150-165.  q:[B,Lq,H,D] k,v:[B,Lk,H,D] -> [B,Lq,H,D]
    contiguous in q's dtype (the Sage/FlashAttn branches cast back to the input dtype, :178, :147)."""
    # Padding lengths: the reference's FlashAttention branch drops keys at positions >= k_lens[b] (attention.py:91-99, :119-147) - its
    # SDPA fallback ignores them with a warning (:198-201).  The hot path passes None.  Here they are honoured exactly (a per-batch
    # key prefix) or refused - never silently ignored: q_lens other than the full length would need the
    # reference's packed output layout, which only exists for full lengths (:146 `unflatten(0, (b, lq))`).
    if q_lens is not None and any(int(n) != q.shape[1] for n in torch.as_tensor(q_lens).tolist()):
        raise NotImplementedError("q_lens shorter than the padded query length are not supported by the MI355X attention backend")
    key_prefix = None
    if k_lens is not None:
        key_prefix = [int(n) for n in torch.as_tensor(k_lens).tolist()]
        if len(key_prefix) != k.shape[0] or any(n <= 0 or n > k.shape[1] for n in key_prefix):
            raise ValueError(f"k_lens {key_prefix} does not fit keys of shape {tuple(k.shape)}")
        if all(n == k.shape[1] for n in key_prefix):
            key_prefix = None
    if dropout_p:
        raise NotImplementedError("attention dropout is not part of the inference hot path")
    if causal or tuple(window_size) != (-1, -1):
        raise NotImplementedError("token-causal / sliding-window masks are not used by the causal Wan path; "
                                  "use attn_op(..., causal_block=...) for the block-causal recompute mask")
    if not q.is_cuda:
        raise RuntimeError("realtime_video_amd attention backend needs GPU tensors (no CPU fallback)")
    og_dtype = q.dtype
    half = (torch.float16, torch.bfloat16)
    if q.dtype not in half or k.dtype != q.dtype or v.dtype != q.dtype:
        q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    if q_scale is not None:
        q = q * q_scale
    scale = -1.0 if softmax_scale is None else float(softmax_scale)
    if key_prefix is None:
        out = attn_op(q, k, v, scale)
    else:       # one launch per batch element over its own key prefix (strided views, no copies)
        out = torch.cat([attn_op(q[b:b + 1], k[b:b + 1, :n], v[b:b + 1, :n], scale) for b, n in enumerate(key_prefix)])
    return out if out.dtype == og_dtype else out.to(og_dtype)


def sageattn_func(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
    """Stand-in for `mylib::sageattn` (wan/modules/sage.py:12-19): BHLD in, BHLD out; used by the
    reference's cross-attention call sites (model.py:201-213)."""
    if attn_mask is not None or is_causal or dropout_p:
        raise NotImplementedError("mask / causal / dropout are not used by the hot path")
    out = attn_op(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return out.transpose(1, 2)


def install(attention_module=None, model_module=None, causal_model_module=None):
    """Plug this backend into an imported copy of the reference (the way SAGEATTN_AVAILABLE selects
    sageattention there): replaces `attention` / `sageattn_func` and raises the availability flag."""
    for mod in (attention_module, model_module, causal_model_module):
        if mod is None:
            continue
        if hasattr(mod, "attention"):
            mod.attention = attention
        if hasattr(mod, "sageattn_func"):
            mod.sageattn_func = sageattn_func
        if hasattr(mod, "SAGEATTN_AVAILABLE"):
            mod.SAGEATTN_AVAILABLE = True
