"""CPU ORACLE (test infrastructure only) — restatement of the reference's causal Wan DiT hot path.

This file is a checker, not a product path: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.  It restates, op for op, the reference's eager PyTorch algorithm for
the per-denoising-step DiT forward with KV cache, so that it can run (a) in bf16 exactly like the
reference's CPU/eager path and (b) in fp32/fp64 as a "gold" graph to bound rounding error.

Parity pinning: the upstream repo has NO tests or golden vectors for this path, so
the oracle is pinned against the reference's own modules imported in the authoring container
(oracle/ref_shim.py + oracle/make_golden.py -> tests/golden/*.pt; tests/test_oracle_vs_golden.py).

Every function cites the reference file:line it follows (paths under the upstream repo root).
Weights are plain dicts keyed by the reference's state_dict names.
"""
import math

import torch
import torch.nn.functional as F

FRAME_SEQLEN = 1560  # hard-coded in the reference: causal_model.py:351, causal_inference.py:35


# ------------------------------------------------------------------------------------------------
# embeddings / tables
# ------------------------------------------------------------------------------------------------
def sinusoidal_embedding_1d(dim, position):
    """wan/modules/model.py:15-24 (float64; device-agnostic)."""
    assert dim % 2 == 0
    half = dim // 2
    dev = position.device                       # table arithmetic always on the host (same bits wherever the graph runs)
    position = position.cpu().type(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half, dtype=torch.float64).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1).to(dev)


def rope_params(max_seq_len, dim, theta=10000):
    """wan/modules/model.py:28-35 -> complex128 [max_seq_len, dim/2]."""
    assert dim % 2 == 0
    freqs = torch.outer(torch.arange(max_seq_len),
                        1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def rope_table(head_dim):
    """wan/modules/causal_model.py:637-645: [1024, head_dim/2] complex128 = cat(frame | h | w parts)."""
    d = head_dim
    return torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                      rope_params(1024, 2 * (d // 6))], dim=1)


def rope_apply(x, grid, freqs, start_frame=0):
    """causal_rope_apply (causal_model.py:143-171); start_frame=0 gives rope_apply (model.py:39-66).
    x: [B, S, H, hd]; grid = (F, h, w); computed in float64, cast back with .type_as(x)."""
    n, c = x.size(2), x.size(3) // 2
    freqs = freqs.to(x.device)
    parts = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    f, h, w = grid
    seq_len = f * h * w
    out = []
    for i in range(x.size(0)):
        x_i = torch.view_as_complex(x[i, :seq_len].to(torch.float64).reshape(seq_len, n, -1, 2))
        freqs_i = torch.cat([
            parts[0][start_frame:start_frame + f].view(f, 1, 1, -1).expand(f, h, w, -1),
            parts[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
            parts[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(seq_len, 1, -1)
        x_i = torch.view_as_real(x_i * freqs_i).flatten(2)
        x_i = torch.cat([x_i, x[i, seq_len:]])
        out.append(x_i)
    return torch.stack(out).type_as(x)


# ------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------
def rms_norm(x, weight, eps=1e-6):
    """WanRMSNorm.forward, model.py:77-85: fp32 math, .type_as(x), then * weight; over the FULL dim."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)).type_as(x) * weight


def layer_norm(x, eps=1e-6, weight=None, bias=None):
    """WanLayerNorm.forward, model.py:88-98."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps).type_as(x)


# ------------------------------------------------------------------------------------------------
# attention backends
# ------------------------------------------------------------------------------------------------
def attention_sdpa(q, k, v, dtype=torch.bfloat16):
    """SDPA fallback of attention(), wan/modules/attention.py:197-212 (BLHD in, BLHD contiguous out,
    result stays in `dtype`).  Pass dtype=None to keep the input dtype (gold fp32 runs)."""
    if dtype is not None:
        q, k, v = q.to(dtype), k.to(dtype), v.to(dtype)
    out = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2))
    return out.transpose(1, 2).contiguous()


def attention_math(q, k, v, kv_limit=None):
    """Definition-level attention in fp32: softmax(q k^T / sqrt(d)) v with optional per-query key-prefix
    limits (kv_limit[i] = number of leading keys query i may see).  BLHD."""
    qf, kf, vf = q.float().transpose(1, 2), k.float().transpose(1, 2), v.float().transpose(1, 2)
    s = qf @ kf.transpose(-1, -2) / math.sqrt(q.shape[-1])
    if kv_limit is not None:
        kv_idx = torch.arange(k.shape[1], device=s.device).view(1, 1, 1, -1)
        s = s.masked_fill(kv_idx >= kv_limit.to(s.device).view(1, 1, -1, 1), float("-inf"))
    return (torch.softmax(s, dim=-1) @ vf).transpose(1, 2).contiguous()


def block_causal_limits(total_len, block_len):
    """`ends` array of get_block_mask, causal_model.py:119-136 (without the 128-padding rows):
    query i attends keys < ends[i] = (i // block_len + 1) * block_len."""
    idx = torch.arange(total_len)
    return torch.clamp((idx // block_len + 1) * block_len, max=total_len)


def attention_block_causal(q, k, v, block_len):
    """flex_attention with the block mask of causal_model.py:108-141 / :339-348.  The reference pads
    q/k/v with zero rows to a multiple of 128; pad keys are masked for real queries (kv < ends[q])
    and pad query rows are dropped (:348), so the un-padded masked attention is equivalent."""
    lim = block_causal_limits(q.shape[1], block_len)
    return attention_math(q, k, v, kv_limit=lim).to(q.dtype)


# ------------------------------------------------------------------------------------------------
# KV cache manager (pipeline/causal_inference.py:279-339)
# ------------------------------------------------------------------------------------------------
def initialize_kv_cache(num_layers, batch_size, kv_cache_size, num_heads, head_dim, dtype, device="cpu"):
    """causal_inference.py:279-314 (fresh allocation branch)."""
    return [{"k": torch.zeros(batch_size, kv_cache_size, num_heads, head_dim, dtype=dtype, device=device),
             "v": torch.zeros(batch_size, kv_cache_size, num_heads, head_dim, dtype=dtype, device=device),
             "global_end_index": 0, "local_end_index": 0} for _ in range(num_layers)]


def reset_kv_cache(kv_cache):
    """causal_inference.py:296-302 (zero re-initialisation branch)."""
    for c in kv_cache:
        c["k"].zero_()
        c["v"].zero_()
        c["global_end_index"] = 0
        c["local_end_index"] = 0


def initialize_crossattn_cache(num_layers, batch_size, num_heads, head_dim, dtype, text_len=512, device="cpu"):
    """causal_inference.py:316-339."""
    return [{"k": torch.zeros(batch_size, text_len, num_heads, head_dim, dtype=dtype, device=device),
             "v": torch.zeros(batch_size, text_len, num_heads, head_dim, dtype=dtype, device=device),
             "is_init": False} for _ in range(num_layers)]


# ------------------------------------------------------------------------------------------------
# DiT block
# ------------------------------------------------------------------------------------------------
FP8_FLAG = "__fp8__"   # weights[FP8_FLAG] = True switches every nn.Linear to the fp8 restatement below
# weights[FP8_ROW_SHARDS] = (world, M): under sequence parallelism every rank quantises the activation tensor IT holds
# (M/world contiguous token rows) with its own dynamic per-tensor scale - torchao's dynamic quantisation runs inside each
# rank's nn.Linear and knows nothing of the other shards.  Applies to the linears fed with the M token rows; the time / text
# MLPs see whole (replicated) tensors.
FP8_ROW_SHARDS = "__fp8_row_shards__"
_FP8_MAX = 448.0


def _fp8_scale(t):
    """Per-tensor scale max|t| / 448 as torchao's choose-scale evaluates it on the GPU (`amax / 448.0` with a Python scalar
    = multiplication by the fp32 reciprocal), clamped away from zero."""
    inv = torch.tensor(1.0, dtype=torch.float32, device="cpu") / torch.tensor(_FP8_MAX, dtype=torch.float32, device="cpu")
    return t.detach().float().abs().max().clamp(min=1e-12) * inv.to(t.device)


def fp8_linear(x, weight, bias, shards=1):
    """nn.Linear under the reference's `enable_fp8` (release_server.py:179-182):
    torchao.quantize_(transformer, Float8DynamicActivationFloat8WeightConfig(granularity=PerTensor())).  torchao is not
    part of the reference tree (third-party, unpinned): this restates its published algorithm - dynamic per-tensor
    activation scale, static per-tensor weight scale, e4m3 (OCP, saturating at 448, round-to-nearest-even) operands,
    fp32 accumulation and bias inside torch._scaled_mm, output in the activation dtype.  PARITY UNPINNED for this mode:
    no golden from the real torchao can be minted offline.  shards > 1: the rows (dim -2) are quantised in that many
    contiguous chunks, each with its own scale (FP8_ROW_SHARDS)."""
    if shards > 1:
        return torch.cat([fp8_linear(c, weight, bias) for c in x.chunk(shards, dim=-2)], dim=-2)
    sx, sw = _fp8_scale(x), _fp8_scale(weight)
    xq = (x.float() / sx).clamp(-_FP8_MAX, _FP8_MAX).to(torch.float8_e4m3fn).float()
    wq = (weight.float() / sw).clamp(-_FP8_MAX, _FP8_MAX).to(torch.float8_e4m3fn).float()
    y = F.linear(xq, wq) * (sx * sw)
    if bias is not None:
        y = y + bias.float()
    return y.to(x.dtype)


def _fp8_shards(w, x):
    """`rows`: the token-row count of the sharded forwards - an int, or a collection of them (a session's denoise forwards hold
    4680 rows, its recompute forwards 4680 / 9360 / 14040 as the context grows)."""
    world, rows = w.get(FP8_ROW_SHARDS, (1, 0))
    rows = (rows,) if isinstance(rows, int) else tuple(rows)
    return world if x.dim() >= 2 and x.shape[-2] in rows else 1


def _linear(w, x, weight, bias):
    if w.get(FP8_FLAG):
        return fp8_linear(x, weight, bias, _fp8_shards(w, x))
    return F.linear(x, weight, bias)


def _lin(x, w, prefix):
    return _linear(w, x, w[prefix + ".weight"], w[prefix + ".bias"])


def _qkv(x, w, pre):
    """self_attn q / k / v projections.  In fp8 mode the reference quantises the FUSED to_qkv Linear (fuse_projections runs
    before quantize_, release_server.py:176-182): one weight scale over [3d, d], one activation quantisation."""
    if not w.get(FP8_FLAG):
        return _lin(x, w, pre + ".q"), _lin(x, w, pre + ".k"), _lin(x, w, pre + ".v")
    wq = torch.cat([w[pre + ".q.weight"], w[pre + ".k.weight"], w[pre + ".v.weight"]])
    bq = torch.cat([w[pre + ".q.bias"], w[pre + ".k.bias"], w[pre + ".v.bias"]])
    return fp8_linear(x, wq, bq, _fp8_shards(w, x)).chunk(3, dim=-1)


def self_attention(w, pre, x, grid, freqs, num_heads, kv_cache, current_start, recompute,
                   num_frame_per_block=3, local_attn_size=-1, sink_size=0, eps=1e-6, attn_fn=None):
    """CausalWanSelfAttention.forward, causal_model.py:218-397.
    recompute=True is the `block_mask is not None` branch (:305-348); otherwise the cached branch
    (:349-392) including the rolling eviction (:363-379).  Mutates kv_cache exactly like the reference."""
    b, s, n = x.shape[0], x.shape[1], num_heads
    d = x.shape[2] // n
    attn_fn = attn_fn or attention_sdpa
    q_, k_, v_ = _qkv(x, w, pre)
    q = rms_norm(q_, w[pre + ".norm_q.weight"], eps).view(b, s, n, d)
    k = rms_norm(k_, w[pre + ".norm_k.weight"], eps).view(b, s, n, d)
    v = v_.reshape(b, s, n, d)

    if recompute:
        rq = rope_apply(q, grid, freqs).type_as(v)
        rk = rope_apply(k, grid, freqs).type_as(v)
        local_end = rk.shape[1]
        kv_cache["k"][:, :local_end] = rk
        kv_cache["v"][:, :local_end] = v
        kv_cache["global_end_index"] = local_end
        kv_cache["local_end_index"] = local_end
        out = attention_block_causal(rq, rk, v, FRAME_SEQLEN * num_frame_per_block)
    else:
        frame_seqlen = FRAME_SEQLEN
        start_frame = current_start // frame_seqlen
        rq = rope_apply(q, grid, freqs, start_frame).type_as(v)
        rk = rope_apply(k, grid, freqs, start_frame).type_as(v)
        current_end = current_start + rq.shape[1]
        sink_tokens = sink_size * frame_seqlen
        kv_cache_size = kv_cache["k"].shape[1]
        num_new = rq.shape[1]
        max_attention_size = 32760 if local_attn_size == -1 else local_attn_size * 1560
        if local_attn_size != -1 and current_end > kv_cache["global_end_index"] and \
                num_new + kv_cache["local_end_index"] > kv_cache_size:
            evicted = num_new + kv_cache["local_end_index"] - kv_cache_size
            rolled = kv_cache["local_end_index"] - evicted - sink_tokens
            kv_cache["k"][:, sink_tokens:sink_tokens + rolled] = \
                kv_cache["k"][:, sink_tokens + evicted:sink_tokens + evicted + rolled].clone()
            kv_cache["v"][:, sink_tokens:sink_tokens + rolled] = \
                kv_cache["v"][:, sink_tokens + evicted:sink_tokens + evicted + rolled].clone()
            local_end = kv_cache["local_end_index"] + current_end - kv_cache["global_end_index"] - evicted
        else:
            local_end = kv_cache["local_end_index"] + current_end - kv_cache["global_end_index"]
        local_start = local_end - num_new
        kv_cache["k"][:, local_start:local_end] = rk
        kv_cache["v"][:, local_start:local_end] = v
        lo = max(0, local_end - max_attention_size)
        out = attn_fn(rq, kv_cache["k"][:, lo:local_end], kv_cache["v"][:, lo:local_end])
        kv_cache["global_end_index"] = current_end
        kv_cache["local_end_index"] = local_end
    return _lin(out.flatten(2), w, pre + ".o")


def cross_attention(w, pre, x, context, num_heads, crossattn_cache, eps=1e-6, attn_fn=None):
    """WanT2VCrossAttention.forward, model.py:171-228 (SDPA branch :216-223)."""
    b, n = x.size(0), num_heads
    d = x.shape[2] // n
    attn_fn = attn_fn or attention_sdpa
    q = rms_norm(_lin(x, w, pre + ".q"), w[pre + ".norm_q.weight"], eps).view(b, -1, n, d)
    if crossattn_cache is not None and crossattn_cache["is_init"]:
        k, v = crossattn_cache["k"], crossattn_cache["v"]
    else:
        k = rms_norm(_lin(context, w, pre + ".k"), w[pre + ".norm_k.weight"], eps).view(b, -1, n, d)
        v = _lin(context, w, pre + ".v").view(b, -1, n, d)
        if crossattn_cache is not None:
            crossattn_cache["is_init"] = True
            crossattn_cache["k"] = k
            crossattn_cache["v"] = v
    out = attn_fn(q, k, v)
    return _lin(out.flatten(2), w, pre + ".o")


def attention_block(w, pre, x, e, grid, freqs, context, num_heads, kv_cache, crossattn_cache,
                    current_start, recompute, eps=1e-6, attn_fn=None, **sa_kwargs):
    """CausalWanAttentionBlock.forward, causal_model.py:440-492.  e: [B, F, 6, C]."""
    num_frames, frame_seqlen = e.shape[1], x.shape[1] // e.shape[1]
    e = (w[pre + ".modulation"].unsqueeze(1) + e).chunk(6, dim=2)

    def per_frame(t):
        return t.unflatten(dim=1, sizes=(num_frames, frame_seqlen))

    y = self_attention(w, pre + ".self_attn", (per_frame(layer_norm(x, eps)) * (1 + e[1]) + e[0]).flatten(1, 2),
                       grid, freqs, num_heads, kv_cache, current_start, recompute, eps=eps, attn_fn=attn_fn,
                       **sa_kwargs)
    x = x + (per_frame(y) * e[2]).flatten(1, 2)
    x = x + cross_attention(w, pre + ".cross_attn",
                            layer_norm(x, eps, w[pre + ".norm3.weight"], w[pre + ".norm3.bias"]),
                            context, num_heads, crossattn_cache, eps, attn_fn=attn_fn)
    h = (per_frame(layer_norm(x, eps)) * (1 + e[4]) + e[3]).flatten(1, 2)
    y = _lin(F.gelu(_lin(h, w, pre + ".ffn.0"), approximate="tanh"), w, pre + ".ffn.2")
    x = x + (per_frame(y) * e[5]).flatten(1, 2)
    return x


def head(w, x, e, eps=1e-6):
    """CausalHead.forward, causal_model.py:512-523.  e: [B, F, 1, C]."""
    num_frames, frame_seqlen = e.shape[1], x.shape[1] // e.shape[1]
    e = (w["head.modulation"].unsqueeze(1) + e).chunk(2, dim=2)
    h = layer_norm(x, eps).unflatten(dim=1, sizes=(num_frames, frame_seqlen)) * (1 + e[1]) + e[0]
    return _linear(w, h, w["head.head.weight"], w["head.head.bias"])


def unpatchify(x, grid, out_dim=16, patch_size=(1, 2, 2)):
    """causal_model.py:1126-1149 for one sample: x [L, out_dim*4] -> [out_dim, F, 2h, 2w]."""
    u = x[:math.prod(grid)].view(*grid, *patch_size, out_dim)
    u = torch.einsum("fhwpqrc->cfphqwr", u)
    return u.reshape(out_dim, *[i * j for i, j in zip(grid, patch_size)])


# ------------------------------------------------------------------------------------------------
# model / wrapper
# ------------------------------------------------------------------------------------------------
def model_forward(w, cfg, x, t, context, kv_cache, crossattn_cache, current_start=0, recompute=False,
                  attn_fn=None):
    """CausalWanModel._forward_inference, causal_model.py:825-954 (B == 1).
    x: [B, 16, F, H, W]; t: [B, F]; context: list of [L_txt, text_dim]; returns [B, 16, F, H, W].
    cfg keys: dim, ffn_dim, num_heads, num_layers, freq_dim, text_len, eps (+ local_attn_size, sink_size).
    `recompute` mirrors `self.block_mask is not None` (release_server.py:611-632)."""
    dim, n_heads, eps = cfg["dim"], cfg["num_heads"], cfg.get("eps", 1e-6)
    freqs = rope_table(dim // n_heads)
    outs = []
    xs = [F.conv3d(u.unsqueeze(0), w["patch_embedding.weight"], w["patch_embedding.bias"], stride=(1, 2, 2))
          for u in x]
    grid = tuple(xs[0].shape[2:])
    xs = torch.cat([u.flatten(2).transpose(1, 2) for u in xs])
    e = sinusoidal_embedding_1d(cfg.get("freq_dim", 256), t.flatten()).type_as(xs)
    e = _linear(w, F.silu(_linear(w, e, w["time_embedding.0.weight"], w["time_embedding.0.bias"])),
                w["time_embedding.2.weight"], w["time_embedding.2.bias"])
    e0 = _linear(w, F.silu(e), w["time_projection.1.weight"], w["time_projection.1.bias"]) \
        .unflatten(1, (6, dim)).unflatten(dim=0, sizes=t.shape)
    text_len = cfg.get("text_len", 512)
    ctx = torch.stack([torch.cat([u, u.new_zeros(text_len - u.size(0), u.size(1))]) for u in context])
    ctx = _linear(w, F.gelu(_linear(w, ctx, w["text_embedding.0.weight"], w["text_embedding.0.bias"]),
                            approximate="tanh"), w["text_embedding.2.weight"], w["text_embedding.2.bias"])
    h = xs
    for i in range(cfg["num_layers"]):
        h = attention_block(w, f"blocks.{i}", h, e0, grid, freqs, ctx, n_heads, kv_cache[i],
                            crossattn_cache[i], current_start, recompute, eps, attn_fn=attn_fn,
                            local_attn_size=cfg.get("local_attn_size", -1), sink_size=cfg.get("sink_size", 0),
                            num_frame_per_block=cfg.get("num_frame_per_block", 3))
    h = head(w, h, e.unflatten(dim=0, sizes=t.shape).unsqueeze(2), eps)
    for u in h:
        outs.append(unpatchify(u, grid))
    return torch.stack(outs)


class FlowMatchScheduler:
    """utils/scheduler.py:106-176 (set_timesteps :118-141 without the training weights; add_noise :159-176)."""

    def __init__(self, shift=5.0, sigma_min=0.0, extra_one_step=True, num_train_timesteps=1000, sigma_max=1.0):
        self.shift, self.sigma_min, self.sigma_max = shift, sigma_min, sigma_max
        self.extra_one_step, self.num_train_timesteps = extra_one_step, num_train_timesteps
        self.set_timesteps(1000)

    def set_timesteps(self, num_inference_steps=1000, denoising_strength=1.0):
        sigma_start = self.sigma_min + (self.sigma_max - self.sigma_min) * denoising_strength
        if self.extra_one_step:
            self.sigmas = torch.linspace(sigma_start, self.sigma_min, num_inference_steps + 1)[:-1]
        else:
            self.sigmas = torch.linspace(sigma_start, self.sigma_min, num_inference_steps)
        self.sigmas = self.shift * self.sigmas / (1 + (self.shift - 1) * self.sigmas)
        self.timesteps = self.sigmas * self.num_train_timesteps

    def add_noise(self, original_samples, noise, timestep):
        if timestep.ndim == 2:
            timestep = timestep.flatten(0, 1)
        dev = noise.device                          # tables live on the host; the arithmetic runs where the samples are
        timestep_id = torch.argmin((self.timesteps.to(dev).unsqueeze(0) - timestep.to(dev).unsqueeze(1)).abs(), dim=1)
        sigma = self.sigmas.to(dev)[timestep_id].reshape(-1, 1, 1, 1)
        return ((1 - sigma) * original_samples + sigma * noise).type_as(noise)


def convert_flow_pred_to_x0(scheduler, flow_pred, xt, timestep):
    """WanDiffusionWrapper._convert_flow_pred_to_x0, utils/wan_wrapper.py:181-205 (float64)."""
    original_dtype = flow_pred.dtype
    dev = flow_pred.device
    flow_pred, xt, sigmas, timesteps = (a.to(dev).double() for a in (flow_pred, xt, scheduler.sigmas, scheduler.timesteps))
    timestep_id = torch.argmin((timesteps.unsqueeze(0) - timestep.to(dev).unsqueeze(1)).abs(), dim=1)
    sigma_t = sigmas[timestep_id].reshape(-1, 1, 1, 1)
    return (xt - sigma_t * flow_pred).to(original_dtype)


def wrapper_forward(w, cfg, scheduler, noisy, prompt_embeds, timestep, kv_cache, crossattn_cache,
                    current_start, recompute=False, attn_fn=None):
    """WanDiffusionWrapper.forward (kv_cache branch), utils/wan_wrapper.py:230-301.
    noisy: [B, F, 16, H, W]; timestep: [B, F]; returns (flow_pred, pred_x0) both [B, F, 16, H, W]."""
    flow = model_forward(w, cfg, noisy.permute(0, 2, 1, 3, 4), timestep, prompt_embeds, kv_cache,
                         crossattn_cache, current_start, recompute, attn_fn).permute(0, 2, 1, 3, 4)
    x0 = convert_flow_pred_to_x0(scheduler, flow.flatten(0, 1), noisy.flatten(0, 1),
                                 timestep.flatten(0, 1)).unflatten(0, flow.shape[:2])
    return flow, x0


def get_denoising_schedule(timesteps, denoising_strength, steps=4):
    """v2v.py:133-136.  `timesteps` = scheduler.timesteps padded with a trailing 0 (release_server.py:559-560)."""
    lst = torch.linspace(denoising_strength * 1000, 0, steps, dtype=torch.float32).to(torch.long)
    return timesteps[1000 - lst]


# ------------------------------------------------------------------------------------------------
# block loop (GenerationSession), DiT part only
# ------------------------------------------------------------------------------------------------
class SessionOracle:
    """DiT side of GenerationSession: recompute_kv_cache (release_server.py:588-633) and the denoising
    loop of generate_block_internal (:636-708) for T2V with keep_first_frame semantics selectable.
    The VAE decode / first-frame re-encode legs are separate (oracle/vae_oracle.py)."""

    def __init__(self, w, cfg, prompt_embeds, noise, kv_cache_num_frames=3, num_steps=4, shift=5.0,
                 seed=0, attn_fn=None, first_frame_fn=None, strength=1.0, webcam_encoder=None, num_blocks=None):
        """`webcam_encoder`: callable with VAEEncoderWrapper's contract; passing it switches the session to webcam (streaming
        v2v) mode (release_server.py:489-527, :651-657).  `strength` only matters with an input video / webcam (:365-366)."""
        self.w, self.cfg, self.attn_fn = w, cfg, attn_fn
        self.webcam_encoder = webcam_encoder
        self.frame_queue = []                       # webcam frames [3, H, W] in [-1, 1] (:470-487 after decoding)
        self.encode_vae_cache = [None] * 55
        self.interpolated_prompt_embeds = []
        self.resume_latents = None
        self.randn_like = torch.randn_like          # the input-noising draw of webcam mode uses the global generator (:657)
        self.num_blocks = num_blocks if num_blocks is not None else noise.shape[1] // 3
        self.prompt_embeds = prompt_embeds          # list of [L_txt, text_dim]
        self.noise = noise                          # [1, num_blocks*3, 16, h, w]
        self.all_latents = torch.zeros_like(noise)
        self.c = kv_cache_num_frames
        self.nfpb = 3
        self.block_idx = 0
        self.current_start_frame = 0
        self.scheduler = FlowMatchScheduler(shift=shift, sigma_min=0.0, extra_one_step=True)
        zp = torch.cat((self.scheduler.timesteps, torch.tensor([0], dtype=torch.float32)))
        self.denoising_step_list = get_denoising_schedule(zp, strength, steps=num_steps)
        n_heads, hd = cfg["num_heads"], cfg["dim"] // cfg["num_heads"]
        kv_size = (self.c + self.nfpb) * FRAME_SEQLEN   # init_models, release_server.py:543-549
        self.dev = noise.device                         # the graph runs where its inputs live (host, or torch eager on a GPU)
        self.kv_cache = initialize_kv_cache(cfg["num_layers"], 1, kv_size, n_heads, hd, noise.dtype, self.dev)
        self.crossattn_cache = initialize_crossattn_cache(cfg["num_layers"], 1, n_heads, hd, noise.dtype,
                                                          cfg.get("text_len", 512), self.dev)
        self.rnd = torch.Generator().manual_seed(seed)
        self.first_frame_fn = first_frame_fn        # callable(block_idx) -> latent [1,1,16,h,w] (VAE re-encode)

    def clean_context_frames(self):
        """get_clean_context_frames, release_server.py:563-576."""
        ctx = self.all_latents[:, :self.current_start_frame]
        if self.first_frame_fn is None or (self.block_idx - 1) * self.nfpb < self.c:
            if self.c == 1:
                return ctx[:, :1]
            return torch.cat((ctx[:, :1], ctx[:, 1:][:, -self.c + 1:]), dim=1)
        tail = ctx[:, 1:][:, -self.c + 1:]
        return torch.cat((self.first_frame_fn(self.block_idx).to(tail), tail), dim=1)

    def recompute_kv_cache(self):
        """release_server.py:588-633."""
        if self.block_idx == 0:
            reset_kv_cache(self.kv_cache)
            if self.resume_latents is not None:     # :592-595: resume behind given latents (image-to-video start)
                self.current_start_frame = self.resume_latents.shape[1]
                self.all_latents[:, :self.current_start_frame] = self.resume_latents
            else:
                return self.current_start_frame
        start = min(self.current_start_frame, self.c)
        ctx = self.clean_context_frames()
        reset_kv_cache(self.kv_cache)
        t0 = torch.zeros([1, ctx.shape[1]], dtype=torch.int64, device=self.dev)
        wrapper_forward(self.w, self.cfg, self.scheduler, ctx, self.prompt_embeds, t0, self.kv_cache,
                        self.crossattn_cache, start * FRAME_SEQLEN, recompute=True, attn_fn=self.attn_fn)
        return start

    def generate_block(self):
        """generate_block_internal, release_server.py:636-708 (T2V branch), returns denoised latents."""
        if self.block_idx >= self.num_blocks:
            return None
        start = self.recompute_kv_cache()
        steps = self.denoising_step_list
        if self.webcam_encoder is not None:         # :651-657
            latents = self.process_webcam_frames(self.block_idx)
            if latents is None:
                return None
            s0 = steps[0] / 1000.0
            latents = latents[None].to(self.noise.dtype).movedim(1, 2)
            noisy = latents * (1.0 - s0) + self.randn_like(latents) * s0
        else:
            noisy = self.noise[:, self.current_start_frame:self.current_start_frame + self.nfpb]
        if self.interpolated_prompt_embeds:         # :659-666: prompt transition -> fresh cross-attention K/V
            for c in self.crossattn_cache:
                c["k"].zero_()
                c["v"].zero_()
                c["is_init"] = False
            self.prompt_embeds = [self.interpolated_prompt_embeds.pop(0)[0]]
        for index, current_timestep in enumerate(steps):
            timestep = (torch.ones([1, self.nfpb], dtype=torch.int64) * current_timestep).to(self.dev)  # -> float32 (trap 4)
            _, denoised = wrapper_forward(self.w, self.cfg, self.scheduler, noisy, self.prompt_embeds, timestep,
                                          self.kv_cache, self.crossattn_cache, start * FRAME_SEQLEN,
                                          attn_fn=self.attn_fn)
            if index < len(steps) - 1:
                nxt = steps[index + 1]
                eps_noise = torch.randn(*denoised.flatten(0, 1).shape, generator=self.rnd, dtype=torch.bfloat16) \
                    .to(denoised)
                noisy = self.scheduler.add_noise(denoised.flatten(0, 1), eps_noise,
                                                 nxt * torch.ones([self.nfpb], dtype=torch.long)) \
                    .unflatten(0, denoised.shape[:2])
        self.all_latents[:, self.current_start_frame:self.current_start_frame + self.nfpb] = denoised
        self.current_start_frame += self.nfpb
        self.block_idx += 1
        self.resume_latents = None
        return denoised

    # ---- input side (the VAE encoder leg is a callable with VAEEncoderWrapper's contract)
    def encode_video_latent(self, encoder, cache, frames, stream=False, max_frames=81):
        """v2v.py:138-158 for in-memory frames [T, 3, H, W] at the target size (the bicubic resize is the identity there).
        Returns (latents [16, T', h, w] fp16, cache)."""
        if max_frames is None:
            max_frames = 1 + ((frames.shape[0] - 1) // 4) * 4
        frames = frames[:max_frames].transpose(0, 1).to(torch.float16)
        lat, cache = encoder(frames.unsqueeze(0), cache, stream=stream)
        return lat.squeeze(0).to(torch.float16), cache

    def process_webcam_frames(self, idx):
        """release_server.py:489-527: 9 (block 0) or 12 queued frames, resampled by index, encoded on the running cache."""
        n = 9 if idx == 0 else 12
        if len(self.frame_queue) < n:
            return None
        frames, self.frame_queue = self.frame_queue, []
        lat, self.encode_vae_cache = self.encode_video_latent(self.webcam_encoder, self.encode_vae_cache,
                                                              torch.stack(resample_array(frames, n)), stream=idx > 0)
        return lat

    def interpolate_prompt_embeds(self, new_embeds, interpolation_steps):
        """release_server.py:459-468: lerp from the current to the new prompt embedding ([512, text_dim] each)."""
        e1, e2 = self.prompt_embeds[0][None], new_embeds[None].to(torch.bfloat16)
        x = torch.lerp(e1, e2, torch.linspace(0, 1, steps=interpolation_steps).unsqueeze(1).unsqueeze(2).to(e1))
        self.interpolated_prompt_embeds = list(x.chunk(interpolation_steps, dim=0))

    def setup_start_frame(self, image01, encoder):
        """release_server.py:578-586: image [3, H, W] in [0, 1] repeated over the pixel context window -> resume_latents."""
        n = 1 + (self.c - 1) * 4
        tensor = image01.to(torch.float16).sub(0.5).mul(2.0)
        lat, _ = self.encode_video_latent(encoder, [None] * 55, torch.stack([tensor] * n))
        self.resume_latents = lat.transpose(0, 1)[None]

    def setup_input_video(self, frames, encoder):
        """release_server.py:417-428: offline v2v - the video's latents noised to the first step's level replace the noise and
        bound the block count."""
        s0 = self.denoising_step_list[0] / 1000
        lat, _ = self.encode_video_latent(encoder, [None] * 55, frames, max_frames=None)
        lat = lat[None].to(self.noise.dtype).movedim(1, 2)
        eps = torch.randn(lat.shape, generator=self.rnd, dtype=self.noise.dtype).to(lat.device)
        self.noise = (lat * (1.0 - s0) + eps * s0).contiguous()
        self.num_blocks = min(lat.shape[1] // self.nfpb - 1, self.num_blocks)


def pipeline_inference(w, cfg, prompt_embeds, noise, initial_latent=None, denoising_step_list=(1000, 750, 500, 250),
                       warp_denoising_step=False, independent_first_frame=False, context_noise=0, shift=5.0, kv_size=32760,
                       randn_like=torch.randn_like, attn_fn=None):
    """CausalInferencePipeline.inference, pipeline/causal_inference.py:48-277 (DiT part; Step 4's VAE decode is separate):
    Step 2 caches the input frames at t = 0, Step 3 runs per block the denoising steps with re-noising between them and one
    forward at `context_noise` that writes the clean K/V.  Returns (latents [1, F_in + F, 16, h, w], kv_cache)."""
    nfpb = cfg.get("num_frame_per_block", 3)
    scheduler = FlowMatchScheduler(shift=shift, sigma_min=0.0, extra_one_step=True)
    steps = torch.tensor(list(denoising_step_list), dtype=torch.long)
    if warp_denoising_step:                                                    # :29-32
        steps = torch.cat((scheduler.timesteps, torch.tensor([0], dtype=torch.float32)))[1000 - steps]
    n_heads, hd = cfg["num_heads"], cfg["dim"] // cfg["num_heads"]
    dev = noise.device                      # the graph runs where its inputs live (host, or torch eager on a GPU)
    kv = initialize_kv_cache(cfg["num_layers"], 1, kv_size, n_heads, hd, noise.dtype, dev)
    ca = initialize_crossattn_cache(cfg["num_layers"], 1, n_heads, hd, noise.dtype, cfg.get("text_len", 512), dev)
    num_frames = noise.shape[1]
    n_in = initial_latent.shape[1] if initial_latent is not None else 0
    if not independent_first_frame or initial_latent is not None:
        num_blocks = num_frames // nfpb
    else:
        num_blocks = (num_frames - 1) // nfpb
    output = torch.zeros([1, num_frames + n_in] + list(noise.shape[2:]), dtype=noise.dtype, device=dev)
    start = 0

    def fwd(x, t):
        return wrapper_forward(w, cfg, scheduler, x, prompt_embeds, t.to(dev), kv, ca, start * FRAME_SEQLEN, attn_fn=attn_fn)

    if initial_latent is not None:                                             # Step 2, :136-168
        chunks = [1] if independent_first_frame else []
        chunks += [nfpb] * ((n_in - len(chunks)) // nfpb)
        for n in chunks:
            ref = initial_latent[:, start:start + n]
            output[:, start:start + n] = ref
            fwd(ref, torch.zeros([1, n], dtype=torch.int64))
            start += n
    all_num_frames = [nfpb] * num_blocks
    if independent_first_frame and initial_latent is None:
        all_num_frames = [1] + all_num_frames
    for cur in all_num_frames:                                                 # Step 3, :176-246
        noisy = noise[:, start - n_in:start + cur - n_in]
        for index, t_cur in enumerate(steps):
            timestep = torch.ones([1, cur], dtype=torch.int64) * t_cur
            _, denoised = fwd(noisy, timestep)
            if index < len(steps) - 1:
                flat = denoised.flatten(0, 1)
                noisy = scheduler.add_noise(flat, randn_like(flat), steps[index + 1] * torch.ones([cur], dtype=torch.long)) \
                    .unflatten(0, denoised.shape[:2])
        output[:, start:start + cur] = denoised
        fwd(denoised, torch.ones_like(timestep) * context_noise)
        start += cur
    return output, kv


def resample_array(array, target_length):
    """release_server.py:57-62: index resampling of a list by rounded linspace."""
    import numpy as np
    if len(array) == target_length:
        return array
    idx = np.round(np.linspace(0, len(array) - 1, target_length)).astype(int)
    return [array[i] for i in idx]


# ------------------------------------------------------------------------------------------------
# synthetic weights (SURVEY.md §8d "Synthetic inputs")
# ------------------------------------------------------------------------------------------------
def make_weights(cfg, seed=0, dtype=torch.bfloat16, text_dim=4096, in_dim=16, out_dim=16):
    """Random weights with the reference's state_dict names and init scheme (init_weights,
    causal_model.py:1151-1173: xavier-uniform Linear weights, zero biases, N(0,.02) text/time MLPs),
    except head.head.weight ~ N(0,.02) (zero-init in the reference would make every output 0) and small
    random biases / norm weights near 1 so that every term of the graph is exercised."""
    g = torch.Generator().manual_seed(seed)
    dim, ffn, L = cfg["dim"], cfg["ffn_dim"], cfg["num_layers"]
    freq_dim = cfg.get("freq_dim", 256)
    w = {}

    def xavier(out_f, in_f):
        a = math.sqrt(6.0 / (in_f + out_f))
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * a

    def normal(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std

    def lin(name, out_f, in_f, init="xavier"):
        w[name + ".weight"] = xavier(out_f, in_f) if init == "xavier" else normal(out_f, in_f)
        w[name + ".bias"] = normal(out_f, std=0.02)

    w["patch_embedding.weight"] = xavier(dim, in_dim * 4).view(dim, in_dim, 1, 2, 2)
    w["patch_embedding.bias"] = normal(dim)
    lin("text_embedding.0", dim, text_dim, "normal")
    lin("text_embedding.2", dim, dim, "normal")
    lin("time_embedding.0", dim, freq_dim, "normal")
    lin("time_embedding.2", dim, dim, "normal")
    lin("time_projection.1", dim * 6, dim)
    for i in range(L):
        p = f"blocks.{i}"
        for a in ("self_attn", "cross_attn"):
            for m in ("q", "k", "v", "o"):
                lin(f"{p}.{a}.{m}", dim, dim)
            w[f"{p}.{a}.norm_q.weight"] = 1 + normal(dim, std=0.1)
            w[f"{p}.{a}.norm_k.weight"] = 1 + normal(dim, std=0.1)
        w[f"{p}.norm3.weight"] = 1 + normal(dim, std=0.1)
        w[f"{p}.norm3.bias"] = normal(dim, std=0.05)
        lin(f"{p}.ffn.0", ffn, dim)
        lin(f"{p}.ffn.2", dim, ffn)
        w[f"{p}.modulation"] = torch.randn(1, 6, dim, generator=g) / dim ** 0.5
    w["head.head.weight"] = normal(out_dim * 4, dim)
    w["head.head.bias"] = normal(out_dim * 4)
    w["head.modulation"] = torch.randn(1, 2, dim, generator=g) / dim ** 0.5
    return {k: v.to(dtype) for k, v in w.items()}
