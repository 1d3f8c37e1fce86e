"""ORACLE (test infrastructure only; never imported by the product path): CPU restatement of the reference's text encoder -
the UMT5-XXL *encoder* of wan/modules/t5.py as `WanTextEncoder` drives it (utils/wan_wrapper.py:20-56).

Pinned against a golden minted from the reference's own `T5Encoder` module (oracle/make_golden.py `t5` ->
tests/golden/t5_encoder.pt, checked by tests/test_oracle_vs_golden.py).  The tokenizer (HuggingfaceTokenizer over
google/umt5-xxl, wan/modules/tokenizers.py:38-82) needs vocabulary files that are not in the reference tree: the
restatement starts at token ids + attention mask.

The reference runs this model in float32 (wan_wrapper.py:24-29) with weights that come from a bf16 checkpoint
(`models_t5_umt5-xxl-enc-bf16.safetensors`, :31), i.e. every weight is bf16-representable.
"""
import math

import torch
import torch.nn.functional as F

UMT5_XXL = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)  # t5.py:456-469
TINY_T5 = dict(vocab=100, dim=256, dim_attn=128, dim_ffn=512, num_heads=2, num_layers=2, num_buckets=32)           # head_dim 64 as in XXL


def relative_position_bucket(rel_pos, num_buckets=32, max_dist=128):
    """T5RelativeEmbedding._relative_position_bucket, bidirectional branch (t5.py:238-265).  rel_pos = key - query."""
    nb = num_buckets // 2
    buckets = (rel_pos > 0).long() * nb
    n = torch.abs(rel_pos)
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(n < max_exact, n, large)


def relative_bias(embedding, lq, lk, num_buckets=32):
    """T5RelativeEmbedding.forward (t5.py:225-236): embedding [num_buckets, H] -> bias [1, H, lq, lk]."""
    rel = torch.arange(lk).unsqueeze(0) - torch.arange(lq).unsqueeze(1)
    e = F.embedding(relative_position_bucket(rel, num_buckets), embedding)
    return e.permute(2, 0, 1).unsqueeze(0).contiguous()


def layer_norm(x, weight, eps=1e-6):
    """T5LayerNorm.forward (t5.py:62-67): RMS normalisation, no centring, no bias."""
    x = x * torch.rsqrt(x.float().pow(2).mean(dim=-1, keepdim=True) + eps)
    if weight.dtype in (torch.float16, torch.bfloat16):
        x = x.type_as(weight)
    return weight * x


def gelu_tanh(x):
    """t5.py:48-52."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def attention(w, pre, x, mask, pos_bias, num_heads):
    """T5Attention.forward, self-attention use (t5.py:96-135): no 1/sqrt(d) scaling; the key mask REPLACES the bias of a
    masked key by finfo.min; softmax in float32."""
    b, L = x.shape[:2]
    q = F.linear(x, w[pre + ".q.weight"]).view(b, L, num_heads, -1)
    k = F.linear(x, w[pre + ".k.weight"]).view(b, L, num_heads, -1)
    v = F.linear(x, w[pre + ".v.weight"]).view(b, L, num_heads, -1)
    bias = x.new_zeros(b, num_heads, L, L)
    if pos_bias is not None:
        bias = bias + pos_bias
    if mask is not None:
        bias = bias.masked_fill(mask.view(b, 1, 1, -1) == 0, torch.finfo(x.dtype).min)
    attn = torch.einsum("binc,bjnc->bnij", q, k) + bias
    attn = F.softmax(attn.float(), dim=-1).type_as(attn)
    out = torch.einsum("bnij,bjnc->binc", attn, v).reshape(b, L, -1)
    return F.linear(out, w[pre + ".o.weight"])


def feed_forward(w, pre, x):
    """T5FeedForward.forward (t5.py:151-157): fc2(fc1(x) * GELU(gate(x)))."""
    return F.linear(F.linear(x, w[pre + ".fc1.weight"]) * gelu_tanh(F.linear(x, w[pre + ".gate.0.weight"])),
                    w[pre + ".fc2.weight"])


def encoder(w, ids, mask, cfg):
    """T5Encoder.forward with shared_pos=False (t5.py:293-302; block :172-178; dropout is identity in eval mode and
    fp16_clamp is the identity for float32 / bfloat16)."""
    x = F.embedding(ids, w["token_embedding.weight"])
    L = x.shape[1]
    for i in range(cfg["num_layers"]):
        pre = f"blocks.{i}"
        e = relative_bias(w[pre + ".pos_embedding.embedding.weight"], L, L, cfg["num_buckets"])
        x = x + attention(w, pre + ".attn", layer_norm(x, w[pre + ".norm1.weight"]), mask, e, cfg["num_heads"])
        x = x + feed_forward(w, pre + ".ffn", layer_norm(x, w[pre + ".norm2.weight"]))
    return layer_norm(x, w["norm.weight"])


def text_encoder_forward(w, ids, mask, cfg):
    """WanTextEncoder.forward after the tokenizer (wan_wrapper.py:47-56): encoder, then rows at and beyond each prompt's
    length are set to zero.  Returns {"prompt_embeds": [B, L, dim]}."""
    seq_lens = mask.gt(0).sum(dim=1).long()
    context = encoder(w, ids, mask, cfg)
    for u, v in zip(context, seq_lens):
        u[v:] = 0.0
    return {"prompt_embeds": context}


def make_t5_weights(cfg, seed=0):
    """Random encoder weights under the reference's state_dict names and init scheme (init_weights, t5.py:27-45), norm
    weights near 1 instead of exactly 1 so that they are exercised; rounded to bf16 like the released checkpoint and held
    in float32 like the reference's model."""
    g = torch.Generator().manual_seed(seed)
    dim, da, dff, H, nb = cfg["dim"], cfg["dim_attn"], cfg["dim_ffn"], cfg["num_heads"], cfg["num_buckets"]

    def n(shape, std):
        return (torch.randn(*shape, generator=g) * std).to(torch.bfloat16).float()

    w = {"token_embedding.weight": n((cfg["vocab"], dim), 1.0), "norm.weight": 1 + n((dim,), 0.1)}
    for i in range(cfg["num_layers"]):
        p = f"blocks.{i}"
        w[p + ".norm1.weight"] = 1 + n((dim,), 0.1)
        w[p + ".norm2.weight"] = 1 + n((dim,), 0.1)
        w[p + ".attn.q.weight"] = n((da, dim), (dim * (da // H)) ** -0.5)
        w[p + ".attn.k.weight"] = n((da, dim), dim ** -0.5)
        w[p + ".attn.v.weight"] = n((da, dim), dim ** -0.5)
        w[p + ".attn.o.weight"] = n((dim, da), da ** -0.5)
        w[p + ".ffn.gate.0.weight"] = n((dff, dim), dim ** -0.5)
        w[p + ".ffn.fc1.weight"] = n((dff, dim), dim ** -0.5)
        w[p + ".ffn.fc2.weight"] = n((dim, dff), dff ** -0.5)
        w[p + ".pos_embedding.embedding.weight"] = n((nb, H), 0.5)
    return w


def t5_inputs(cfg, seed=7, L=48, lens=(29, 48)):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(2, cfg["vocab"], (len(lens), L), generator=g)
    mask = torch.zeros(len(lens), L, dtype=torch.long)
    for b, n_ in enumerate(lens):
        mask[b, :n_] = 1
        ids[b, n_ - 1] = 1      # </s>
        ids[b, n_:] = 0         # <pad>
    return ids, mask
