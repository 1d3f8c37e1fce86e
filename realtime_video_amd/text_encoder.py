"""Mirror of the reference's `WanTextEncoder` (utils/wan_wrapper.py:20-56): UMT5-XXL encoder (wan/modules/t5.py:267-313,
config :456-469) from prompts to `{"prompt_embeds": [B, 512, 4096]}` with the padding rows zeroed, backed by the native
`rtv_t5_encode` (include/rtv_hip.h; realtime_video_amd/csrc/t5_encoder.hip).  Runs once per prompt and once per prompt
transition (release_server.py:402-404 / interpolate_prompt_embeds), never inside the block loop.

State-dict keys are the reference's (`models_t5_umt5-xxl-enc-bf16.safetensors`, wan_wrapper.py:30-33):
`token_embedding.weight`, `blocks.N.{norm1,norm2}.weight`, `blocks.N.attn.{q,k,v,o}.weight`,
`blocks.N.ffn.{gate.0,fc1,fc2}.weight`, `blocks.N.pos_embedding.embedding.weight`, `norm.weight`.

The tokenizer (HuggingfaceTokenizer over google/umt5-xxl with whitespace cleaning, wan/modules/tokenizers.py:38-82) needs
vocabulary files: pass `tokenizer=` a callable `texts -> (ids[B, L], mask[B, L])` or a local directory for
`transformers.AutoTokenizer`; `encode_ids(ids, mask)` is the entry without one.
"""
import ctypes
import html
import math
import re

import torch

from . import _lib

c_vp = ctypes.c_void_p
c_int = ctypes.c_int

UMT5_XXL = dict(vocab=256384, dim=4096, dim_attn=4096, dim_ffn=10240, num_heads=64, num_layers=24, num_buckets=32)


class _T5Cfg(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("vocab", "dim", "dim_attn", "dim_ffn", "num_heads", "num_layers")] + [("eps", ctypes.c_float)]


class _T5Layer(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("norm1_w", "qk_w", "v_w", "o_w", "norm2_w", "gate_fc1_w", "fc2_w", "pos_bias")]


class _T5Weights(ctypes.Structure):
    _fields_ = [("token_embedding", c_vp), ("final_norm_w", c_vp), ("layers", ctypes.POINTER(_T5Layer)), ("max_len", c_int)]


_lib.EXTRA_SIGNATURES.update({
    "rtv_t5_encode": [ctypes.POINTER(_T5Cfg), ctypes.POINTER(_T5Weights), c_vp, c_int, c_int, c_vp, ctypes.c_size_t, c_vp, c_vp],
})


def relative_position_bucket(rel_pos, num_buckets=32, max_dist=128):
    """Bidirectional bucket of key - query (T5RelativeEmbedding._relative_position_bucket, t5.py:238-265): half the buckets per
    direction, exact below num_buckets/4, log-spaced up to max_dist."""
    nb = num_buckets // 2
    n = rel_pos.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_dist / max_exact) * (nb - max_exact)).long()
    large = torch.clamp(large, max=nb - 1)
    return (rel_pos > 0).long() * nb + torch.where(n < max_exact, n, large)


def whitespace_clean(text):
    """tokenizers.py:12-22 without ftfy.fix_text (ftfy is not available here; it only repairs mojibake)."""
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class WanTextEncoder:
    def __init__(self, tokenizer=None, device="cuda", text_len=512, **cfg):
        self.cfg = dict(UMT5_XXL)
        self.cfg.update(cfg)
        self.device = torch.device(device)
        self.text_len = text_len
        self._t = {}
        self._w = None
        self._ws = None
        if isinstance(tokenizer, str):
            from transformers import AutoTokenizer
            hf = AutoTokenizer.from_pretrained(tokenizer)

            def tokenizer(texts, _hf=hf):   # noqa: F811  (HuggingfaceTokenizer.__call__, tokenizers.py:50-71)
                enc = _hf([whitespace_clean(t) for t in texts], return_tensors="pt", padding="max_length", truncation=True,
                          max_length=self.text_len, add_special_tokens=True)
                return enc.input_ids, enc.attention_mask
        self.tokenizer = tokenizer

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def requires_grad_(self, flag=False):
        return self

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=True):
        c = self.cfg
        L, H = c["num_layers"], c["num_heads"]
        if c["dim_attn"] != 64 * H:
            raise ValueError("the native encoder is built for head_dim 64 (UMT5-XXL)")
        dev = self.device

        def bf16(name):
            if name not in sd:
                raise KeyError(f"missing key {name}")
            t = sd[name]
            return t.to(device=dev, dtype=torch.bfloat16).contiguous()

        t = {"token_embedding": bf16("token_embedding.weight"), "final_norm": bf16("norm.weight")}
        rel = torch.arange(-(self.text_len - 1), self.text_len)          # index r + text_len - 1
        bucket = relative_position_bucket(rel, c["num_buckets"]).to(dev)
        self._layers = (_T5Layer * L)()
        for i in range(L):
            p = f"blocks.{i}"
            t[p + ".norm1"], t[p + ".norm2"] = bf16(p + ".norm1.weight"), bf16(p + ".norm2.weight")
            t[p + ".qk"] = torch.cat([bf16(p + ".attn.q.weight"), bf16(p + ".attn.k.weight")]).contiguous()
            t[p + ".v"], t[p + ".o"] = bf16(p + ".attn.v.weight"), bf16(p + ".attn.o.weight")
            t[p + ".gate_fc1"] = torch.cat([bf16(p + ".ffn.gate.0.weight"), bf16(p + ".ffn.fc1.weight")]).contiguous()
            t[p + ".fc2"] = bf16(p + ".ffn.fc2.weight")
            emb = sd[p + ".pos_embedding.embedding.weight"].to(device=dev, dtype=torch.float32)       # [buckets, H]
            t[p + ".pos_bias"] = emb[bucket].t().contiguous()                                       # [H, 2*text_len-1]
            lw = self._layers[i]
            for f, k in (("norm1_w", ".norm1"), ("qk_w", ".qk"), ("v_w", ".v"), ("o_w", ".o"), ("norm2_w", ".norm2"),
                         ("gate_fc1_w", ".gate_fc1"), ("fc2_w", ".fc2"), ("pos_bias", ".pos_bias")):
                setattr(lw, f, t[p + k].data_ptr())
        if strict:
            known = {"token_embedding.weight", "norm.weight"} | {
                f"blocks.{i}.{s}" for i in range(L) for s in (
                    "norm1.weight", "norm2.weight", "attn.q.weight", "attn.k.weight", "attn.v.weight", "attn.o.weight",
                    "ffn.gate.0.weight", "ffn.fc1.weight", "ffn.fc2.weight", "pos_embedding.embedding.weight")}
            extra = set(sd) - known
            if extra:
                raise KeyError(f"unexpected keys: {sorted(extra)[:4]}")
        self._t = t
        self._cfg = _T5Cfg(c["vocab"], c["dim"], c["dim_attn"], c["dim_ffn"], H, L, 1e-6)
        self._w = _T5Weights(t["token_embedding"].data_ptr(), t["final_norm"].data_ptr(), self._layers, self.text_len)
        return self

    def init_random_weights(self, seed=0):
        """Random weights of this architecture generated on the GPU (there is no checkpoint offline), init_weights' scales
        (t5.py:27-45)."""
        c = self.cfg
        g = torch.Generator(device=self.device).manual_seed(seed)
        dim, da, dff, H = c["dim"], c["dim_attn"], c["dim_ffn"], c["num_heads"]

        def n(shape, std):
            return (torch.randn(*shape, generator=g, device=self.device) * std).to(torch.bfloat16)

        sd = {"token_embedding.weight": n((c["vocab"], dim), 1.0), "norm.weight": 1 + n((dim,), 0.1)}
        for i in range(c["num_layers"]):
            p = f"blocks.{i}"
            sd[p + ".norm1.weight"], sd[p + ".norm2.weight"] = 1 + n((dim,), 0.1), 1 + n((dim,), 0.1)
            sd[p + ".attn.q.weight"] = n((da, dim), (dim * (da // H)) ** -0.5)
            sd[p + ".attn.k.weight"], sd[p + ".attn.v.weight"] = n((da, dim), dim ** -0.5), n((da, dim), dim ** -0.5)
            sd[p + ".attn.o.weight"] = n((dim, da), da ** -0.5)
            sd[p + ".ffn.gate.0.weight"], sd[p + ".ffn.fc1.weight"] = n((dff, dim), dim ** -0.5), n((dff, dim), dim ** -0.5)
            sd[p + ".ffn.fc2.weight"] = n((dim, dff), dff ** -0.5)
            sd[p + ".pos_embedding.embedding.weight"] = n((c["num_buckets"], H), 0.5)
        return self.load_state_dict(sd)

    # ------------------------------------------------------------------ forward
    def encode_ids(self, ids, mask):
        """ids, mask: [B, L] (L <= text_len).  Returns {"prompt_embeds": float32 [B, L, dim]}: T5Encoder output on the rows
        of each prompt, zeros behind them (wan_wrapper.py:47-56)."""
        if self._w is None:
            raise RuntimeError("weights not loaded")
        if ids.dim() != 2 or ids.shape != mask.shape or ids.shape[1] > self.text_len:
            raise ValueError(f"ids / mask must be [B, L <= {self.text_len}]")
        if self.device.type != "cuda":
            raise RuntimeError("the text encoder runs on the GPU only (no CPU fallback)")
        B, L = ids.shape
        dim = self.cfg["dim"]
        out = torch.empty((B, L, dim), dtype=torch.float32, device=self.device)
        lens = mask.gt(0).sum(dim=1).tolist()
        _lib.load().rtv_t5_workspace_bytes.restype = ctypes.c_size_t
        _lib.load().rtv_t5_workspace_bytes.argtypes = [ctypes.POINTER(_T5Cfg), c_int]
        stream = c_vp(torch.cuda.current_stream().cuda_stream)
        for b in range(B):
            n = int(lens[b])
            if n <= 0:
                out[b].zero_()
                continue
            if not bool(mask[b, :n].all()):
                raise ValueError("attention masks must be prefixes (tokens first, padding after)")
            need = _lib.load().rtv_t5_workspace_bytes(ctypes.byref(self._cfg), n)
            if self._ws is None or self._ws.numel() < need + 256:
                self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            ws_ptr = (self._ws.data_ptr() + 255) & ~255
            row = ids[b, :n].to(device=self.device, dtype=torch.int32).contiguous()
            _lib.call("rtv_t5_encode", ctypes.byref(self._cfg), ctypes.byref(self._w), c_vp(row.data_ptr()), n, L,
                      c_vp(ws_ptr), ctypes.c_size_t(self._ws.numel() - (ws_ptr - self._ws.data_ptr())),
                      c_vp(out[b].data_ptr()), stream)
        return {"prompt_embeds": out}

    def forward(self, text_prompts):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: pass tokenizer=<callable or local umt5-xxl directory>, or call encode_ids()")
        ids, mask = self.tokenizer(list(text_prompts))
        return self.encode_ids(ids, mask)

    __call__ = forward
