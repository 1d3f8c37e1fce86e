"""Mirror of the reference's streaming `VAEEncoderWrapper` (demo_utils/vae_block3.py:116-175) and of
`encode_video_latent` (v2v.py:138-158):

    mu, feat_cache = encoder(frames[B,3,T,H,W] fp16 in [-1,1], feat_cache, stream=False)   # feat_cache: 55 x (Tensor | None)

backed by the native `rtv_vae_encode` (include/rtv_hip.h): the decoder's implicit-GEMM convolution kernel with a
stride on the output grid for the encoder's `Resample` downsampling (wan/modules/vae.py:84-96, :132-158).  On the T2V
path it re-encodes the first context frame once per block (release_server.py:572-575); in v2v / webcam mode it
encodes the incoming frames (release_server.py:489-527).  State-dict keys are the reference's (`encoder.*`, `conv1.*`
of Wan2.1_VAE.pth, release_server.py:199-201).  The 24 live cache slots returned are views into one arena allocation.
"""
import ctypes
import math

import torch

from . import _lib
from .vae_decoder import MEAN, STD, CacheArenas, _Attn, _Conv, _Res, pack_conv_weight

c_vp = ctypes.c_void_p
c_int = ctypes.c_int


class _VaeEncWeights(ctypes.Structure):
    _fields_ = [("conv1", _Conv), ("down", _Res * 8), ("resample", _Conv * 3), ("time_conv", _Conv * 2),
                ("mid0", _Res), ("mid2", _Res), ("attn", _Attn), ("head_gamma", c_vp), ("head", _Conv),
                ("conv1x1_w", c_vp), ("conv1x1_b", c_vp), ("mean", c_vp), ("std", c_vp)]


_lib.EXTRA_SIGNATURES.update({
    "rtv_vae_set_fresh_tap_skip": [c_int],      # include/rtv_hip_lab.h (A/B switch for the tests)
    "rtv_vae_encode": [ctypes.POINTER(_VaeEncWeights), c_vp] + [c_int] * 6 + [c_vp, ctypes.c_size_t, c_vp, c_int, c_int, c_vp],
    "rtv_vae_enc_cache_slot": [c_int, c_int, c_int, ctypes.POINTER(ctypes.c_size_t)] + [ctypes.POINTER(c_int)] * 4,
})

ENC_DIMS = (96, 96, 192, 384, 384)       # _video_vae: dim 96, dim_mult [1,2,4,4] (wan/modules/vae.py:591-598)
RES_LAYERS = (0, 1, 3, 4, 6, 7, 9, 10)   # encoder.downsamples.N that are ResidualBlocks
RESAMPLE_LAYERS = (2, 5, 8)              # downsample2d, downsample3d, downsample3d


class VAEEncoderWrapper:
    z_dim = 16

    def __init__(self, vae=None, device="cuda"):
        """`vae`: optional object with `.model.state_dict()` (the reference passes its WanVAE, vae_block3.py:117-120)."""
        self.device = torch.device(device)
        self._t = {}
        self._w = None
        self._arenas = CacheArenas()
        if vae is not None:
            self.load_state_dict(vae.model.state_dict())

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def half(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=True):
        dev, f16 = self.device, torch.float16
        t = {}
        W = _VaeEncWeights()

        def dv(x):
            return x.to(dev).contiguous()

        def conv(dst, name, cin_pad=None):
            wt = dv(pack_conv_weight(sd[name + ".weight"], cin_pad))
            b = dv(sd[name + ".bias"].detach().to(f16))
            t[name + ".w"], t[name + ".b"] = wt, b
            dst.w, dst.b = wt.data_ptr(), b.data_ptr()

        def gam(name):
            g = dv(sd[name].detach().to(f16).reshape(-1))
            t[name] = g
            return g.data_ptr()

        def res(dst, pre):
            dst.gamma0 = gam(pre + ".residual.0.gamma")
            conv(dst.conv_a, pre + ".residual.2")
            dst.gamma3 = gam(pre + ".residual.3.gamma")
            conv(dst.conv_b, pre + ".residual.6")
            if pre + ".shortcut.weight" in sd:
                conv(dst.shortcut, pre + ".shortcut")
            else:
                dst.shortcut.w, dst.shortcut.b = None, None

        conv(W.conv1, "encoder.conv1", cin_pad=32)
        for i, li in enumerate(RES_LAYERS):
            res(W.down[i], f"encoder.downsamples.{li}")
        for i, li in enumerate(RESAMPLE_LAYERS):
            conv(W.resample[i], f"encoder.downsamples.{li}.resample.1")
            if i > 0:
                conv(W.time_conv[i - 1], f"encoder.downsamples.{li}.time_conv")
        res(W.mid0, "encoder.middle.0")
        res(W.mid2, "encoder.middle.2")
        C = 384
        pre = "encoder.middle.1"   # single-head attention: fold the softmax scale into the query projection
        W.attn.gamma = gam(pre + ".norm.gamma")
        qkv_w = sd[pre + ".to_qkv.weight"].detach().to(f16).float().reshape(3 * C, C)
        qkv_b = sd[pre + ".to_qkv.bias"].detach().to(f16).float()
        sc = 1.0 / math.sqrt(C)
        parts = {"wq": qkv_w[:C] * sc, "bq": qkv_b[:C] * sc, "wk": qkv_w[C:2 * C], "bk": qkv_b[C:2 * C],
                 "wv": qkv_w[2 * C:], "bv": qkv_b[2 * C:],
                 "wproj": sd[pre + ".proj.weight"].detach().float().reshape(C, C), "bproj": sd[pre + ".proj.bias"].detach().float()}
        for k, v in parts.items():
            t["attn." + k] = dv(v.to(f16))
            setattr(W.attn, k, t["attn." + k].data_ptr())
        W.head_gamma = gam("encoder.head.0.gamma")
        conv(W.head, "encoder.head.2")
        t["conv1x1_w"] = dv(sd["conv1.weight"].detach().to(f16).float().reshape(32, 32))
        t["conv1x1_b"] = dv(sd["conv1.bias"].detach().to(f16).float())
        for nm, val in (("mean", MEAN), ("std", STD)):
            t[nm] = torch.tensor(val, dtype=torch.float32, device=dev)
        W.conv1x1_w, W.conv1x1_b = t["conv1x1_w"].data_ptr(), t["conv1x1_b"].data_ptr()
        W.mean, W.std = t["mean"].data_ptr(), t["std"].data_ptr()
        self._t, self._w = t, W
        return [], []

    @staticmethod
    def state_dict_spec():
        """(name, shape) of every encoder-side tensor of the reference VAE state_dict (Encoder3d, vae.py:254-305, plus
        WanVAE_.conv1, vae.py:479), in module order."""
        spec = [("encoder.conv1.weight", (96, 3, 3, 3, 3)), ("encoder.conv1.bias", (96,))]

        def res(pre, cin, cout):
            out = [(pre + ".residual.0.gamma", (cin, 1, 1, 1)), (pre + ".residual.2.weight", (cout, cin, 3, 3, 3)),
                   (pre + ".residual.2.bias", (cout,)), (pre + ".residual.3.gamma", (cout, 1, 1, 1)),
                   (pre + ".residual.6.weight", (cout, cout, 3, 3, 3)), (pre + ".residual.6.bias", (cout,))]
            if cin != cout:
                out += [(pre + ".shortcut.weight", (cout, cin, 1, 1, 1)), (pre + ".shortcut.bias", (cout,))]
            return out

        li = 0
        for s in range(4):
            cin, cout = ENC_DIMS[s], ENC_DIMS[s + 1]
            for _ in range(2):
                spec += res(f"encoder.downsamples.{li}", cin, cout)
                cin = cout
                li += 1
            if s != 3:
                pre = f"encoder.downsamples.{li}"
                spec += [(pre + ".resample.1.weight", (cout, cout, 3, 3)), (pre + ".resample.1.bias", (cout,))]
                if s > 0:
                    spec += [(pre + ".time_conv.weight", (cout, cout, 3, 1, 1)), (pre + ".time_conv.bias", (cout,))]
                li += 1
        spec += res("encoder.middle.0", 384, 384)
        spec += [("encoder.middle.1.norm.gamma", (384, 1, 1)), ("encoder.middle.1.to_qkv.weight", (1152, 384, 1, 1)),
                 ("encoder.middle.1.to_qkv.bias", (1152,)), ("encoder.middle.1.proj.weight", (384, 384, 1, 1)),
                 ("encoder.middle.1.proj.bias", (384,))]
        spec += res("encoder.middle.2", 384, 384)
        spec += [("encoder.head.0.gamma", (384, 1, 1, 1)), ("encoder.head.2.weight", (32, 384, 3, 3, 3)),
                 ("encoder.head.2.bias", (32,)), ("conv1.weight", (32, 32, 1, 1, 1)), ("conv1.bias", (32,))]
        return spec

    def init_random_weights(self, seed=1):
        """Synthetic encoder weights generated on the GPU (bench.py / smoke: no Wan2.1_VAE.pth offline)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        spec = dict(self.state_dict_spec())
        sd = {}
        for name, shape in spec.items():
            if name.endswith("gamma"):
                sd[name] = 1 + 0.1 * torch.randn(shape, generator=g, device=self.device)
            else:
                wshape = spec[name[:-5] + ".weight"] if name.endswith(".bias") else shape
                bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
                sd[name] = (torch.rand(shape, generator=g, device=self.device) * 2 - 1) * bound
        self.load_state_dict(sd)
        return self

    # ------------------------------------------------------------------ arena / cache views
    def _new_arena(self, H, W):
        lib = _lib.load()
        lib.rtv_vae_enc_arena_bytes.restype = ctypes.c_size_t
        lib.rtv_vae_enc_arena_bytes.argtypes = [c_int, c_int]
        n = lib.rtv_vae_enc_arena_bytes(H, W)
        if n == 0:
            raise ValueError(f"VAE encoder: frame size {H}x{W} not supported (multiples of 8, (H/8)(W/8) % 8 == 0)")
        return torch.zeros(n + 256, dtype=torch.uint8, device=self.device)

    def _cache_views(self, arena, base, H, W):
        views = [None] * 55
        off, C, h, w, ns = ctypes.c_size_t(0), c_int(0), c_int(0), c_int(0), c_int(0)
        for i in range(24):
            _lib.call("rtv_vae_enc_cache_slot", H, W, i, ctypes.byref(off), ctypes.byref(C), ctypes.byref(h), ctypes.byref(w),
                      ctypes.byref(ns))
            n = ns.value * h.value * w.value * C.value
            start = base + off.value
            v = arena[start:start + n * 2].view(torch.float16).view(ns.value, h.value, w.value, C.value)
            if i == 0:
                v = v[..., :3]
            views[i] = v.permute(3, 0, 1, 2).unsqueeze(0)   # [1, C, n, H, W] like the reference
        self._arenas.register(views, arena, base, (H, W))
        return views

    # ------------------------------------------------------------------ forward
    def forward(self, z, feat_cache, stream=False):
        """z: [1, 3, T, H, W] pixels in [-1, 1].  Time is cut like the reference (vae_block3.py:146-166): fresh cache ->
        frame 0 alone then 4-frame chunks from frame 1 (non-stream) / from frame 4i (stream); existing cache + stream ->
        4-frame chunks from frame 0."""
        if self._w is None:
            raise RuntimeError("weights not loaded")
        if not z.is_cuda:
            raise RuntimeError("realtime_video_amd.VAEEncoderWrapper needs GPU tensors (no CPU fallback)")
        B, Cc, T, H, W = z.shape
        if B != 1 or Cc != 3:
            raise NotImplementedError("VAE encoder: batch 1, RGB frames")
        frames = z[0].to(torch.float16).contiguous()
        fresh = feat_cache is None or len(feat_cache) == 0 or feat_cache[0] is None
        if fresh:
            # a stream whose cache list the caller has dropped (the session's one-shot re-encode does that every block) gives
            # its arena back: no allocation, no zero-fill - rtv_vae_encode(first=1) clears the cache slices a fresh stream reads
            rec = self._arenas.recycle((H, W))
            if rec is not None:
                arena, base = rec
            else:
                arena = self._new_arena(H, W)
                base = (-arena.data_ptr()) % 256
        else:
            if not isinstance(feat_cache, list):
                feat_cache = list(feat_cache)
            arena, base = self._arenas.lookup(feat_cache, (H, W), lambda: self._new_arena(H, W),
                                              lambda a, b: self._cache_views(a, b, H, W))
        iter_ = 1 + (T - 1) // 4
        chunks = []                       # (t0, tn, first)
        offset = 1
        have_cache = not fresh
        for i in range(iter_):
            if i == 0 and not have_cache:
                chunks.append((0, 1, True))
                have_cache = True
            else:
                slice_start = i - 1
                if stream:
                    offset, slice_start = 0, i
                t0 = offset + 4 * slice_start
                if t0 < 0 or t0 + 4 > T:
                    raise ValueError(f"VAE encoder: chunk [{t0}, {t0 + 4}) outside the {T}-frame clip "
                                     "(the reference would feed Encoder3d a short or empty slice here)")
                chunks.append((t0, 4, False))
        mu = torch.empty((16, len(chunks), H // 8, W // 8), dtype=torch.float16, device=z.device)
        for j, (t0, tn, first) in enumerate(chunks):
            _lib.call("rtv_vae_encode", ctypes.byref(self._w), c_vp(frames.data_ptr()), T, t0, tn, H, W, int(first),
                      c_vp(arena.data_ptr() + base), ctypes.c_size_t(arena.numel() - base), c_vp(mu.data_ptr()),
                      len(chunks), j, c_vp(torch.cuda.current_stream().cuda_stream))
        cache = self._cache_views(arena, base, H, W) if fresh else list(feat_cache)
        return mu.unsqueeze(0).to(z.dtype), cache

    __call__ = forward


def encode_video_latent(vae, encode_vae_cache, resample_to=16, max_frames=81, video_path_or_url=None, frames=None,
                        height=None, width=None, stream=False, dtype=torch.float16):
    """Mirror of v2v.py:138-158 for in-memory frames [T, 3, H, W] in [-1, 1] (file / URL decoding is outside the hot
    path).  Returns (latents [16, T', h, w], cache)."""
    if frames is None:
        raise NotImplementedError("encode_video_latent: pass decoded frames (video file / URL loading is out of scope)")
    if not frames.is_cuda:
        raise RuntimeError("encode_video_latent needs GPU frames (no CPU fallback)")
    h, w = (frames.shape[2:]) if (height is None and width is None) else (height, width)
    if max_frames is None:
        max_frames = 1 + ((frames.shape[0] - 1) // 4) * 4
    if max_frames:
        frames = frames[:max_frames]
    h, w = h // 8 * 8, w // 8 * 8
    if tuple(frames.shape[2:]) != (h, w):
        # torch plumbing: the reference resizes with F.interpolate(mode='bicubic') (v2v.py:153); identity at equal size
        frames = torch.nn.functional.interpolate(frames.float(), size=(h, w), mode="bicubic")
    frames = frames.transpose(0, 1).to(dtype)
    latents, encode_vae_cache = vae(frames.unsqueeze(0), encode_vae_cache, stream=stream)
    return latents.squeeze(0).to(dtype), encode_vae_cache
