"""Mirror of the reference's streaming `VAEDecoderWrapper` (demo_utils/vae_block3.py:177-230):

    pixels, feat_cache = decoder(z[B,T,16,h,w] fp16, *feat_cache)     # feat_cache: 55 x (Tensor | None)

backed by the native `rtv_vae_decode` (include/rtv_hip.h): implicit-GEMM causal convolutions with the
feature-cache concat, nearest-2x upsampling, temporal-upsampling scatter and residual adds fused in.
State-dict keys are the reference's (`decoder.*`, `conv2.*`, loaded from Wan2.1_VAE.pth in
release_server.py:204-215).  The 32 live cache slots returned are *views* into one arena allocation
(channels-last in memory, exposed in the reference's [1, C, 2, H, W] shape); passing the list back
continues the stream, passing `[None] * 55` starts a new one.
"""
import ctypes
import math
import weakref

import torch

from . import _lib

c_vp = ctypes.c_void_p


class _Conv(ctypes.Structure):
    _fields_ = [("w", c_vp), ("b", c_vp)]


class _Res(ctypes.Structure):
    _fields_ = [("gamma0", c_vp), ("conv_a", _Conv), ("gamma3", c_vp), ("conv_b", _Conv), ("shortcut", _Conv)]


class _Attn(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in ("gamma", "wq", "bq", "wk", "bk", "wv", "bv", "wproj", "bproj")]


class _VaeWeights(ctypes.Structure):
    _fields_ = [("conv2_w", c_vp), ("conv2_b", c_vp), ("mean", c_vp), ("std", c_vp), ("conv1", _Conv),
                ("mid0", _Res), ("mid2", _Res), ("up", _Res * 12), ("attn", _Attn),
                ("time_conv", _Conv * 2), ("resample", _Conv * 3), ("head_gamma", c_vp), ("head", _Conv)]


_lib.EXTRA_SIGNATURES.update({
    "rtv_conv_cl": [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_vp, ctypes.c_int] + [ctypes.c_int] * 10 + [c_vp, c_vp],
    "rtv_rmsnorm_silu_cl": [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, c_vp],
    "rtv_conv3_norm_silu_cl": [c_vp, c_vp, c_vp, c_vp, c_vp, ctypes.c_int] + [ctypes.c_int] * 6 + [c_vp, c_vp],
    "rtv_conv_set_fuse_norm": [ctypes.c_int],
    "rtv_softmax_rows": [c_vp, ctypes.c_int, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp],
    "rtv_vae_decode": [ctypes.POINTER(_VaeWeights), c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                       c_vp, ctypes.c_size_t, c_vp, c_vp],
    "rtv_vae_cache_slot": [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                           ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)],
    "rtv_conv_cl_win": [c_vp, c_vp, c_vp, c_vp, ctypes.c_int, c_vp, ctypes.c_int] + [ctypes.c_int] * 10 + [c_vp]
                       + [ctypes.c_int] * 4 + [c_vp],
    "rtv_vae_decode_rows": [ctypes.POINTER(_VaeWeights), c_vp] + [ctypes.c_int] * 6 + [c_vp, ctypes.c_size_t, c_vp, c_vp],
    "rtv_vae_decode_single": [ctypes.POINTER(_VaeWeights), c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, ctypes.c_size_t,
                              c_vp, c_vp],
    "rtv_conv_set_halo": [ctypes.c_int],
    "rtv_vae_cache_slot_rows": [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_size_t)] + [ctypes.POINTER(ctypes.c_int)] * 4,
})

MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
        0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
       3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]


def pack_conv_weight(w, cin_pad=None, cout_pad=None):
    """torch conv weight [Cout, Cin, *k] -> fp16 [Cout][taps][Cin] (K-contiguous im2col order)."""
    w = w.detach().to(torch.float16)
    nd = w.dim() - 2
    perm = [0] + list(range(2, 2 + nd)) + [1]
    w = w.permute(*perm).contiguous()
    cout, cin = w.shape[0], w.shape[-1]
    w = w.reshape(cout, -1, cin)
    if cin_pad and cin_pad > cin:
        w = torch.cat([w, w.new_zeros(cout, w.shape[1], cin_pad - cin)], dim=2)
    if cout_pad and cout_pad > cout:
        w = torch.cat([w, w.new_zeros(cout_pad - cout, w.shape[1], w.shape[2])], dim=0)
    return w.contiguous()


class VAEDecoderWrapper:
    z_dim = 16

    def __init__(self, device="cuda", row_shard=None):
        """row_shard = (index, count): this instance decodes only the index-th of `count` horizontal stripes of every
        frame (spatially sharded decode of the context-parallel path; realtime_video_amd.parallel.ShardedVAEDecoder
        gathers the stripes).  None = whole frames, the reference behaviour."""
        self.device = torch.device(device)
        self._t = {}
        self._w = None
        self._arenas = CacheArenas()
        self.row_shard = row_shard

    def row_range(self, h):
        """Pixel rows [r0, r1) this instance produces for latent height h."""
        H = 8 * h
        if self.row_shard is None:
            return 0, H
        i, n = self.row_shard
        return H * i // n, H * (i + 1) // n

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def half(self):
        return self

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=True):
        dev, f16 = self.device, torch.float16
        t = {}
        W = _VaeWeights()

        def dv(x):
            return x.to(dev).contiguous()

        def conv(dst, name, cin_pad=None, cout_pad=None):
            wt = dv(pack_conv_weight(sd[name + ".weight"], cin_pad, cout_pad))
            b = sd[name + ".bias"].detach().to(f16)
            if cout_pad and cout_pad > b.numel():
                b = torch.cat([b, b.new_zeros(cout_pad - b.numel())])
            b = dv(b)
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

        for nm, val in (("mean", MEAN), ("std", STD)):
            t[nm] = torch.tensor(val, dtype=torch.float32, device=dev)
        t["conv2_w"] = dv(sd["conv2.weight"].detach().to(f16).float().reshape(16, 16))
        t["conv2_b"] = dv(sd["conv2.bias"].detach().to(f16).float())
        W.conv2_w, W.conv2_b = t["conv2_w"].data_ptr(), t["conv2_b"].data_ptr()
        W.mean, W.std = t["mean"].data_ptr(), t["std"].data_ptr()
        conv(W.conv1, "decoder.conv1", cin_pad=32)
        res(W.mid0, "decoder.middle.0")
        res(W.mid2, "decoder.middle.2")
        # attention (single head, C = 384): fold the softmax scale into the query projection
        C = 384
        pre = "decoder.middle.1"
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
        li = 0
        for s in range(4):
            for r in range(3):
                res(W.up[s * 3 + r], f"decoder.upsamples.{li}")
                li += 1
            if s != 3:
                if s < 2:
                    conv(W.time_conv[s], f"decoder.upsamples.{li}.time_conv")
                conv(W.resample[s], f"decoder.upsamples.{li}.resample.1")
                li += 1
        W.head_gamma = gam("decoder.head.0.gamma")
        conv(W.head, "decoder.head.2", cout_pad=8)
        self._t, self._w = t, W
        return [], []

    @staticmethod
    def state_dict_spec():
        """(name, shape) of every tensor of the reference decoder state_dict (vae_block3.py:334-384 built
        from wan/modules/vae.py blocks), in module order."""
        spec = [("conv2.weight", (16, 16, 1, 1, 1)), ("conv2.bias", (16,)),
                ("decoder.conv1.weight", (384, 16, 3, 3, 3)), ("decoder.conv1.bias", (384,))]

        def res(pre, cin, cout):
            out = [(pre + ".residual.0.gamma", (cin, 1, 1, 1)), (pre + ".residual.2.weight", (cout, cin, 3, 3, 3)),
                   (pre + ".residual.2.bias", (cout,)), (pre + ".residual.3.gamma", (cout, 1, 1, 1)),
                   (pre + ".residual.6.weight", (cout, cout, 3, 3, 3)), (pre + ".residual.6.bias", (cout,))]
            if cin != cout:
                out += [(pre + ".shortcut.weight", (cout, cin, 1, 1, 1)), (pre + ".shortcut.bias", (cout,))]
            return out

        spec += res("decoder.middle.0", 384, 384)
        spec += [("decoder.middle.1.norm.gamma", (384, 1, 1)), ("decoder.middle.1.to_qkv.weight", (1152, 384, 1, 1)),
                 ("decoder.middle.1.to_qkv.bias", (1152,)), ("decoder.middle.1.proj.weight", (384, 384, 1, 1)),
                 ("decoder.middle.1.proj.bias", (384,))]
        spec += res("decoder.middle.2", 384, 384)
        li, cin = 0, 384
        for s, cout in enumerate((384, 384, 192, 96)):
            if s > 0:
                cin //= 2
            for _ in range(3):
                spec += res(f"decoder.upsamples.{li}", cin, cout)
                cin = cout
                li += 1
            if s != 3:
                pre = f"decoder.upsamples.{li}"
                if s < 2:
                    spec += [(pre + ".time_conv.weight", (2 * cout, cout, 3, 1, 1)), (pre + ".time_conv.bias", (2 * cout,))]
                spec += [(pre + ".resample.1.weight", (cout // 2, cout, 3, 3)), (pre + ".resample.1.bias", (cout // 2,))]
                li += 1
        spec += [("decoder.head.0.gamma", (96, 1, 1, 1)), ("decoder.head.2.weight", (3, 96, 3, 3, 3)),
                 ("decoder.head.2.bias", (3,))]
        return spec

    def init_random_weights(self, seed=0):
        """Synthetic decoder weights generated on the GPU (bench.py / smoke: no Wan2.1_VAE.pth offline):
        PyTorch-default uniform(-1/sqrt(fan_in), +) convs, gammas near 1."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        sd = {}
        for name, shape in self.state_dict_spec():
            if name.endswith("gamma"):
                sd[name] = 1 + 0.1 * torch.randn(shape, generator=g, device=self.device)
            else:
                wname = name[:-5] + ".weight" if name.endswith(".bias") else name
                wshape = dict(self.state_dict_spec())[wname]
                bound = 1.0 / math.sqrt(math.prod(wshape[1:]))
                sd[name] = (torch.rand(shape, generator=g, device=self.device) * 2 - 1) * bound
        self.load_state_dict(sd)
        return self

    # ------------------------------------------------------------------ arena / cache views
    def _new_arena(self, h, w):
        lib = _lib.load()
        lib.rtv_vae_arena_bytes_rows.restype = ctypes.c_size_t
        lib.rtv_vae_arena_bytes_rows.argtypes = [ctypes.c_int] * 4
        n = lib.rtv_vae_arena_bytes_rows(h, w, *self.row_range(h))
        if n == 0:
            raise ValueError("VAE decoder: bad latent size / row range")
        return torch.zeros(n + 256, dtype=torch.uint8, device=self.device)

    def _cache_views(self, arena, base, h, w):
        views = [None] * 55
        off, C, H, Wd, fr = ctypes.c_size_t(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        r0, r1 = self.row_range(h)
        for i in range(32):
            _lib.call("rtv_vae_cache_slot_rows", h, w, r0, r1, i, ctypes.byref(off), ctypes.byref(C), ctypes.byref(H),
                      ctypes.byref(Wd), ctypes.byref(fr))
            n = 2 * H.value * Wd.value * C.value
            start = base + off.value
            flat = arena[start:start + n * 2].view(torch.float16)
            v = flat.view(2, H.value, Wd.value, C.value)
            if i == 0:
                v = v[..., :16]
            views[i] = v.permute(3, 0, 1, 2).unsqueeze(0)   # [1, C, 2, H, W] like the reference
        self._arenas.register(views, arena, base, (h, w))
        return views

    # ------------------------------------------------------------------ forward
    def forward(self, z, *feat_cache):
        if self._w is None:
            raise RuntimeError("weights not loaded")
        if not z.is_cuda:
            raise RuntimeError("realtime_video_amd.VAEDecoderWrapper needs GPU tensors (no CPU fallback)")
        B, T, C, h, w = z.shape
        if B != 1 or C != 16:
            raise NotImplementedError("VAE decoder: batch 1, 16 latent channels")
        zz = z[0].to(torch.float16).contiguous()
        first = len(feat_cache) == 0 or feat_cache[0] is None
        if first:
            arena = self._new_arena(h, w)
            base = (-arena.data_ptr()) % 256
        else:
            feat_cache = list(feat_cache)
            arena, base = self._arenas.lookup(feat_cache, (h, w), lambda: self._new_arena(h, w),
                                              lambda a, b: self._cache_views(a, b, h, w))
        n_out = 4 * T - 3 if first else 4 * T
        r0, r1 = self.row_range(h)
        pixels = torch.empty((n_out, 3, r1 - r0, 8 * w), dtype=torch.float32, device=z.device)
        _lib.call("rtv_vae_decode_rows", ctypes.byref(self._w), c_vp(zz.data_ptr()), T, h, w, int(first), r0, r1,
                  c_vp(arena.data_ptr() + base), ctypes.c_size_t(arena.numel() - base), c_vp(pixels.data_ptr()),
                  c_vp(torch.cuda.current_stream().cuda_stream))
        cache = list(feat_cache) if not first else self._cache_views(arena, base, h, w)
        return pixels.unsqueeze(0), cache

    __call__ = forward


class VAEDecoderWrapperSingle(VAEDecoderWrapper):
    """Mirror of the reference's `VAEDecoderWrapperSingle` (demo_utils/vae.py:150-195, its TensorRT-export / one-latent-frame
    form): `forward(z[B, 1, 16, h, w], is_first_frame, *feat_cache)` -> `(pixels[B, 4, 3, 8h, 8w] in [-1, 1], feat_cache)` with
    the 32 feature caches ALWAYS present as tensors `[1, C, 2, H, W]` in execution order (zeros before the first frame:
    `zero_cache(h, w)` = demo_utils/constant.py:6-39 at another latent size) and the first frame marked by the caller.  Same
    weights, same kernels and arena as `VAEDecoderWrapper`; the graph differs in the temporal upsampling only
    (`rtv_vae_decode_single`, include/rtv_hip.h): four frames also on the first call.  The returned caches are views of the
    stream's arena, updated in place by the next call (as for VAEDecoderWrapper: clone a slot to keep a snapshot)."""

    NUM_CACHES = 32

    _SHAPES = [(16, 1)] + [(384, 1)] * 11 + [(192, 2)] + [(384, 2)] * 6 + [(192, 4)] * 6 + [(96, 8)] * 7

    def zero_cache(self, h, w, dtype=torch.float16):
        """The 32 zero caches of a new stream (demo_utils/constant.py:6-39 at latent size h x w) as PLAIN tensors that stay zero:
        the reference keeps them as a reusable constant (`feat_cache = ZERO_VAE_CACHE` for every new stream,
        demo_utils/vae_torch2trt.py:168), so the same list may start any number of streams - the first forward() of each copies
        them into a fresh arena and returns that arena's views (rounds 4-5 returned arena views here, which the
        stream then updated in place, so a caller reusing the "constant" started its second stream on the first one's state)."""
        return [torch.zeros(1, c, 2, h * k, w * k, dtype=dtype, device=self.device) for c, k in self._SHAPES]

    def new_stream_cache(self, h, w):
        """Opt-in fast form of `zero_cache` for ONE new stream: views of a fresh, zero-filled, registered arena, so the first
        forward() finds its arena by lookup and neither allocates nor copies 32 slots of zeros.  SINGLE USE: forward() updates
        these tensors in place - do not keep the list as a constant for further streams (use `zero_cache`), and mind that every
        call takes one of the wrapper's `cache_streams` arena slots."""
        if self.row_range(h) != (0, 8 * h):
            raise NotImplementedError("the single-frame form decodes whole frames")
        arena = self._new_arena(h, w)                      # zero-filled
        views = self._cache_views(arena, (-arena.data_ptr()) % 256, h, w)[:self.NUM_CACHES]
        assert [tuple(v.shape) for v in views] == [(1, c, 2, h * k, w * k) for c, k in self._SHAPES]
        return views

    def forward(self, z, is_first_frame, *feat_cache):
        if self._w is None:
            raise RuntimeError("weights not loaded")
        if not z.is_cuda:
            raise RuntimeError("realtime_video_amd.VAEDecoderWrapperSingle needs GPU tensors (no CPU fallback)")
        B, T, C, h, w = z.shape
        assert T == 1                                                   # demo_utils/vae.py:180
        if B != 1 or C != 16:
            raise NotImplementedError("VAE decoder: batch 1, 16 latent channels")
        if len(feat_cache) != self.NUM_CACHES or any(c is None for c in feat_cache):
            raise ValueError(f"VAEDecoderWrapperSingle takes its {self.NUM_CACHES} feature caches as tensors (zero_cache(h, w) "
                             "before the first frame)")
        if self.row_range(h) != (0, 8 * h):
            raise NotImplementedError("the single-frame form decodes whole frames")
        # a Python bool / int is the fast path; a TENSOR flag (what the reference's torch.cond form takes, demo_utils/vae.py:
        # 112-123) is read back here - one device-to-host sync per frame, and the call cannot be captured in a hipGraph
        first = bool(is_first_frame.item() if torch.is_tensor(is_first_frame) else is_first_frame)
        cache = list(feat_cache) + [None] * (55 - self.NUM_CACHES)
        arena, base = self._arenas.lookup(cache, (h, w), lambda: self._new_arena(h, w),
                                          lambda a, b: self._cache_views(a, b, h, w))
        zz = z[0].to(torch.float16).contiguous()
        pixels = torch.empty((4, 3, 8 * h, 8 * w), dtype=torch.float32, device=z.device)
        _lib.call("rtv_vae_decode_single", ctypes.byref(self._w), c_vp(zz.data_ptr()), h, w, int(first),
                  c_vp(arena.data_ptr() + base), ctypes.c_size_t(arena.numel() - base), c_vp(pixels.data_ptr()),
                  c_vp(torch.cuda.current_stream().cuda_stream))
        return pixels.unsqueeze(0), cache[:self.NUM_CACHES]

    __call__ = forward


class CacheArenas:
    """Owner lookup for the streaming feature caches.  The 55-slot cache list a wrapper returns is a set of VIEWS into one arena
    allocation that the next call updates IN PLACE (the reference allocates fresh tensors per call; keep a `.clone()` of a slot
    if you need a snapshot - it will not be overwritten, and it can be fed back).  The arena of a list is found through the
    storage address of its first slot; a list whose slots were cloned / moved (no longer views of a known arena) gets a new
    arena with the slot contents copied in, and a list that belongs to another frame size is refused."""

    KEEP = 8          # concurrent streams per wrapper whose arenas stay registered (least recently used goes first)

    def __init__(self, keep=None):
        self._by_ptr = {}
        self.keep = int(keep) if keep is not None else self.KEEP

    def register(self, views, arena, base, size):
        # The entry holds the arena, so its address cannot be reused while it is listed.  An evicted stream that comes back
        # takes the copy-in path of lookup() (correct, but 55 slot copies + a new arena): warn, it is a performance cliff.
        # Weak references to the views handed out tell recycle() when the owner has dropped the whole cache list.
        self._by_ptr[views[0].data_ptr()] = (arena, base, size, [weakref.ref(v) for v in views if v is not None])
        while len(self._by_ptr) > self.keep:
            self._by_ptr.pop(next(iter(self._by_ptr)))
            import warnings
            warnings.warn(f"rtv: more than {self.keep} concurrent feature-cache streams on one VAE wrapper; the least recently "
                          "used arena was dropped (raise CacheArenas.KEEP / the wrapper's cache_streams)")

    def recycle(self, size):
        """(arena, base) of a registered stream of this frame size whose cache list is GONE - every view this wrapper handed
        out for it has been garbage-collected, so nobody can feed it back - or None.  The entry is removed; the caller
        re-registers the arena with the views of the new stream.  This is what keeps the session's per-block one-shot encode
        (release_server.py:572-575: a fresh cache every block, dropped right after) on ONE arena instead of allocating,
        zero-filling and registering a new one per block.  (Sub-views a caller sliced out of a dropped list keep pointing
        into the arena and see the next stream's data - the same in-place contract as for a live stream.)"""
        for key, ent in self._by_ptr.items():
            if ent[2] == size and all(r() is None for r in ent[3]):
                del self._by_ptr[key]
                return ent[0], ent[1]
        return None

    def lookup(self, cache, size, new_arena, make_views):
        """`cache` (a list, updated in place when it has to be rebuilt) -> (arena, base)."""
        key = cache[0].data_ptr()
        ent = self._by_ptr.get(key)
        if ent is not None:
            lo, hi = ent[0].data_ptr(), ent[0].data_ptr() + ent[0].numel() * ent[0].element_size()
            # a hit only if EVERY slot still is a view of that arena (a restored snapshot may have replaced some of them:
            # their contents must not be ignored -> copy-in path below)
            if all(c is None or lo <= c.data_ptr() < hi for c in cache):
                if ent[2] != size:
                    raise ValueError(f"feature cache belongs to a {ent[2]} stream, this call is {size}")
                self._by_ptr[key] = self._by_ptr.pop(key)      # most recently used
                return ent[0], ent[1]
        arena = new_arena()
        base = (-arena.data_ptr()) % 256
        views = make_views(arena, base)       # registers the new arena
        for i, (dst, src) in enumerate(zip(views, cache)):
            if (dst is None) != (src is None):
                raise ValueError(f"feature cache slot {i}: unexpected {'tensor' if dst is None else 'None'}")
            if dst is not None:
                if tuple(dst.shape) != tuple(src.shape):
                    raise ValueError(f"feature cache slot {i} has shape {tuple(src.shape)}, this stream needs {tuple(dst.shape)}")
                dst.copy_(src)
        cache[:] = views
        return arena, base


class WanVAEWrapper:
    """Mirror of the reference's `WanVAEWrapper` (utils/wan_wrapper.py:58-118) for Step 4 of
    `CausalInferencePipeline.inference` (pipeline/causal_inference.py:252): `decode_to_pixel(latent[B, T, 16, h, w],
    use_cache=False)` -> pixels [B, T', 3, 8h, 8w] float32 in [-1, 1], T' = 4T - 3.  `WanVAE_.decode` (wan/modules/vae.py:
    519-543) runs its decoder frame by frame over a cleared feature cache - the same computation as one streaming call on
    fresh caches, which is what this delegates to.  `use_cache=True` (`cached_decode`, :545-567) keeps the caches across calls
    until `clear_cache()`.  State-dict keys: the reference module's (`model.decoder.*`, `model.conv2.*`; the encoder half of
    the checkpoint is ignored here - see vae_encoder.VAEEncoderWrapper)."""

    def __init__(self, device="cuda"):
        self.decoder = VAEDecoderWrapper(device)
        self.encoder = None            # built on first use (encode_to_latent is off the inference path)
        self._enc_sd = None
        self._device = device
        self._caches = {}

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def requires_grad_(self, flag=False):
        return self

    def load_state_dict(self, sd, strict=True):
        inner = {}
        for k, v in sd.items():
            k = k[len("model."):] if k.startswith("model.") else k
            if k.startswith("decoder.") or k.startswith("conv2."):
                inner[k] = v
        self.decoder.load_state_dict(inner, strict)
        enc = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}
        self._enc_sd = {k: v for k, v in enc.items() if k.startswith("encoder.") or k.startswith("conv1.")} or None
        self.encoder = None
        return self

    def init_random_weights(self, seed=0):
        self.decoder.init_random_weights(seed)
        return self

    def encode_to_latent(self, pixel):
        """utils/wan_wrapper.py:79-93: pixel [B, 3, T, H, W] in [-1, 1] (T = 1 + 4k) -> normalised latents [B, T', 16, h, w];
        `WanVAE_.encode` (vae.py:491-517) chunks time 1, 4, 4, ... over a cleared cache = one non-streamed encoder call."""
        from .vae_encoder import VAEEncoderWrapper
        if self.encoder is None:
            if self._enc_sd is None:
                raise RuntimeError("encode_to_latent needs the encoder half of the VAE state dict (encoder.*, conv1.*)")
            self.encoder = VAEEncoderWrapper(device=self._device)
            self.encoder.load_state_dict(self._enc_sd)
        out = [self.encoder(pixel[b:b + 1].half(), [None] * 55, stream=False)[0][0].float() for b in range(pixel.shape[0])]
        return torch.stack(out, dim=0).permute(0, 2, 1, 3, 4)

    def clear_cache(self):
        self._caches = {}

    def decode_to_pixel(self, latent, use_cache=False):
        if use_cache and latent.shape[0] != 1:
            raise AssertionError("Batch size must be 1 when using cache")       # wan_wrapper.py:100
        out = []
        for b in range(latent.shape[0]):
            cache = self._caches.get(b, [None] * 55) if use_cache else [None] * 55
            px, cache = self.decoder(latent[b:b + 1], *cache)
            if use_cache:
                self._caches[b] = cache
            out.append(px[0])
        return torch.stack(out, dim=0)
