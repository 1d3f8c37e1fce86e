"""CPU ORACLE (test infrastructure only) — restatement of the reference's streaming causal-conv3d VAE
(decoder first, the streaming encoder at the end of the file): `VAEDecoderWrapper` / `VAEDecoder3d` / `Resample` (demo_utils/vae_block3.py:8-114, :177-230,
:334-443) over the building blocks of wan/modules/vae.py (`CausalConv3d` :17-36, `RMS_norm` :39-54,
`Upsample` :57-63, `ResidualBlock` :175-209, `AttentionBlock` :212-251).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  Pinned
against the reference's own modules through oracle/make_golden.py -> tests/golden/vae_decoder.pt
(the upstream repo has no tests for this path).  Weights: plain dict with the reference's
state_dict names (`decoder.*`, `conv2.*`).
"""
import math

import torch
import torch.nn.functional as F

MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508,
        0.4134, -0.0715, 0.5517, -0.3632, -0.1922, -0.9497, 0.2503, -0.2921]
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743,
       3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251, 1.9160]
DIMS = [384, 384, 384, 192, 96]  # vae_block3.py:354 with dim=96, dim_mult=[1,2,4,4]
CACHE_T = 2


def causal_conv3d(x, weight, bias, padding, cache_x=None, stride=1):
    """CausalConv3d.forward, wan/modules/vae.py:27-36.  padding = (pt, ph, pw) as given to the ctor."""
    pad = [padding[2], padding[2], padding[1], padding[1], 2 * padding[0], 0]
    if cache_x is not None and pad[4] > 0:
        x = torch.cat([cache_x, x], dim=2)
        pad[4] -= cache_x.shape[2]
    return F.conv3d(F.pad(x, pad), weight, bias, stride=stride)


def rms_norm(x, gamma):
    """RMS_norm.forward (channel_first), wan/modules/vae.py:51-54."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * gamma


def _cached_conv(x, w, name, feat_cache, feat_idx):
    """The cache idiom of ResidualBlock.forward (vae.py:193-206) / VAEDecoder3d.forward
    (vae_block3.py:406-413): new cache = last 2 time slices of the conv INPUT (prepending the old
    cache's last slice when only one new slice exists)."""
    idx = feat_idx[0]
    cache_x = x[:, :, -CACHE_T:].clone()
    if cache_x.shape[2] < 2 and feat_cache[idx] is not None:
        cache_x = torch.cat([feat_cache[idx][:, :, -1:], cache_x], dim=2)
    out = causal_conv3d(x, w[name + ".weight"], w[name + ".bias"], (1, 1, 1), feat_cache[idx])
    feat_cache[idx] = cache_x
    feat_idx[0] += 1
    return out


def residual_block(x, w, pre, feat_cache, feat_idx):
    """ResidualBlock.forward, vae.py:191-209 (residual = RMS_norm, SiLU, conv, RMS_norm, SiLU, Dropout, conv)."""
    if pre + ".shortcut.weight" in w:
        h = causal_conv3d(x, w[pre + ".shortcut.weight"], w[pre + ".shortcut.bias"], (0, 0, 0))
    else:
        h = x
    x = F.silu(rms_norm(x, w[pre + ".residual.0.gamma"]))
    x = _cached_conv(x, w, pre + ".residual.2", feat_cache, feat_idx)
    x = F.silu(rms_norm(x, w[pre + ".residual.3.gamma"]))
    x = _cached_conv(x, w, pre + ".residual.6", feat_cache, feat_idx)
    return x + h


def attention_block(x, w, pre):
    """AttentionBlock.forward, vae.py:229-251 (single head over h*w tokens per frame)."""
    identity = x
    b, c, t, h, wd = x.shape
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, wd)
    x = rms_norm(x, w[pre + ".norm.gamma"])
    qkv = F.conv2d(x, w[pre + ".to_qkv.weight"], w[pre + ".to_qkv.bias"])
    q, k, v = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
    x = F.scaled_dot_product_attention(q, k, v)
    x = x.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, wd)
    x = F.conv2d(x, w[pre + ".proj.weight"], w[pre + ".proj.bias"])
    x = x.reshape(b, t, c, h, wd).permute(0, 2, 1, 3, 4)
    return x + identity


def resample_up(x, w, pre, mode, feat_cache, feat_idx):
    """Resample.forward for 'upsample2d' / 'upsample3d', vae_block3.py:46-72 (same as vae.py:104-130)."""
    b, c, t, h, wd = x.shape
    if mode == "upsample3d":
        idx = feat_idx[0]
        if feat_cache[idx] is None:
            feat_cache[idx] = torch.zeros(b, c, CACHE_T, h, wd, dtype=x.dtype, device=x.device)
            feat_idx[0] += 1
        else:
            cache_x = x[:, :, -CACHE_T:].clone()
            if cache_x.shape[2] < 2:
                padding = torch.where(feat_cache[idx][:, :, -1:] == 0, 0, cache_x)
                cache_x = torch.cat([padding, cache_x], dim=2)
            x = causal_conv3d(x, w[pre + ".time_conv.weight"], w[pre + ".time_conv.bias"], (1, 0, 0), feat_cache[idx])
            feat_cache[idx] = cache_x
            feat_idx[0] += 1
            x = x.reshape(b, 2, c, t, h, wd)
            x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, c, t * 2, h, wd)
    t = x.shape[2]
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, wd)
    x = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest").type_as(x)  # Upsample, vae.py:57-63
    x = F.conv2d(x, w[pre + ".resample.1.weight"], w[pre + ".resample.1.bias"], padding=1)
    return x.reshape(b, t, c // 2, 2 * h, 2 * wd).permute(0, 2, 1, 3, 4)


def decoder3d(x, w, feat_cache):
    """VAEDecoder3d.forward, vae_block3.py:386-443.  x: [B, 16, 1, h, w] (already through conv2)."""
    feat_idx = [0]
    x = _cached_conv(x, w, "decoder.conv1", feat_cache, feat_idx)
    x = residual_block(x, w, "decoder.middle.0", feat_cache, feat_idx)
    x = attention_block(x, w, "decoder.middle.1")
    x = residual_block(x, w, "decoder.middle.2", feat_cache, feat_idx)
    li = 0
    for i in range(4):
        for _ in range(3):
            x = residual_block(x, w, f"decoder.upsamples.{li}", feat_cache, feat_idx)
            li += 1
        if i != 3:
            x = resample_up(x, w, f"decoder.upsamples.{li}", "upsample3d" if i < 2 else "upsample2d",
                            feat_cache, feat_idx)
            li += 1
    x = F.silu(rms_norm(x, w["decoder.head.0.gamma"]))
    # head conv: the cache rule written out with an explicit zero slot (vae_block3.py:427-440)
    idx = feat_idx[0]
    b, c, t, h, wd = x.shape
    cache_x = torch.zeros(b, c, CACHE_T, h, wd, dtype=x.dtype, device=x.device)
    fill = x[:, :, -CACHE_T:].clone()
    cache_x[:, :, -fill.shape[2]:] = fill
    if fill.shape[2] < 2 and feat_cache[idx] is not None:
        cache_x = torch.cat([feat_cache[idx][:, :, -1:], fill], dim=2)
    x = causal_conv3d(x, w["decoder.head.2.weight"], w["decoder.head.2.bias"], (1, 1, 1), feat_cache[idx])
    feat_cache[idx] = cache_x
    feat_idx[0] += 1
    return x, feat_cache


def decoder_wrapper_forward(w, z, feat_cache):
    """VAEDecoderWrapper.forward, vae_block3.py:195-230.  z: [B, T, 16, h, w]; feat_cache: list of 55
    (Tensor | None).  Returns (pixels [B, T', 3, 8h, 8w] float32 in [-1, 1], feat_cache)."""
    z = z.permute(0, 2, 1, 3, 4)
    feat_cache = list(feat_cache)
    mean = torch.tensor(MEAN, dtype=z.dtype, device=z.device).view(1, 16, 1, 1, 1)
    inv_std = 1.0 / torch.tensor(STD, dtype=z.dtype, device=z.device).view(1, 16, 1, 1, 1)
    z = z / inv_std + mean
    x = causal_conv3d(z, w["conv2.weight"], w["conv2.bias"], (0, 0, 0))
    outs = []
    for i in range(x.shape[2]):
        o, feat_cache = decoder3d(x[:, :, i:i + 1], w, feat_cache)
        outs.append(o)
    out = torch.cat(outs, 2).float().clamp_(-1, 1)
    return out.permute(0, 2, 1, 3, 4), feat_cache


# ------------------------------------------------------------------------------------------------
# VAEDecoderWrapperSingle (demo_utils/vae.py): the one-latent-frame form with explicit caches and an `is_first_frame` input
# ------------------------------------------------------------------------------------------------
def _single_cached_conv(x, w, name, cache):
    """The cache idiom of demo_utils/vae.py:33-43 / :263-272 (caches are always tensors here): -> (conv output, new cache)."""
    cache_x = x[:, :, -CACHE_T:].clone()
    if cache_x.shape[2] < 2 and cache is not None:
        cache_x = torch.cat([cache[:, :, -1:], cache_x], dim=2)
    return causal_conv3d(x, w[name + ".weight"], w[name + ".bias"], (1, 1, 1), cache), cache_x


def _single_residual_block(x, w, pre, cache_1, cache_2):
    """ResidualBlock.forward, demo_utils/vae.py:28-47."""
    if pre + ".shortcut.weight" in w:
        h = causal_conv3d(x, w[pre + ".shortcut.weight"], w[pre + ".shortcut.bias"], (0, 0, 0))
    else:
        h = x
    x = F.silu(rms_norm(x, w[pre + ".residual.0.gamma"]))
    x, c1 = _single_cached_conv(x, w, pre + ".residual.2", cache_1)
    x = F.silu(rms_norm(x, w[pre + ".residual.3.gamma"]))
    x, c2 = _single_cached_conv(x, w, pre + ".residual.6", cache_2)
    return x + h, c1, c2


def _single_resample(x, w, pre, mode, is_first_frame, cache):
    """Resample.forward + temporal_conv, demo_utils/vae.py:72-123.  On the first frame the temporal doubling is a ZERO frame in
    front of every frame (cat([zeros, x], dim=1) through the same reshape / stack) and the cache stays what it was (:86-90);
    otherwise time_conv over [cache | x] and the new cache is the last two input frames - [zeros, x] for a one-frame input
    (:106-111: zeros_like, not the old cache's last frame)."""
    b, c, t, h, wd = x.shape
    out_cache = None
    if mode == "upsample3d":
        cache_x = x[:, :, -CACHE_T:].clone()
        if cache_x.shape[2] < 2 and cache is not None:
            cache_x = torch.cat([torch.zeros_like(cache_x), cache_x], dim=2)
        if is_first_frame:
            x = torch.cat([torch.zeros_like(x), x], dim=1)
            out_cache = cache.clone()
        else:
            x = causal_conv3d(x, w[pre + ".time_conv.weight"], w[pre + ".time_conv.bias"], (1, 0, 0), cache)
            out_cache = cache_x
        x = x.reshape(b, 2, c, t, h, wd)
        x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, c, t * 2, h, wd)
    t = x.shape[2]
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, wd)
    x = F.interpolate(x.float(), scale_factor=(2.0, 2.0), mode="nearest").type_as(x)
    x = F.conv2d(x, w[pre + ".resample.1.weight"], w[pre + ".resample.1.bias"], padding=1)
    return x.reshape(b, t, c // 2, 2 * h, 2 * wd).permute(0, 2, 1, 3, 4), out_cache


def decoder_single_forward(w, z, is_first_frame, feat_cache):
    """VAEDecoderWrapperSingle.forward, demo_utils/vae.py:171-195 over VAEDecoder3d.forward :255-314.  z: [B, 1, 16, h, w];
    feat_cache: the 32 cache tensors in execution order (zeros before the first frame, demo_utils/constant.py:6-39).  Returns
    (pixels [B, 4, 3, 8h, 8w] in [-1, 1] - four frames on the first call as well -, the 32 new caches)."""
    assert z.shape[1] == 1
    fc = list(feat_cache)
    first = bool(is_first_frame)
    z = z.permute(0, 2, 1, 3, 4)
    mean = torch.tensor(MEAN, dtype=z.dtype, device=z.device).view(1, 16, 1, 1, 1)
    inv_std = 1.0 / torch.tensor(STD, dtype=z.dtype, device=z.device).view(1, 16, 1, 1, 1)
    x = causal_conv3d(z / inv_std + mean, w["conv2.weight"], w["conv2.bias"], (0, 0, 0))
    out, idx = [], 0
    x, c = _single_cached_conv(x, w, "decoder.conv1", fc[idx])
    out.append(c)
    idx += 1
    x, c1, c2 = _single_residual_block(x, w, "decoder.middle.0", fc[idx], fc[idx + 1])
    out += [c1, c2]
    idx += 2
    x = attention_block(x, w, "decoder.middle.1")
    x, c1, c2 = _single_residual_block(x, w, "decoder.middle.2", fc[idx], fc[idx + 1])
    out += [c1, c2]
    idx += 2
    li = 0
    for i in range(4):
        for _ in range(3):
            x, c1, c2 = _single_residual_block(x, w, f"decoder.upsamples.{li}", fc[idx], fc[idx + 1])
            out += [c1, c2]
            idx += 2
            li += 1
        if i != 3:
            x, c = _single_resample(x, w, f"decoder.upsamples.{li}", "upsample3d" if i < 2 else "upsample2d", first, fc[idx])
            if c is not None:
                out.append(c)
                idx += 1
            li += 1
    x = F.silu(rms_norm(x, w["decoder.head.0.gamma"]))
    x, c = _single_cached_conv(x, w, "decoder.head.2", fc[idx])
    out.append(c)
    return x.clamp(-1, 1).permute(0, 2, 1, 3, 4), out


def single_zero_cache(h, w, dtype=torch.float32, device="cpu"):
    """demo_utils/constant.py:6-39 (ZERO_VAE_CACHE) for an h x w latent: 32 zero tensors [1, C, 2, H, W] in execution order."""
    shapes = [(16, 1)] + [(384, 1)] * 11 + [(192, 2)] + [(384, 2)] * 6 + [(192, 4)] * 6 + [(96, 8)] * 7
    return [torch.zeros(1, c, 2, h * k, w * k, dtype=dtype, device=device) for c, k in shapes]


def decoder_conv_specs():
    """(state_dict prefix, kind, Cin, Cout) for every conv of the decoder in execution order."""
    specs = [("decoder.conv1", "c3", 16, 384)]

    def res(pre, cin, cout):
        out = [(pre + ".residual.2", "c3", cin, cout), (pre + ".residual.6", "c3", cout, cout)]
        if cin != cout:
            out.append((pre + ".shortcut", "c1", cin, cout))
        return out

    specs += res("decoder.middle.0", 384, 384) + res("decoder.middle.2", 384, 384)
    li, cin = 0, 384
    for i, cout in enumerate(DIMS[1:]):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(3):
            specs += res(f"decoder.upsamples.{li}", cin, cout)
            cin = cout
            li += 1
        if i != 3:
            if i < 2:
                specs.append((f"decoder.upsamples.{li}.time_conv", "t3", cout, 2 * cout))
            specs.append((f"decoder.upsamples.{li}.resample.1", "c2", cout, cout // 2))
            li += 1
    specs.append(("decoder.head.2", "c3", 96, 3))
    return specs


def make_vae_weights(seed=0, dtype=torch.float32):
    """Deterministic synthetic decoder weights (there is no Wan2.1_VAE.pth offline): PyTorch-default-like
    uniform conv init (bound 1/sqrt(fan_in)), gammas near 1, and a NON-zero attention proj (zero-init in
    the reference, vae.py:227, would make the attention branch vanish)."""
    g = torch.Generator().manual_seed(seed)
    w = {}

    def conv(name, cout, cin, *k):
        fan_in = cin * math.prod(k)
        bound = 1.0 / math.sqrt(fan_in)
        w[name + ".weight"] = (torch.rand(cout, cin, *k, generator=g) * 2 - 1) * bound
        w[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def gamma(name, c, dims):
        w[name] = 1 + 0.1 * torch.randn(c, *([1] * dims), generator=g)

    conv("conv2", 16, 16, 1, 1, 1)
    for pre, kind, cin, cout in decoder_conv_specs():
        if kind == "c3":
            conv(pre, cout, cin, 3, 3, 3)
        elif kind == "c1":
            conv(pre, cout, cin, 1, 1, 1)
        elif kind == "t3":
            conv(pre, cout, cin, 3, 1, 1)
        elif kind == "c2":
            conv(pre, cout, cin, 3, 3)
    # norms
    def res_norms(pre, cin, cout):
        gamma(pre + ".residual.0.gamma", cin, 3)
        gamma(pre + ".residual.3.gamma", cout, 3)

    res_norms("decoder.middle.0", 384, 384)
    res_norms("decoder.middle.2", 384, 384)
    gamma("decoder.middle.1.norm.gamma", 384, 2)
    conv("decoder.middle.1.to_qkv", 384 * 3, 384, 1, 1)
    conv("decoder.middle.1.proj", 384, 384, 1, 1)
    li, cin = 0, 384
    for i, cout in enumerate(DIMS[1:]):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(3):
            res_norms(f"decoder.upsamples.{li}", cin, cout)
            cin = cout
            li += 1
        if i != 3:
            li += 1
    gamma("decoder.head.0.gamma", 96, 3)
    return {k: v.to(dtype) for k, v in w.items()}


# ======================================================================================= encoder
# `VAEEncoderWrapper` (demo_utils/vae_block3.py:116-175) over `Encoder3d` (wan/modules/vae.py:254-345) and the
# 'downsample2d' / 'downsample3d' branches of `Resample` (vae.py:84-96, :104-158); configuration of `_video_vae`
# (vae.py:591-598): dim=96, z_dim=16 (encoder emits 2*z_dim), dim_mult [1,2,4,4], 2 res blocks per stage,
# temperal_downsample [False, True, True].  Pinned through oracle/make_golden.py -> tests/golden/vae_encoder.pt.
ENC_DIMS = [96, 96, 192, 384, 384]
ENC_TEMPORAL_DOWN = [False, True, True]


def resample_down(x, w, pre, mode, feat_cache, feat_idx):
    """Resample.forward for 'downsample2d' / 'downsample3d', vae.py:132-158: ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2)
    per frame; downsample3d then runs time_conv (3,1,1)/stride (2,1,1) over [last cached frame | new frames], except on
    the very first call, which only stores the frame."""
    b, c, t, h, wd = x.shape
    x = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, wd)
    x = F.conv2d(F.pad(x, (0, 1, 0, 1)), w[pre + ".resample.1.weight"], w[pre + ".resample.1.bias"], stride=2)
    x = x.reshape(b, t, c, x.shape[-2], x.shape[-1]).permute(0, 2, 1, 3, 4)
    if mode == "downsample3d":
        idx = feat_idx[0]
        if feat_cache[idx] is None:
            feat_cache[idx] = x.clone()
            feat_idx[0] += 1
        else:
            cache_x = x[:, :, -1:].clone()
            x = causal_conv3d(torch.cat([feat_cache[idx][:, :, -1:], x], 2), w[pre + ".time_conv.weight"],
                              w[pre + ".time_conv.bias"], (0, 0, 0), stride=(2, 1, 1))
            feat_cache[idx] = cache_x
            feat_idx[0] += 1
    return x


def encoder3d(x, w, feat_cache, feat_idx):
    """Encoder3d.forward, vae.py:307-345.  x: [B, 3, T, H, W] -> [B, 32, T', H/8, W/8]."""
    x = _cached_conv(x, w, "encoder.conv1", feat_cache, feat_idx)
    li = 0
    for i in range(4):
        for _ in range(2):
            x = residual_block(x, w, f"encoder.downsamples.{li}", feat_cache, feat_idx)
            li += 1
        if i != 3:
            mode = "downsample3d" if ENC_TEMPORAL_DOWN[i] else "downsample2d"
            x = resample_down(x, w, f"encoder.downsamples.{li}", mode, feat_cache, feat_idx)
            li += 1
    x = residual_block(x, w, "encoder.middle.0", feat_cache, feat_idx)
    x = attention_block(x, w, "encoder.middle.1")
    x = residual_block(x, w, "encoder.middle.2", feat_cache, feat_idx)
    x = F.silu(rms_norm(x, w["encoder.head.0.gamma"]))
    return _cached_conv(x, w, "encoder.head.2", feat_cache, feat_idx)


def encoder_wrapper_forward(w, z, feat_cache, stream=False):
    """VAEEncoderWrapper.forward, vae_block3.py:138-175.  z: [B, 3, T, H, W] pixels in [-1, 1]; feat_cache: list of
    55 (Tensor | None).  Time is split 1,4,4,... (non-stream, fresh cache) or 4,4,... (stream).  Returns
    (mu [B, 16, T', H/8, W/8] normalised with the latent mean / std, feat_cache)."""
    feat_cache = list(feat_cache)
    t = z.shape[2]
    iter_ = 1 + (t - 1) // 4
    offset = 1
    out = None
    for i in range(iter_):
        feat_idx = [0]
        if i == 0 and feat_cache[0] is None:
            out = encoder3d(z[:, :, :1], w, feat_cache, feat_idx)
        else:
            slice_start = i - 1
            if stream:
                offset = 0
                slice_start = i
            out_ = encoder3d(z[:, :, offset + 4 * slice_start:offset + 4 * (slice_start + 1)], w, feat_cache, feat_idx)
            out = out_ if (i == 0 and stream) else torch.cat([out, out_], 2)
    mu = causal_conv3d(out, w["conv1.weight"], w["conv1.bias"], (0, 0, 0)).chunk(2, dim=1)[0]
    mean = torch.tensor(MEAN, dtype=z.dtype, device=z.device).view(1, 16, 1, 1, 1)
    inv_std = 1.0 / torch.tensor(STD, dtype=z.dtype, device=z.device).view(1, 16, 1, 1, 1)
    return (mu - mean) * inv_std, feat_cache


def encoder_conv_specs():
    """(state_dict prefix, kind, Cin, Cout) for every conv of the encoder in execution order
    (kinds: c3 3x3x3 causal, c1 1x1x1, d2 Conv2d 3x3 stride 2, t3 time_conv (3,1,1) stride 2)."""
    specs = [("encoder.conv1", "c3", 3, 96)]

    def res(pre, cin, cout):
        out = [(pre + ".residual.2", "c3", cin, cout), (pre + ".residual.6", "c3", cout, cout)]
        if cin != cout:
            out.append((pre + ".shortcut", "c1", cin, cout))
        return out

    li = 0
    for i, (cin, cout) in enumerate(zip(ENC_DIMS[:-1], ENC_DIMS[1:])):
        for _ in range(2):
            specs += res(f"encoder.downsamples.{li}", cin, cout)
            cin = cout
            li += 1
        if i != 3:
            specs.append((f"encoder.downsamples.{li}.resample.1", "d2", cout, cout))
            if ENC_TEMPORAL_DOWN[i]:
                specs.append((f"encoder.downsamples.{li}.time_conv", "t3", cout, cout))
            li += 1
    specs += res("encoder.middle.0", 384, 384) + res("encoder.middle.2", 384, 384)
    specs.append(("encoder.head.2", "c3", 384, 32))
    return specs


def make_vae_encoder_weights(seed=1, dtype=torch.float32):
    """Deterministic synthetic encoder weights (same recipe as make_vae_weights; reference names `encoder.*`,
    `conv1.*`)."""
    g = torch.Generator().manual_seed(seed)
    w = {}

    def conv(name, cout, cin, *k):
        bound = 1.0 / math.sqrt(cin * math.prod(k))
        w[name + ".weight"] = (torch.rand(cout, cin, *k, generator=g) * 2 - 1) * bound
        w[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def gamma(name, c, dims):
        w[name] = 1 + 0.1 * torch.randn(c, *([1] * dims), generator=g)

    kshape = {"c3": (3, 3, 3), "c1": (1, 1, 1), "d2": (3, 3), "t3": (3, 1, 1)}
    for pre, kind, cin, cout in encoder_conv_specs():
        conv(pre, cout, cin, *kshape[kind])
        if pre.endswith(".residual.2"):
            gamma(pre[:-len(".residual.2")] + ".residual.0.gamma", cin, 3)
            gamma(pre[:-len(".residual.2")] + ".residual.3.gamma", cout, 3)
    gamma("encoder.middle.1.norm.gamma", 384, 2)
    conv("encoder.middle.1.to_qkv", 384 * 3, 384, 1, 1)
    conv("encoder.middle.1.proj", 384, 384, 1, 1)
    gamma("encoder.head.0.gamma", 384, 3)
    conv("conv1", 32, 32, 1, 1, 1)
    return {k: v.to(dtype) for k, v in w.items()}


# ----------------------------------------------------------------------------------------- frame output format
def frames_to_rgb8(pixels):
    """The bytes the reference's frame path hands to the JPEG encoder, from the decoder's float pixels [.., 3, H, W] in
    [-1, 1]: release_server.py:980-984 (`cpu_tensor.add_(1.0).mul_(0.5).clamp_(0.0, 1.0)` on the pinned host copy) followed
    by `TF.to_pil_image(frames[0, idx], "RGB")` (:972).  torchvision is third-party and not in the reference tree; its
    published to_pil_image converts a float tensor with `pic.mul(255).byte()` (truncation) and lays it out H x W x C.
    Returns uint8 [.., H, W, 3]."""
    x = pixels.detach().float().cpu().clone()
    x = x.add_(1.0).mul_(0.5).clamp_(0.0, 1.0)
    return x.mul(255).byte().movedim(-3, -1).contiguous()
