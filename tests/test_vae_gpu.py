"""VAE decoder parity on the MI355X (through the C ABI): implicit-GEMM convolution vs torch fp32
convolutions, fused RMS_norm+SiLU / softmax kernels, and the full streaming decoder vs golden pixels
minted from the upstream VAEDecoderWrapper (fp32 CPU).  Stated tolerance: fp16 pipeline vs fp32
reference, max-abs <= 2e-2 on pixels in [-1, 1]."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from conftest import max_abs, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else 0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _conv_cl(inp, w, bias, T, H, W, kt, kh, kw, ups=0, n_split=0, residual=None):
    """inp: channels-last concat buffer [Tin][inH][inW][Cin] fp16; w: torch conv weight."""
    from realtime_video_amd import _lib
    from realtime_video_amd.vae_decoder import pack_conv_weight
    import realtime_video_amd.vae_decoder  # noqa: F401  (registers signatures)
    wp = pack_conv_weight(w).to(DEV)
    Cout, Cin = wp.shape[0], wp.shape[2]
    zeros = torch.zeros(64, dtype=torch.float16, device=DEV)
    if n_split:
        out = torch.empty(2 * T, H, W, n_split, dtype=torch.float16, device=DEV)
        out_ld = n_split
    else:
        out = torch.empty(T, H, W, Cout, dtype=torch.float16, device=DEV)
        out_ld = Cout
    _lib.call("rtv_conv_cl", _p(inp), _p(wp), _p(bias), _p(residual), Cout, _p(out), out_ld, T, H, W, Cin, Cout,
              kt, kh, kw, ups, n_split, _p(zeros), _stream())
    return out


@pytest.mark.parametrize("Cin,Cout", [(32, 384), (96, 96), (192, 192), (384, 384), (96, 8), (192, 384)])
@pytest.mark.parametrize("T,H,W", [(1, 8, 12), (4, 16, 24)])
def test_causal_conv3d_with_cache_concat(Cin, Cout, T, H, W):
    g = torch.Generator().manual_seed(Cin + Cout + T)
    x = torch.randn(T + 2, H, W, Cin, generator=g).half().to(DEV)          # [2 cached | T new]
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).half().to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.1).half().to(DEV)
    res = torch.randn(T, H, W, Cout, generator=g).half().to(DEV)
    out = _conv_cl(x, w, b, T, H, W, 3, 3, 3, residual=res)
    xin = x.permute(3, 0, 1, 2).unsqueeze(0).float()                        # [1, C, T+2, H, W]
    ref = F.conv3d(F.pad(xin, (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0].permute(1, 2, 3, 0)
    ref = (ref.half().float() + res.float())
    assert out.shape == ref.shape
    assert max_abs(out, ref) <= 1e-2 and rel_l2(out, ref) <= 2e-3


@pytest.mark.parametrize("Cin,Cout,T,H,W", [(96, 96, 2, 33, 70),      # ragged on both axes: 3 x 3 tiles, partial last row / column
                                             (32, 96, 1, 16, 32),       # the encoder's first conv (3 -> 32 padded channels): one chunk per slice
                                             (192, 384, 3, 17, 31),     # four filter tiles per pixel tile, a single ragged tile
                                             (384, 192, 1, 48, 64),     # twelve channel chunks per time slice
                                             (96, 192, 4, 5, 100)])     # fewer rows than a tile, four column tiles
def test_halo_conv_kernel_edges_and_gather_flag(Cin, Cout, T, H, W):
    """The halo-tile kernel (conv_halo_kernel: 16 x 32 pixel tiles, halo staged once per time slice and channel chunk) on
    ragged image sizes, every channel configuration it is dispatched for, with and without bias / residual - against torch's
    fp32 conv3d and against the gather kernel selected by RTV_CONV_GATHER (same math, another K order: equal up to fp32
    association, i.e. isolated one-ulp fp16 differences).  Repeated launches must be bit-identical (race screen of the
    DMA / barrier schedule)."""
    RTV_CONV_GATHER = 16
    g = torch.Generator().manual_seed(Cin + Cout + T + H)
    x = (torch.randn(T + 2, H, W, Cin, generator=g) * 0.7).half().to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).half().to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.1).half().to(DEV)
    res = torch.randn(T, H, W, Cout, generator=g).half().to(DEV)
    xin = x.permute(3, 0, 1, 2).unsqueeze(0).float()
    conv = F.conv3d(F.pad(xin, (1, 1, 1, 1, 0, 0)), w.float())[0].permute(1, 2, 3, 0)
    for bias, residual in ((b, res), (None, None), (b, None)):
        ref = conv + (bias.float() if bias is not None else 0.0)
        ref = ref.half().float() + (residual.float() if residual is not None else 0.0)
        outs = [_conv_cl(x, w, bias, T, H, W, 3, 3, 3, residual=residual) for _ in range(3)]
        assert max_abs(outs[0], ref) <= 1e-2 and rel_l2(outs[0], ref) <= 2e-3
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        gather = _conv_cl(x, w, bias, T, H, W, 3, 3, 3, ups=RTV_CONV_GATHER, residual=residual)
        assert max_abs(gather, ref) <= 1e-2
        assert float((gather != outs[0]).float().mean()) <= 5e-3 and max_abs(gather, outs[0]) <= 4e-3


def test_halo_conv_kernel_forms_are_bit_identical():
    """The halo-tile kernel exists in three forms - two waves per SIMD (conv_halo_kernel, eight waves, 64 x 96 per wave), one wave
    per SIMD (conv_halo4_kernel, r05: four waves, 128 x 96 per wave, accumulation registers named in inline asm) and the latter
    PERSISTENT (conv_halo4p_kernel, the default: one workgroup per CU walks its tiles, the operand stream runs across tile
    boundaries, the epilogue goes through 24 KiB of spare LDS) - with the same tile, K order and epilogue arithmetic: the same bits
    on every layer kind: 3x3x3 with bias / residual / neither, ragged tiles, one to twelve channel chunks per slice, more tiles
    than CUs (several tiles per workgroup) and fewer, the 3x3 conv behind the nearest-2x upsampling, the fused RMS_norm + SiLU
    epilogue.  Repeated launches bit-identical."""
    from realtime_video_amd import _lib
    from realtime_video_amd.vae_decoder import pack_conv_weight
    lib = _lib.load()
    g = torch.Generator().manual_seed(77)
    zeros = torch.zeros(64, dtype=torch.float16, device=DEV)

    def both(fn):
        outs = {}
        try:
            for mode in (3, 2, 6, 6):                   # two waves per SIMD, one wave per SIMD, persistent (twice)
                lib.rtv_conv_set_halo(mode)
                outs.setdefault(mode, []).append(fn())
        finally:
            lib.rtv_conv_set_halo(1)
        assert torch.isfinite(outs[3][0].float()).all()
        assert torch.equal(outs[2][0], outs[3][0]) and torch.equal(outs[6][0], outs[3][0]) and torch.equal(outs[6][0], outs[6][1])

    for (Cin, Cout, T, H, W) in ((96, 96, 2, 33, 70), (32, 96, 1, 16, 32), (192, 384, 3, 17, 31), (384, 192, 1, 48, 64),
                                 (96, 192, 4, 5, 100), (192, 192, 2, 64, 96),
                                 (96, 96, 3, 200, 330),      # 3 x 13 x 11 = 429 tiles: up to two per workgroup, ragged on both axes, 9 groups per tile
                                 (192, 96, 2, 200, 330)):    # 286 tiles of 18 groups (the halo buffer parity continues across a tile boundary either way)
        x = (torch.randn(T + 2, H, W, Cin, generator=g) * 0.7).half().to(DEV)
        w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).half().to(DEV)
        b = (torch.randn(Cout, generator=g) * 0.1).half().to(DEV)
        res = torch.randn(T, H, W, Cout, generator=g).half().to(DEV)
        for bias, residual in ((b, res), (None, None), (b, None)):
            both(lambda: _conv_cl(x, w, bias, T, H, W, 3, 3, 3, residual=residual))
    for (Ci, Co, Tn, Hn, Wn) in ((384, 192, 1, 17, 23), (384, 192, 2, 8, 40), (192, 96, 3, 33, 70), (192, 96, 4, 80, 160)):   # the last: 400 tiles
        xx = (torch.randn(Tn, Hn, Wn, Ci, generator=g) * 0.7).half().to(DEV)
        ww = (torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5).half().to(DEV)
        bb = (torch.randn(Co, generator=g) * 0.1).half().to(DEV)
        both(lambda: _conv_cl(xx, ww, bb, Tn, 2 * Hn, 2 * Wn, 1, 3, 3, ups=1))
    for (Cin, T, H, W) in ((96, 2, 33, 70), (384, 1, 5, 9)):     # fused RMS_norm + SiLU epilogue (96 filters)
        x = (torch.randn(T + 2, H, W, Cin, generator=g) * 0.7).half().to(DEV)
        wp = pack_conv_weight((torch.randn(96, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).half()).to(DEV)
        b = (torch.randn(96, generator=g) * 0.1).half().to(DEV)
        gamma = (1.0 + 0.2 * torch.randn(96, generator=g)).half().to(DEV)

        def fused():
            out = torch.full((T, H, W, 96), float("nan"), dtype=torch.float16, device=DEV)
            fn = lib.rtv_conv3_norm_silu_cl
            fn.argtypes = _lib.EXTRA_SIGNATURES["rtv_conv3_norm_silu_cl"]
            assert fn(_p(x), _p(wp), _p(b), _p(gamma), _p(out), 96, T, H, W, Cin, 96, 0, _p(zeros), _stream()) == 0
            return out
        both(fused)


def test_upsample_conv2d_and_time_conv():
    g = torch.Generator().manual_seed(3)
    T, H, W, C = 2, 6, 10, 192
    x = torch.randn(T, H, W, C, generator=g).half().to(DEV)
    w = (torch.randn(C // 2, C, 3, 3, generator=g) * (9 * C) ** -0.5).half().to(DEV)
    b = (torch.randn(C // 2, generator=g) * 0.1).half().to(DEV)
    out = _conv_cl(x, w, b, T, 2 * H, 2 * W, 1, 3, 3, ups=1)
    xin = x.permute(0, 3, 1, 2).float()
    ref = F.conv2d(F.interpolate(xin, scale_factor=(2.0, 2.0), mode="nearest"), w.float(), b.float(), padding=1)
    assert max_abs(out, ref.permute(0, 2, 3, 1)) <= 1e-2
    # r05: this layer runs on the halo-tile kernel (upsampling folded into the halo gather); RTV_CONV_GATHER keeps it on the
    # implicit-GEMM gather kernel - same math, another K order (isolated one-ulp fp16 differences); ragged sizes, all three
    # channel configurations of the decoder's Resample layers, repeated launches bit-identical
    for (Ci, Co, Tn, Hn, Wn) in ((384, 192, 1, 17, 23), (384, 192, 2, 8, 40), (192, 96, 3, 33, 70)):
        xx = (torch.randn(Tn, Hn, Wn, Ci, generator=g) * 0.7).half().to(DEV)
        ww = (torch.randn(Co, Ci, 3, 3, generator=g) * (9 * Ci) ** -0.5).half().to(DEV)
        bb = (torch.randn(Co, generator=g) * 0.1).half().to(DEV)
        outs = [_conv_cl(xx, ww, bb, Tn, 2 * Hn, 2 * Wn, 1, 3, 3, ups=1) for _ in range(3)]
        gat = _conv_cl(xx, ww, bb, Tn, 2 * Hn, 2 * Wn, 1, 3, 3, ups=1 | 16)
        rf = F.conv2d(F.interpolate(xx.permute(0, 3, 1, 2).float(), scale_factor=(2.0, 2.0), mode="nearest"), ww.float(), bb.float(),
                      padding=1).permute(0, 2, 3, 1)
        assert max_abs(outs[0], rf) <= 1e-2 and rel_l2(outs[0], rf) <= 2e-3 and max_abs(gat, rf) <= 1e-2
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        assert float((gat != outs[0]).float().mean()) <= 5e-3 and max_abs(gat, outs[0]) <= 4e-3
    # time_conv (3,1,1), C -> 2C, channel halves interleaved into frames (vae_block3.py:61-67)
    C = 384
    xc = torch.randn(T + 2, H, W, C, generator=g).half().to(DEV)
    wt = (torch.randn(2 * C, C, 3, 1, 1, generator=g) * (3 * C) ** -0.5).half().to(DEV)
    bt = (torch.randn(2 * C, generator=g) * 0.1).half().to(DEV)
    out = _conv_cl(xc, wt, bt, T, H, W, 3, 1, 1, n_split=C)
    y = F.conv3d(xc.permute(3, 0, 1, 2).unsqueeze(0).float(), wt.float(), bt.float())   # [1, 2C, T, H, W]
    y = y.reshape(1, 2, C, T, H, W)
    y = torch.stack((y[:, 0], y[:, 1]), 3).reshape(1, C, 2 * T, H, W)[0].permute(1, 2, 3, 0)
    assert out.shape == y.shape and max_abs(out, y) <= 1e-2


@pytest.mark.parametrize("Cin,T,H,W", [(96, 2, 33, 70), (192, 1, 16, 32), (96, 4, 48, 64), (384, 1, 5, 9)])
def test_conv_with_fused_rmsnorm_silu_epilogue(Cin, T, H, W):
    """rtv_conv3_norm_silu_cl: the first conv of a ResidualBlock with the RMS_norm * gamma + SiLU behind it
    (wan/modules/vae.py:39-54, :186-192) in the halo kernel's epilogue (96 filters = all channels of a pixel in one workgroup).
    Against the two separate launches on the same input: same arithmetic on the fp16-rounded conv output, only the fp32 order of
    the 96 squares differs - isolated one-ulp fp16 differences; against torch fp32; repeated launches bit-identical; layers the
    kernel does not take (192 filters, RTV_CONV_GATHER) return 1 and launch nothing."""
    from realtime_video_amd import _lib
    from realtime_video_amd.vae_decoder import pack_conv_weight
    Cout = 96
    g = torch.Generator().manual_seed(Cin + T + H)
    x = (torch.randn(T + 2, H, W, Cin, generator=g) * 0.7).half().to(DEV)
    w = (torch.randn(Cout, Cin, 3, 3, 3, generator=g) * (27 * Cin) ** -0.5).half().to(DEV)
    b = (torch.randn(Cout, generator=g) * 0.1).half().to(DEV)
    gamma = (1.0 + 0.2 * torch.randn(Cout, generator=g)).half().to(DEV)
    wp = pack_conv_weight(w).to(DEV)
    zeros = torch.zeros(64, dtype=torch.float16, device=DEV)
    lib = _lib.load()

    def fused(flags=0, cout=Cout, weight=wp):
        out = torch.full((T, H, W, cout), float("nan"), dtype=torch.float16, device=DEV)
        fn = lib.rtv_conv3_norm_silu_cl
        fn.argtypes = _lib.EXTRA_SIGNATURES["rtv_conv3_norm_silu_cl"]
        st = fn(_p(x), _p(weight), _p(b), _p(gamma), _p(out), cout, T, H, W, Cin, cout, flags, _p(zeros), _stream())
        return st, out

    conv = _conv_cl(x, w, b, T, H, W, 3, 3, 3)
    sep = torch.empty_like(conv)
    _lib.call("rtv_rmsnorm_silu_cl", _p(conv), _p(sep), _p(gamma), Cout, T * H * W, 1, _stream())
    st, a = fused()
    assert st == 0
    st, a2 = fused()
    assert st == 0 and torch.equal(a, a2) and torch.isfinite(a.float()).all()
    assert float((a != sep).float().mean()) <= 2e-3 and max_abs(a, sep) <= 4e-3
    xin = x.permute(3, 0, 1, 2).unsqueeze(0).float()
    y = F.conv3d(F.pad(xin, (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0].permute(1, 2, 3, 0)
    ref = F.silu(F.normalize(y, dim=-1) * Cout ** 0.5 * gamma.float())
    assert max_abs(a, ref) <= 2e-2 and rel_l2(a, ref) <= 3e-3
    st, untouched = fused(flags=16)                      # RTV_CONV_GATHER: not taken, nothing written
    assert st == 1 and torch.isnan(untouched.float()).all()
    _lib.call("rtv_conv_set_fuse_norm", 0)
    try:
        assert fused()[0] == 1
    finally:
        _lib.call("rtv_conv_set_fuse_norm", 1)


@pytest.mark.parametrize("C", [96, 192, 384])
def test_rmsnorm_silu_channels_last(C):
    from realtime_video_amd import _lib
    import realtime_video_amd.vae_decoder  # noqa: F401
    g = torch.Generator().manual_seed(C)
    npix = 1000
    x = (torch.randn(npix, C, generator=g) * 3).half().to(DEV)
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).half().to(DEV)
    for silu in (1, 0):
        out = torch.empty_like(x)
        _lib.call("rtv_rmsnorm_silu_cl", _p(x), _p(out), _p(gamma), C, npix, silu, _stream())
        ref = F.normalize(x.float(), dim=1) * C ** 0.5 * gamma.float()
        if silu:
            ref = F.silu(ref)
        assert max_abs(out, ref) <= 4e-3


def test_softmax_rows():
    from realtime_video_amd import _lib
    import realtime_video_amd.vae_decoder  # noqa: F401
    n, ldp = 96, 128
    s = (torch.randn(n, n) * 3).half().to(DEV)
    p = torch.full((n, ldp), 7.0, dtype=torch.float16, device=DEV)
    _lib.call("rtv_softmax_rows", _p(s), n, _p(p), ldp, n, n, _stream())
    assert max_abs(p[:, :n], torch.softmax(s.float(), -1)) <= 1e-3
    assert float(p[:, n:].abs().max()) == 0


def test_decoder_fused_norm_epilogue_vs_separate_pass():
    """The decoder with the conv + RMS_norm + SiLU fusion of the 96-channel ResidualBlocks (default) against the same decoder with
    the separate normalisation pass (rtv_conv_set_fuse_norm(0)): two streamed blocks at a small latent, pixels within 2e-3 of each
    other, mean difference below one fp16 ulp (the two forms differ by the fp32 summation order of a pixel's 96 squares: isolated
    one-ulp differences of the normalised activations, carried through the remaining layers)."""
    from oracle import vae_oracle as vo
    from realtime_video_amd import _lib
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    g = torch.Generator().manual_seed(5)
    zs = [torch.randn(1, 3, 16, 12, 20, generator=g).half().to(DEV) for _ in range(2)]
    outs = []
    for fuse in (1, 0):
        _lib.call("rtv_conv_set_fuse_norm", fuse)
        try:
            dec = VAEDecoderWrapper(DEV)
            dec.load_state_dict(vo.make_vae_weights(seed=0))
            cache = [None] * 55
            px = []
            for z in zs:
                p, cache = dec(z, *cache)
                px.append(p.clone())
            outs.append(px)
        finally:
            _lib.call("rtv_conv_set_fuse_norm", 1)
    for a, b in zip(*outs):
        assert a.shape == b.shape and torch.isfinite(a).all()
        assert max_abs(a, b) <= 2e-3 and float((a - b).abs().mean()) <= 3e-4      # (fp16 pixels: one ulp is 2.4e-4 .. 4.9e-4)
    assert not all(torch.equal(a, b) for a, b in zip(*outs))      # the fused path really ran


def test_streaming_decoder_matches_reference_golden(golden):
    from oracle import vae_oracle as vo
    from oracle.make_golden import vae_inputs
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    g = golden("vae_decoder.pt")
    dec = VAEDecoderWrapper(DEV)
    dec.load_state_dict(vo.make_vae_weights(seed=0))
    cache = [None] * 55
    # the reference's own precision: the same graph in eager fp16 on this GPU (what release_server.py runs)
    w16 = {k: v.half().to(DEV) for k, v in vo.make_vae_weights(seed=0).items()}
    cache16 = [None] * 55
    for i, z in enumerate(vae_inputs()):
        px, cache = dec(z.half().to(DEV), *cache)
        ref = g["pixels"][i]
        assert px.shape == ref.shape and px.dtype == torch.float32
        px16, cache16 = vo.decoder_wrapper_forward(w16, z.half().to(DEV), cache16)
        err, err16 = max_abs(px.cpu(), ref), max_abs(px16.cpu(), ref)
        mean_err = float((px.cpu() - ref).abs().mean())
        # stated tolerance: within 2x the eager-fp16 reference error (+ floor), hard cap 5e-2 on [-1, 1] pixels
        assert err <= max(2 * err16, 2e-2) and err <= 5e-2, (i, err, err16)
        assert mean_err <= 2e-3, (i, mean_err)
        assert float(px.abs().max()) <= 1.0
    assert len(cache) == 55 and sum(c is not None for c in cache) == 32
    for c, shp in zip(cache, g["cache_shapes"][-1]):
        if c is not None:
            assert tuple(c.shape) == shp
    for c, gs in zip(cache, g["cache_sample"]):
        if c is not None:
            assert rel_l2(c[0, ::7, :, ::3, ::5].float().cpu(), gs) <= 2e-2
    # a fresh stream restarts from the 9-frame first block
    px, _ = dec(vae_inputs()[0].half().to(DEV), *([None] * 55))
    assert px.shape[1] == 9 and max_abs(px.cpu(), g["pixels"][0]) <= 3e-2


def test_single_frame_decoder_wrapper_matches_reference_golden(golden):
    """`VAEDecoderWrapperSingle` (demo_utils/vae.py:150-195: one latent frame per call, `is_first_frame` given by the caller, the 32
    caches always tensors) against the golden minted from the reference's own module: first frame on zero caches - four frames out,
    the zero-interleaved temporal doubling - then two frames on the returned caches, one of them fed back as CLONED tensors.  Same
    stated tolerance as the block wrapper: within 2x the eager-fp16 oracle's error (+ floor), hard cap 5e-2, mean-abs 2e-3."""
    from oracle import vae_oracle as vo
    from oracle.make_golden import vae_inputs
    from realtime_video_amd.vae_decoder import VAEDecoderWrapperSingle
    g = golden("vae_decoder_single.pt")
    dec = VAEDecoderWrapperSingle(DEV)
    dec.load_state_dict(vo.make_vae_weights(seed=0))
    w16 = {k: v.half().to(DEV) for k, v in vo.make_vae_weights(seed=0).items()}
    zs = vae_inputs(seed=23)[0][:, :3]
    cache = dec.new_stream_cache(8, 12)
    arenas0, slot0 = len(dec._arenas._by_ptr), cache[0].data_ptr()      # new_stream_cache() hands out views of a registered arena ...
    cache16 = vo.single_zero_cache(8, 12, torch.float16, DEV)
    for i in range(3):
        first = torch.tensor([1.0 if i == 0 else 0.0], device=DEV, dtype=torch.float16)     # vae_torch2trt.py:167,174
        if i == 2:
            cache = [c.clone() for c in cache]
        px, cache = dec(zs[:, i:i + 1].half().to(DEV), first, *cache)
        if i == 0:                                                       # ... which the first call finds: no new arena, no copy-in
            assert len(dec._arenas._by_ptr) == arenas0 and cache[0].data_ptr() == slot0
        ref = g["pixels"][i]
        assert px.shape == ref.shape == (1, 4, 3, 64, 96) and px.dtype == torch.float32 and len(cache) == 32
        px16, cache16 = vo.decoder_single_forward(w16, zs[:, i:i + 1].half().to(DEV), i == 0, cache16)
        err, err16 = max_abs(px.cpu(), ref), max_abs(px16.float().cpu(), ref)
        assert err <= max(2 * err16, 2e-2) and err <= 5e-2, (i, err, err16)
        assert float((px.cpu() - ref).abs().mean()) <= 2e-3, i
        for c, gs, shp in zip(cache, g["cache_sample"][i], g["cache_shapes"]):
            assert tuple(c.shape) == shp
            assert max_abs(c[0, ::7, :, ::3, ::5].float().cpu(), gs) <= 2e-2 + 2e-2 * float(gs.abs().max()), i
    with pytest.raises(ValueError):
        dec(zs[:, :1].half().to(DEV), True, *([None] * 32))


def test_single_frame_decoder_zero_cache_is_a_reusable_constant():
    """The reference keeps the zero caches as a constant and starts EVERY stream from it (`feat_cache = ZERO_VAE_CACHE`,
    demo_utils/vae_torch2trt.py:168; demo_utils/constant.py:6-39).  `zero_cache()` therefore returns plain tensors that stay zero:
    two streams started from the same list give the same first-frame pixels bit for bit, the list is still all zeros afterwards,
    and the streams do not share state (an arena-backed list was updated in place by the first stream)."""
    from realtime_video_amd.vae_decoder import VAEDecoderWrapperSingle
    dec = VAEDecoderWrapperSingle(DEV).init_random_weights(seed=3)
    g = torch.Generator(device="cpu").manual_seed(6)
    z = [torch.randn(1, 1, 16, 8, 12, generator=g).half().to(DEV) for _ in range(3)]
    zero = dec.zero_cache(8, 12)
    px_a0, cache_a = dec(z[0], True, *zero)
    px_a1, cache_a = dec(z[1], False, *cache_a)
    assert all(float(c.abs().max()) == 0.0 for c in zero)                       # the constant is untouched
    px_b0, cache_b = dec(z[0], True, *zero)                                     # second stream from the SAME list
    assert torch.equal(px_a0, px_b0)
    assert cache_b[0].data_ptr() != cache_a[0].data_ptr()                       # its own arena
    px_b1, cache_b = dec(z[1], False, *cache_b)
    assert torch.equal(px_a1, px_b1)
    px_a2, _ = dec(z[2], False, *cache_a)                                       # stream a continues unaffected by stream b
    px_b2, _ = dec(z[2], False, *cache_b)
    assert torch.equal(px_a2, px_b2)
    one_shot = dec.new_stream_cache(8, 12)                                      # the in-place form gives the same first frame
    assert torch.equal(dec(z[0], True, *one_shot)[0], px_a0)


def test_cloned_feature_cache_continues_the_stream_and_foreign_cache_is_refused():
    """The cache list is a set of views into one arena that the next call updates in place.  A snapshot of it (every
    slot cloned, as a caller keeping state across requests would) must continue the stream bit-identically, and a cache
    of another frame size must be refused instead of decoded against the wrong arena."""
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    dec = VAEDecoderWrapper(DEV).init_random_weights(seed=3)
    g = torch.Generator(device="cpu").manual_seed(5)
    z0, z1 = (torch.randn(1, 3, 16, 8, 12, generator=g).half().to(DEV) for _ in range(2))
    _, cache = dec(z0, *([None] * 55))
    snap = [None if c is None else c.clone() for c in cache]
    px_a, cache_a = dec(z1, *cache)
    assert cache_a[0].data_ptr() == cache[0].data_ptr()          # same arena, updated in place
    assert not torch.equal(snap[1], cache[1])                    # ... so the old list now shows the new state
    px_b, cache_b = dec(z1, *snap)
    assert torch.equal(px_a, px_b)
    for a, b in zip(cache_a, cache_b):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
    px_c, _ = dec(z1, *cache_b)                                  # the rebuilt list is a registered arena again
    assert px_c.shape == px_a.shape
    _, other = dec(torch.zeros(1, 3, 16, 8, 20, dtype=torch.float16, device=DEV), *([None] * 55))
    with pytest.raises(ValueError):
        dec(z1, *other)
    with pytest.raises(ValueError):
        dec(z1, *[None if c is None else c.clone() for c in other])


# ------------------------------------------------------------------------------------------------ encoder
@pytest.mark.parametrize("C,T,H,W", [(96, 4, 16, 24), (192, 1, 8, 12), (384, 2, 8, 12)])
def test_downsample_conv2d_stride2(C, T, H, W):
    """ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2) of the encoder's Resample (wan/modules/vae.py:84-92)."""
    g = torch.Generator().manual_seed(C + T)
    x = torch.randn(T, 2 * H, 2 * W, C, generator=g).half().to(DEV)
    w = (torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5).half().to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).half().to(DEV)
    out = _conv_cl(x, w, b, T, H, W, 1, 3, 3, ups=2)
    ref = F.conv2d(F.pad(x.permute(0, 3, 1, 2).float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2)
    assert out.shape == (T, H, W, C) and max_abs(out, ref.permute(0, 2, 3, 1)) <= 1e-2


@pytest.mark.parametrize("C,T", [(192, 2), (384, 1)])
def test_time_conv_stride2(C, T):
    """Encoder time_conv (3,1,1) / stride (2,1,1) over [cached frame | 2T new frames] (vae.py:96, :151-156)."""
    g = torch.Generator().manual_seed(C)
    H, W = 6, 8
    x = torch.randn(2 * T + 1, H, W, C, generator=g).half().to(DEV)
    w = (torch.randn(C, C, 3, 1, 1, generator=g) * (3 * C) ** -0.5).half().to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).half().to(DEV)
    out = _conv_cl(x, w, b, T, H, W, 3, 1, 1, ups=3)
    ref = F.conv3d(x.permute(3, 0, 1, 2).unsqueeze(0).float(), w.float(), b.float(), stride=(2, 1, 1))[0]
    assert out.shape == (T, H, W, C) and max_abs(out, ref.permute(1, 2, 3, 0)) <= 1e-2


def test_conv1x1_k96_shortcut():
    """1x1x1 shortcut with Cin = 96 (encoder ResidualBlock 96 -> 192) runs through the conv kernel."""
    g = torch.Generator().manual_seed(5)
    T, H, W = 2, 8, 12
    x = torch.randn(T, H, W, 96, generator=g).half().to(DEV)
    w = (torch.randn(192, 96, 1, 1, 1, generator=g) * 96 ** -0.5).half().to(DEV)
    b = (torch.randn(192, generator=g) * 0.1).half().to(DEV)
    out = _conv_cl(x, w, b, T, H, W, 1, 1, 1)
    ref = x.float() @ w.float().reshape(192, 96).t() + b.float()
    assert max_abs(out, ref) <= 1e-2


def test_streaming_encoder_matches_reference_golden(golden):
    """VAEEncoderWrapper (native) vs the golden minted from the reference's VAEEncoderWrapper (fp32 CPU): fresh
    non-stream call (chunks 1 + 4), then stream=True on the returned cache (chunks 4 + 4).  Stated tolerance: fp16
    pipeline vs fp32 reference on normalised latents (|mu| ~ 0.2-1): max-abs <= max(2 x eager-fp16 error, 2e-2), cap
    5e-2; cache contents rel-L2 <= 2e-2."""
    from oracle import vae_oracle as vo
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    from test_vae_oracle_vs_golden import encoder_inputs
    g = golden("vae_encoder.pt")
    enc = VAEEncoderWrapper(device=DEV)
    enc.load_state_dict(vo.make_vae_encoder_weights(seed=1))
    w16 = {k: v.half().to(DEV) for k, v in vo.make_vae_encoder_weights(seed=1).items()}
    cache, cache16 = [None] * 55, [None] * 55
    for i, (f, stream) in enumerate(zip(encoder_inputs(), (False, True))):
        f16 = f.half().to(DEV)
        mu, cache = enc(f16, cache, stream=stream)
        mu16, cache16 = vo.encoder_wrapper_forward(w16, f16, cache16, stream=stream)
        ref = g["mu"][i]
        assert mu.shape == ref.shape and mu.dtype == torch.float16
        err, err16 = max_abs(mu.float().cpu(), ref), max_abs(mu16.float().cpu(), ref)
        assert err <= max(2 * err16, 2e-2) and err <= 5e-2, (i, err, err16)
        assert rel_l2(mu.float().cpu(), ref) <= 2e-2
        assert [None if c is None else tuple(c.shape) for c in cache] == g["cache_shapes"][i]
    assert sum(c is not None for c in cache) == 24
    for c, gs in zip(cache, g["cache_sample"]):
        if c is not None:
            assert rel_l2(c[0, ::7, :, ::3, ::5].float().cpu(), gs) <= 2e-2
    # single-frame encode on a fresh cache = the first-frame re-encode of release_server.py:572-575
    mu1, _ = enc(encoder_inputs()[0][:, :, :1].half().to(DEV), [None] * 55, stream=False)
    assert mu1.shape == (1, 16, 1, 8, 12) and max_abs(mu1.float().cpu(), g["mu"][0][:, :, :1]) <= 3e-2


@pytest.mark.parametrize("size", [(64, 96), (480, 832)])
def test_fresh_one_frame_encode_runs_the_last_time_tap_only_and_is_bit_identical(size):
    """r06: the first chunk of an encoder stream is ONE frame over zero caches (vae.py:17-36 pads two zero slices in front;
    release_server.py:572-575 runs exactly this once per block), so taps 0-17 of every causal 3x3x3 convolution multiply zeros.
    The native encoder runs the last time tap only (`conv3_last_tap`: the weight used in place through its row stride, the new
    slice as a 1x3x3 convolution, the same kernel per layer) - a third of the matrix work - and must give the SAME BITS as the
    full 27-tap launch over the zero slices (`rtv_vae_set_fresh_tap_skip(0)`): latents, every cache slot, and the streamed
    chunk that follows on those caches."""
    from realtime_video_amd import _lib
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    H, W = size
    g = torch.Generator().manual_seed(29)
    frames = (torch.rand(1, 3, 5, H, W, generator=g) * 2 - 1).half().to(DEV)
    outs = []
    try:
        for skip in (1, 0):
            _lib.call("rtv_vae_set_fresh_tap_skip", skip)
            enc = VAEEncoderWrapper(device=DEV).init_random_weights(seed=4)
            mu0, cache = enc(frames[:, :, :1], [None] * 55, stream=False)
            snap = [None if c is None else c.clone() for c in cache]
            mu1, cache = enc(frames[:, :, 1:5], cache, stream=True)
            outs.append((mu0.clone(), snap, mu1.clone(), [None if c is None else c.clone() for c in cache]))
    finally:
        _lib.call("rtv_vae_set_fresh_tap_skip", 1)
    (a0, ca, a1, cb), (b0, cc, b1, cd) = outs
    assert torch.isfinite(a0.float()).all() and float(a0.float().abs().max()) > 0
    assert torch.equal(a0, b0) and torch.equal(a1, b1)
    for x, y in list(zip(ca, cc)) + list(zip(cb, cd)):
        assert (x is None) == (y is None) and (x is None or torch.equal(x, y))


def test_encoder_rejects_cpu_and_bad_chunks():
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    enc = VAEEncoderWrapper(device=DEV).init_random_weights()
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, 3, 1, 64, 96, dtype=torch.float16), [None] * 55)
    mu, cache = enc(torch.zeros(1, 3, 1, 64, 96, dtype=torch.float16, device=DEV), [None] * 55)
    with pytest.raises(ValueError):   # non-stream continuation: the reference slices z[:, :, -3:1] here
        enc(torch.zeros(1, 3, 4, 64, 96, dtype=torch.float16, device=DEV), cache, stream=False)


# ------------------------------------------------------------------------------------------------ sharded decode
@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_sharded_decode_is_bit_identical(world):
    """rtv_vae_decode_rows: every rank's stripe (stage 0 replicated, stages 1-3 on row windows with conv halos) must
    equal the same rows of the unsharded decode bit for bit, across the first (9-frame) and later (12-frame) blocks."""
    from oracle import vae_oracle as vo
    from oracle.make_golden import vae_inputs
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    w = vo.make_vae_weights(seed=0)
    full = VAEDecoderWrapper(DEV)
    full.load_state_dict(w)
    shards = [VAEDecoderWrapper(DEV, row_shard=(r, world)) for r in range(world)]
    for s in shards:
        s.load_state_dict(w)
    cache = [None] * 55
    caches = [[None] * 55 for _ in range(world)]
    for z in vae_inputs(h=16, w=12)[:2]:
        zz = z.half().to(DEV)
        ref, cache = full(zz, *cache)
        parts = []
        for r, s in enumerate(shards):
            px, caches[r] = s(zz, *caches[r])
            r0, r1 = s.row_range(16)
            assert px.shape == (1, ref.shape[1], 3, r1 - r0, 96)
            parts.append(px)
        assert torch.equal(torch.cat(parts, dim=3), ref)


def test_gather_row_stripes_single_process():
    from realtime_video_amd.parallel import SimulatedContextParallel, gather_row_stripes
    cp = SimulatedContextParallel(1)
    x = torch.randn(2, 3, 16, 8, device=DEV)
    assert torch.equal(gather_row_stripes(cp, x, 16), x)


# ----------------------------------------------------------------------------------------- frame output format (8f-3)
def _special_pixels(T=3, H=48, W=64, seed=21):
    g = torch.Generator().manual_seed(seed)
    px = torch.randn(T, 3, H, W, generator=g) * 0.8
    flat = px.view(-1)
    ks = torch.arange(256, dtype=torch.float32)
    flat[:256] = ks / 255.0 * 2.0 - 1.0                      # values that land on (or next to) the k/255 boundaries
    flat[256:512] = torch.nextafter(flat[:256], torch.tensor(2.0))
    flat[512:768] = torch.nextafter(flat[:256], torch.tensor(-2.0))
    flat[768:776] = torch.tensor([-1.0, 1.0, 0.0, -0.0, 5.0, -5.0, 1.0000001, -1.0000001])
    return px


def test_pixels_to_rgb8_is_bit_exact_with_the_reference_frame_arithmetic():
    """rtv_pixels_to_rgb8 vs the oracle's restatement of release_server.py:984 + to_pil_image (bytes must be identical:
    they are what the JPEG encoder sees)."""
    from oracle import vae_oracle as vo
    from realtime_video_amd import ops
    px = _special_pixels()
    out = ops.pixels_to_rgb8(px.to(DEV))
    assert out.shape == (3, 48, 64, 3) and out.dtype == torch.uint8
    assert torch.equal(out.cpu(), vo.frames_to_rgb8(px))
    with pytest.raises(ValueError):
        ops.pixels_to_rgb8(px.to(DEV).half())


def test_frame_downloader_matches_reference_callback():
    """FrameDownloader as the session's frame_callback (pixels, frame_ids, event): pinned uint8 frames, one ticket per
    block, slots reused round-robin."""
    from oracle import vae_oracle as vo
    from realtime_video_amd.frames import FrameDownloader
    dl = FrameDownloader(DEV, slots=2)
    blocks = [_special_pixels(T=t, seed=s) for t, s in ((9, 1), (12, 2), (12, 3))]
    tickets = []
    for i, px in enumerate(blocks):
        gpu = (px.to(DEV) * 1.0).unsqueeze(0)                 # produced on the current stream
        ev = torch.cuda.Event()
        ev.record()
        tickets.append(dl(gpu, [f"id{i}"], ev))
        if i >= 1:                                            # 2 slots: the previous ticket is still held
            got = dl.fetch(tickets[i - 1])
            assert got.is_pinned() and torch.equal(got, vo.frames_to_rgb8(blocks[i - 1]))
    assert torch.equal(dl.fetch(tickets[-1]), vo.frames_to_rgb8(blocks[-1]))
    assert dl.frame_ids(tickets[-1]) == ["id2"]
    with pytest.raises(KeyError):
        dl.fetch(tickets[0])                                  # slot already reused


# ----------------------------------------------------------------------------------------- BASELINE sizes (480 x 832)
def test_conv3d_full_resolution_layer():
    """The decoder's dominant layer at its production size: causal 3x3x3 conv 96 -> 96 on 4 frames of 480 x 832 (two cached
    time slices in front) + bias + residual, against torch's fp32 conv3d on the GPU."""
    T, H, W, C = 4, 480, 832, 96
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(T + 2, H, W, C, generator=g) * 0.5).half().to(DEV)
    w = (torch.randn(C, C, 3, 3, 3, generator=g) * (27 * C) ** -0.5).half().to(DEV)
    b = (torch.randn(C, generator=g) * 0.1).half().to(DEV)
    res = torch.randn(T, H, W, C, generator=g).half().to(DEV)
    out = _conv_cl(x, w, b, T, H, W, 3, 3, 3, residual=res)
    xin = x.permute(3, 0, 1, 2).unsqueeze(0).float()
    ref = F.conv3d(F.pad(xin, (1, 1, 1, 1, 0, 0)), w.float(), b.float())[0].permute(1, 2, 3, 0)
    ref = ref.half().float() + res.float()
    assert max_abs(out, ref) <= 1e-2 and rel_l2(out, ref) <= 2e-3


def test_full_resolution_streaming_decode_matches_fp32_oracle():
    """The decoder at the benchmarked size against the ORACLE: latents 60 x 104 -> 480 x 832 pixels, a first call on fresh
    caches (2 latent frames -> 1 + 4 pixel frames) and a streamed second call on the returned caches (1 latent frame -> 4
    frames), vs `vae_oracle.decoder_wrapper_forward` in fp32.  The oracle graph is evaluated by torch on the GPU here (fp32
    eager, torch's native convolutions): on the host cores the same three latent frames cost 4-25 minutes depending on the box, and
    the CPU evaluation of this oracle is what the small-size tests pin to the reference golden
    (test_vae_oracle_vs_golden.py, test_streaming_decoder_matches_reference_golden).  Tolerances of the golden test above:
    max-abs within 2x the error of the same graph in eager fp16 (the reference's precision) with a 2e-2 floor, hard cap
    5e-2, mean-abs <= 2e-3 on [-1, 1] pixels."""
    from oracle import vae_oracle as vo
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    w = vo.make_vae_weights(seed=0)
    g = torch.Generator().manual_seed(21)
    zs = [torch.randn(1, 2, 16, 60, 104, generator=g), torch.randn(1, 1, 16, 60, 104, generator=g)]
    dec = VAEDecoderWrapper(DEV)
    dec.load_state_dict(w)
    w32 = {k: v.float().to(DEV) for k, v in w.items()}
    w16 = {k: v.half().to(DEV) for k, v in w.items()}
    cache, cache32, cache16 = [None] * 55, [None] * 55, [None] * 55
    for i, z in enumerate(zs):
        px, cache = dec(z.half().to(DEV), *cache)
        with torch.inference_mode():
            ref, cache32 = vo.decoder_wrapper_forward(w32, z.half().float().to(DEV), cache32)
            px16, cache16 = vo.decoder_wrapper_forward(w16, z.half().to(DEV), cache16)
        assert ref.dtype == torch.float32
        assert px.shape == ref.shape == (1, 5 if i == 0 else 4, 3, 480, 832) and px.dtype == torch.float32
        err, err16 = max_abs(px, ref), max_abs(px16.float(), ref)
        mean_err = float((px - ref).abs().mean())
        assert err <= max(2 * err16, 2e-2) and err <= 5e-2, (i, err, err16)
        assert mean_err <= 2e-3, (i, mean_err)
        assert float(px.abs().max()) <= 1.0
    for c, c32 in zip(cache, cache32):
        assert (c is None) == (c32 is None)
        if c is not None:
            assert tuple(c.shape) == tuple(c32.shape)
            assert rel_l2(c[0, ::5, :, ::7, ::9].float(), c32[0, ::5, :, ::7, ::9]) <= 2e-2


def test_full_resolution_streaming_encode_matches_fp32_oracle():
    """The ENCODER at the benchmarked size: what bench.py's timed region and the session run every
    block is a fresh single-frame encode of a 480 x 832 frame (the first-frame re-encode, release_server.py:572-575); webcam
    mode streams 4-frame chunks on the returned cache (demo_utils/vae_block3.py:116-175).  Both against
    `vae_oracle.encoder_wrapper_forward` evaluated in fp32 by torch on the device (its host evaluation is pinned to the
    reference golden at the small size by test_vae_oracle_vs_golden.py).  Tolerances of the golden test above: latents
    max-abs <= max(2 x the error of the same graph in eager fp16, 2e-2), hard cap 5e-2, rel-L2 <= 2e-2; cache contents
    rel-L2 <= 2e-2.  The fresh call is then repeated after the streamed one has dirtied every buffer: a dropped stream's arena
    is reused (no allocation, no registration: VERDICT r04 weak 8) and must give the same bits."""
    from oracle import vae_oracle as vo
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    w = vo.make_vae_encoder_weights(seed=1)
    g = torch.Generator().manual_seed(23)
    # smooth frames + noise in [-1, 1]: low-frequency content like video, plus pixel noise that exercises every tap
    base = F.interpolate(torch.rand(5, 3, 30, 52, generator=g) * 2 - 1, size=(480, 832), mode="bilinear", align_corners=False)
    frames = (0.8 * base + 0.2 * (torch.rand(5, 3, 480, 832, generator=g) * 2 - 1)).permute(1, 0, 2, 3).unsqueeze(0)
    enc = VAEEncoderWrapper(device=DEV)
    enc.load_state_dict(w)
    w32 = {k: v.float().to(DEV) for k, v in w.items()}
    w16 = {k: v.half().to(DEV) for k, v in w.items()}
    calls = [(frames[:, :, :1], False), (frames[:, :, 1:5], True)]
    cache, cache32, cache16 = [None] * 55, [None] * 55, [None] * 55
    first_mu = None
    for i, (f, stream) in enumerate(calls):
        f16 = f.half().to(DEV)
        mu, cache = enc(f16, cache, stream=stream)
        with torch.inference_mode():
            ref, cache32 = vo.encoder_wrapper_forward(w32, f16.float(), cache32, stream=stream)
            mu16, cache16 = vo.encoder_wrapper_forward(w16, f16, cache16, stream=stream)
        assert mu.shape == ref.shape == (1, 16, 1, 60, 104) and mu.dtype == torch.float16
        err, err16 = max_abs(mu.float(), ref), max_abs(mu16.float(), ref)
        assert err <= max(2 * err16, 2e-2) and err <= 5e-2, (i, err, err16)
        assert rel_l2(mu.float(), ref) <= 2e-2, (i, rel_l2(mu.float(), ref))
        if i == 0:
            first_mu = mu.clone()
    assert sum(c is not None for c in cache) == 24
    for c, c32 in zip(cache, cache32):
        assert (c is None) == (c32 is None)
        if c is not None:
            assert tuple(c.shape) == tuple(c32.shape)
            assert rel_l2(c[0, :, :, ::7, ::9].float(), c32[0, :, :, ::7, ::9]) <= 2e-2
    # a fresh one-shot encode on the wrapper's recycled arena: same bits as the first call, nothing newly registered
    del cache, c
    n_arenas = len(enc._arenas._by_ptr)
    for _ in range(3):
        again, _ = enc(frames[:, :, :1].half().to(DEV), [None] * 55, stream=False)
        assert torch.equal(again, first_mu)
    assert len(enc._arenas._by_ptr) <= max(n_arenas, 1)


def test_full_size_decode_row_sharded_equals_unsharded():
    """Streaming decode at the benchmarked size (latents 60 x 104 -> 480 x 832): first block (3 latent frames -> 9 pixel
    frames) and a streamed second block (12 frames); the 8 row stripes of the sharded decode (BASELINE config 4) are
    bit-identical to the rows of the unsharded one, pixels stay finite and inside [-1, 1]."""
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    full = VAEDecoderWrapper(DEV).init_random_weights(seed=1)
    g = torch.Generator().manual_seed(9)
    zs = [torch.randn(1, 3, 16, 60, 104, generator=g).half().to(DEV) for _ in range(2)]
    refs, cache = [], [None] * 55
    for z in zs:
        px, cache = full(z, *cache)
        assert torch.isfinite(px).all() and float(px.abs().max()) <= 1.0
        refs.append(px)
    assert refs[0].shape == (1, 9, 3, 480, 832) and refs[1].shape == (1, 12, 3, 480, 832)
    del full, cache
    for r in (0, 3, 7):                                                     # first, interior and last stripe
        shard = VAEDecoderWrapper(DEV, row_shard=(r, 8)).init_random_weights(seed=1)
        r0, r1 = shard.row_range(60)
        c = [None] * 55
        for z, ref in zip(zs, refs):
            px, c = shard(z, *c)
            assert torch.equal(px, ref[:, :, :, r0:r1])
        del shard, c


def test_wan_vae_wrapper_decode_to_pixel_matches_reference_golden(golden):
    """The pipeline's `vae.decode_to_pixel` (utils/wan_wrapper.py:95-118) through the native streaming decoder vs the golden
    minted from the reference's WanVAEWrapper (fp32): fp16 pipeline tolerance as for the streaming decoder; use_cache=True
    continues the stream like WanVAE_.cached_decode."""
    from oracle import vae_oracle as vo
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper, WanVAEWrapper
    gold = golden("wan_vae_wrapper.pt")
    w = vo.make_vae_weights(seed=0)
    vae = WanVAEWrapper(DEV).load_state_dict({"model." + k: v for k, v in w.items()})
    px = vae.decode_to_pixel(gold["z"].to(DEV), use_cache=False)
    assert px.shape == gold["pixels"].shape and px.dtype == torch.float32
    assert max_abs(px.cpu(), gold["pixels"]) <= 5e-2 and float((px.cpu() - gold["pixels"]).abs().mean()) <= 2e-3
    # cached decode == the streaming wrapper fed the same two calls
    ref = VAEDecoderWrapper(DEV)
    ref.load_state_dict(w)
    a, c = ref(gold["z"].to(DEV).half())
    b, _ = ref(gold["z"].to(DEV).half().flip(1), *c)
    assert torch.equal(vae.decode_to_pixel(gold["z"].to(DEV), use_cache=True), a)
    assert torch.equal(vae.decode_to_pixel(gold["z"].to(DEV).flip(1), use_cache=True), b)
    vae.clear_cache()
    assert torch.equal(vae.decode_to_pixel(gold["z"].to(DEV), use_cache=True), a)
    # encode_to_latent (utils/wan_wrapper.py:79-93) through the native streaming encoder
    we = vo.make_vae_encoder_weights(seed=1)
    vae.load_state_dict({"model." + k: v for k, v in {**w, **we}.items()})
    lat = vae.encode_to_latent(gold["frames"].to(DEV))
    assert lat.shape == gold["latents"].shape and lat.dtype == torch.float32
    assert max_abs(lat.cpu(), gold["latents"]) <= 3e-2 and rel_l2(lat.cpu(), gold["latents"]) <= 1e-2
