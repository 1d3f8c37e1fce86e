"""Pins the CPU VAE-decoder oracle (oracle/vae_oracle.py) against goldens minted from the upstream
reference's VAEDecoderWrapper (oracle/make_golden.py golden_vae).  CPU only."""
import torch

from conftest import max_abs

from oracle import vae_oracle as vo
from oracle.make_golden import vae_inputs


def test_vae_decoder_streaming_matches_reference(golden):
    g = golden("vae_decoder.pt")
    w = vo.make_vae_weights(seed=0)
    cs = float(sum(v.double().abs().sum() for v in w.values()))
    assert abs(cs - g["weights_checksum"]) <= 1e-6 * g["weights_checksum"]
    cache = [None] * 55
    for i, z in enumerate(vae_inputs()):
        px, cache = vo.decoder_wrapper_forward(w, z, cache)
        assert px.shape == g["pixels"][i].shape and px.dtype == torch.float32
        assert max_abs(px, g["pixels"][i]) <= 1e-4, i
        shapes = [None if c is None else tuple(c.shape) for c in cache]
        assert shapes == g["cache_shapes"][i]
    assert [p.shape[1] for p in g["pixels"]] == [9, 12, 12]   # first block: 1 + 4 + 4 frames
    assert sum(c is not None for c in cache) == 32             # 32 cached convs (demo_utils/constant.py:6-39)
    for c, gs in zip(cache, g["cache_sample"]):
        if c is not None:
            assert max_abs(c[0, ::7, :, ::3, ::5], gs) <= 1e-4


def test_vae_decoder_single_frame_form_matches_reference(golden):
    """oracle decoder_single_forward vs the reference's VAEDecoderWrapperSingle (demo_utils/vae.py:150-195): first frame on zero
    caches (4 frames: the zero-interleaved temporal doubling), then two frames on the returned caches."""
    g = golden("vae_decoder_single.pt")
    w = vo.make_vae_weights(seed=0)
    zs = vae_inputs(seed=23)[0][:, :3]
    cache = vo.single_zero_cache(8, 12)
    for i in range(3):
        px, cache = vo.decoder_single_forward(w, zs[:, i:i + 1], i == 0, cache)
        assert px.shape == g["pixels"][i].shape == (1, 4, 3, 64, 96)
        assert max_abs(px, g["pixels"][i]) <= 1e-4, i
        assert len(cache) == 32
        for c, gs in zip(cache, g["cache_sample"][i]):
            assert max_abs(c[0, ::7, :, ::3, ::5], gs) <= 1e-4, i
    assert [tuple(c.shape) for c in cache] == g["cache_shapes"]
    # the two wrappers are different graphs on the first frame: the single form's real frame sees a bias-only frame behind it
    px3, _ = vo.decoder_wrapper_forward(w, zs[:, :1], [None] * 55)
    assert px3.shape[1] == 1 and max_abs(px3[:, 0], g["pixels"][0][:, 3]) > 1e-3


def test_decoder_conv_inventory():
    specs = vo.decoder_conv_specs()
    assert sum(k in ("c3", "t3") for _, k, _, _ in specs) == 32 and sum(k == "c1" for _, k, _, _ in specs) == 1


def encoder_inputs():
    """Same frames as oracle/make_golden.py golden_vae_encoder (seed 33, pixels in [-1, 1])."""
    g = torch.Generator().manual_seed(33)
    return [torch.rand(1, 3, 5, 64, 96, generator=g) * 2 - 1, torch.rand(1, 3, 8, 64, 96, generator=g) * 2 - 1]


def test_vae_encoder_streaming_matches_reference(golden):
    """Encoder oracle vs the reference's VAEEncoderWrapper: fresh non-stream call (chunks 1 + 4), then a stream=True
    call on the returned cache (chunks 4 + 4)."""
    g = golden("vae_encoder.pt")
    w = vo.make_vae_encoder_weights(seed=1)
    cs = float(sum(v.double().abs().sum() for v in w.values()))
    assert abs(cs - g["weights_checksum"]) <= 1e-6 * g["weights_checksum"]
    frames = encoder_inputs()
    cache = [None] * 55
    for i, (f, stream) in enumerate(zip(frames, (False, True))):
        mu, cache = vo.encoder_wrapper_forward(w, f, cache, stream=stream)
        assert mu.shape == g["mu"][i].shape == (1, 16, 2, 8, 12)
        assert max_abs(mu, g["mu"][i]) <= 1e-4, i
        assert [None if c is None else tuple(c.shape) for c in cache] == g["cache_shapes"][i]
    assert sum(c is not None for c in cache) == 24   # 22 cached 3x3x3 convs + 2 downsample3d frame caches
    for c, gs in zip(cache, g["cache_sample"]):
        if c is not None:
            assert max_abs(c[0, ::7, :, ::3, ::5], gs) <= 1e-4


def test_encoder_conv_inventory():
    specs = vo.encoder_conv_specs()
    kinds = [k for _, k, _, _ in specs]
    assert kinds.count("c3") == 22 and kinds.count("d2") == 3 and kinds.count("t3") == 2 and kinds.count("c1") == 2


def test_frames_to_rgb8_known_values():
    """Frame output format (release_server.py:984 + to_pil_image): u8(trunc(clamp((x+1)/2, 0, 1) * 255)), H x W x C."""
    from oracle import vae_oracle as vo
    x = torch.tensor([-1.0, 1.0, 0.0, 5.0, -5.0, 0.5, -0.5, 1 / 255.0]).view(1, 1, 8).repeat(3, 1, 1)   # [3, 1, 8]
    x[1] = -x[1]
    out = vo.frames_to_rgb8(x)
    assert out.shape == (1, 8, 3) and out.dtype == torch.uint8
    assert out[0, :, 0].tolist() == [0, 255, 127, 255, 0, 191, 63, 128]
    assert out[0, :, 1].tolist() == [255, 0, 127, 0, 255, 63, 191, 127]
    assert torch.equal(out[0, :, 2], out[0, :, 0])


def test_whole_sequence_decode_equals_streaming_decode_from_fresh_caches(golden):
    """WanVAEWrapper.decode_to_pixel (the whole-sequence decode of pipeline.inference's Step 4: WanVAE_.decode, frame by frame
    over a cleared cache) vs the streaming decoder oracle called once on fresh caches: the same computation, which is why the
    native WanVAEWrapper facade delegates to the streaming decoder."""
    from oracle import vae_oracle as vo
    gold = golden("wan_vae_wrapper.pt")
    w = vo.make_vae_weights(seed=0)
    px, _ = vo.decoder_wrapper_forward(w, gold["z"], [None] * 55)
    assert px.shape == gold["pixels"].shape == (1, 9, 3, 64, 96)
    assert torch.allclose(px, gold["pixels"], atol=2e-5, rtol=1e-5)


def test_whole_sequence_encode_equals_non_streamed_encoder_call(golden):
    """WanVAEWrapper.encode_to_latent (WanVAE_.encode: chunks 1, 4, 4 over a cleared cache) vs the encoder-wrapper oracle
    called once, non-streamed, on fresh caches."""
    from oracle import vae_oracle as vo
    gold = golden("wan_vae_wrapper.pt")
    w = vo.make_vae_encoder_weights(seed=1)
    mu, _ = vo.encoder_wrapper_forward(w, gold["frames"], [None] * 55, stream=False)
    assert gold["latents"].shape == (1, 3, 16, 8, 12)
    assert torch.allclose(mu.permute(0, 2, 1, 3, 4), gold["latents"], atol=2e-5, rtol=1e-5)
