"""ORACLE SUPPORT (test infrastructure only) — mint golden vectors from the upstream reference.

Run in the authoring container (needs /root/reference):   python oracle/make_golden.py
Writes tests/golden/*.pt.  Inputs are regenerated from fixed seeds by the tests (only outputs and the
few inputs that are cheap to keep are stored).  The reference modules run on CPU in bf16 through
oracle/ref_shim.py — its SDPA fallback (attention.py:197-212) and the flex_attention recompute
branch (causal_model.py:305-348, Inductor C++ backend) are the only reference backends that exist
off-NVIDIA.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_shim, wan_oracle as wo  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

TINY = dict(dim=256, ffn_dim=512, num_heads=2, num_layers=2, freq_dim=256, text_len=512, eps=1e-6,
            num_frame_per_block=3)
TEXT_DIM = 128
GRID = (3, 30, 52)  # 3 latent frames of 60x104 -> 1560 tokens/frame (the only size the reference supports)


def tiny_inputs(seed=42):
    g = torch.Generator().manual_seed(seed)
    lat = [torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16) for _ in range(4)]
    ctx = torch.randn(64, TEXT_DIM, generator=g).to(torch.bfloat16)
    return lat, ctx


def fresh_caches(cfg, kv_size, dtype=torch.bfloat16):
    hd = cfg["dim"] // cfg["num_heads"]
    return (wo.initialize_kv_cache(cfg["num_layers"], 1, kv_size, cfg["num_heads"], hd, dtype),
            wo.initialize_crossattn_cache(cfg["num_layers"], 1, cfg["num_heads"], hd, dtype))


def cache_sample(kv_cache):
    return [{"k": c["k"][0, ::197].clone(), "v": c["v"][0, ::197].clone(),
             "global_end_index": int(c["global_end_index"]), "local_end_index": int(c["local_end_index"])}
            for c in kv_cache]


def golden_ops(ref):
    g = torch.Generator().manual_seed(1)
    out = {}
    q = torch.randn(1, 300, 2, 128, generator=g).to(torch.bfloat16)
    k = torch.randn(1, 500, 2, 128, generator=g).to(torch.bfloat16)
    v = torch.randn(1, 500, 2, 128, generator=g).to(torch.bfloat16)
    out["attn_q"], out["attn_k"], out["attn_v"] = q, k, v
    out["attn_out"] = ref.attention.attention(q, k, v)
    # RoPE (causal, start frame 3) and non-causal
    x = torch.randn(1, 2 * 6 * 8, 2, 128, generator=g).to(torch.bfloat16)
    freqs = ref.cm.CausalWanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=TEXT_DIM).freqs
    grid = torch.tensor([[2, 6, 8]])
    out["rope_x"] = x
    out["rope_causal_s3"] = ref.cm.causal_rope_apply(x, grid, freqs, start_frame=3)
    out["rope_s0"] = ref.model.rope_apply(x, grid, freqs)
    out["freqs_sample"] = torch.view_as_real(freqs[[0, 1, 5, 100, 1023]]).clone()
    # norms
    y = torch.randn(4, 7, 256, generator=g).to(torch.bfloat16)
    wgt = (1 + 0.1 * torch.randn(256, generator=g)).to(torch.bfloat16)
    rn = ref.model.WanRMSNorm(256, eps=1e-6).to(torch.bfloat16)
    rn.weight.data.copy_(wgt)
    out["norm_x"], out["norm_w"] = y, wgt
    out["rms"] = rn(y)
    out["ln"] = ref.model.WanLayerNorm(256, eps=1e-6)(y)
    # scheduler / schedule / x0
    sch = ref.scheduler.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    sch.set_timesteps(1000, training=True)
    out["sched_timesteps"] = sch.timesteps.clone()
    out["sched_sigmas"] = sch.sigmas.clone()
    zp = torch.cat((sch.timesteps, torch.tensor([0], dtype=torch.float32)))
    out["schedule_4"] = ref.get_denoising_schedule(zp, 1.0, 4)
    out["schedule_5"] = ref.get_denoising_schedule(zp, 1.0, 5)
    out["schedule_4_s07"] = ref.get_denoising_schedule(zp, 0.7, 4)
    x0 = torch.randn(3, 16, 6, 8, generator=g).to(torch.bfloat16)
    nz = torch.randn(3, 16, 6, 8, generator=g).to(torch.bfloat16)
    tt = out["schedule_4"][1] * torch.ones([3], dtype=torch.long)
    out["an_x0"], out["an_noise"], out["an_t"] = x0, nz, tt
    out["an_out"] = sch.add_noise(x0, nz, tt)
    W = ref.wan_wrapper.WanDiffusionWrapper
    wr = W.__new__(W)
    torch.nn.Module.__init__(wr)
    wr.scheduler = sch
    out["x0_out"] = wr._convert_flow_pred_to_x0(x0, nz, tt.to(torch.float32))
    out["sinus"] = ref.cm.sinusoidal_embedding_1d(256, out["schedule_4"])
    torch.save(out, os.path.join(OUT, "ops.pt"))
    print("ops.pt", sum(v.numel() * v.element_size() for v in out.values() if torch.is_tensor(v)) / 1e6, "MB")


def golden_dit(ref):
    """Server-path sequence on the tiny model, c = 3."""
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    lat, ctx = tiny_inputs()
    model = build = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
    wr = ref_shim.build_reference_wrapper(ref, model)
    kv, ca = fresh_caches(cfg, 9360)
    cond = {"prompt_embeds": [ctx]}
    sched = wr.scheduler
    zp = torch.cat((sched.timesteps, torch.tensor([0], dtype=torch.float32)))
    steps = ref.get_denoising_schedule(zp, 1.0, 4)
    out = {"weights_checksum": float(sum(v.double().abs().sum() for v in w.values())), "steps": steps}

    def ts(v):
        return torch.ones([1, 3], dtype=torch.int64) * v

    with torch.inference_mode():
        # block 0, step 0 and step 1 at current_start = 0 (second call overwrites the same rows)
        flow, x0 = wr(lat[0], cond, ts(steps[0]), kv, ca, current_start=0)
        out["b0s0_flow"], out["b0s0_x0"] = flow.clone(), x0.clone()
        out["b0s0_cache"] = cache_sample(kv)
        flow, x0 = wr(lat[1], cond, ts(steps[1]), kv, ca, current_start=0)
        out["b0s1_flow"] = flow.clone()
        out["b0s1_cache"] = cache_sample(kv)
        # recompute (flex branch): zero caches, block mask, t = 0, clean frames = lat[2]
        for c in kv:
            c["k"].zero_()
            c["v"].zero_()
            c["global_end_index"] = 0
            c["local_end_index"] = 0
        model.block_mask = model._prepare_blockwise_causal_attn_mask(
            device="cpu", num_frames=3, frame_seqlen=1560, num_frame_per_block=3, local_attn_size=-1)
        flow, _ = wr(lat[2], cond, torch.zeros([1, 3], dtype=torch.int64), kv, ca, current_start=3 * 1560)
        model.block_mask = None
        out["rc_flow"] = flow.clone()
        out["rc_cache"] = cache_sample(kv)
        # block 1 denoise at current_start = 4680 -> attends 9360 keys
        flow, x0 = wr(lat[3], cond, ts(steps[0]), kv, ca, current_start=4680)
        out["b1s0_flow"], out["b1s0_x0"] = flow.clone(), x0.clone()
        out["b1s0_cache"] = cache_sample(kv)
    torch.save(out, os.path.join(OUT, "dit_server_path.pt"))
    print("dit_server_path.pt done")


def golden_rolling(ref):
    """Rolling cache with attention sink (causal_model.py:359-385): local_attn_size=6, sink_size=1."""
    cfg = dict(TINY, local_attn_size=6, sink_size=1, num_layers=1)
    w = wo.make_weights(cfg, seed=3, text_dim=TEXT_DIM)
    lat, ctx = tiny_inputs(seed=7)
    model = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
    wr = ref_shim.build_reference_wrapper(ref, model)
    kv, ca = fresh_caches(cfg, 6 * 1560)
    cond = {"prompt_embeds": [ctx]}
    out = {"indices": [], "flow_sample": []}
    with torch.inference_mode():
        for b in range(4):
            t = torch.ones([1, 3], dtype=torch.int64) * 500
            flow, _ = wr(lat[b], cond, t, kv, ca, current_start=b * 4680)
            out["indices"].append((int(kv[0]["global_end_index"]), int(kv[0]["local_end_index"])))
            out["flow_sample"].append(flow[0, :, :, ::3, ::4].clone())
        out["cache"] = cache_sample(kv)
    torch.save(out, os.path.join(OUT, "dit_rolling.pt"))
    print("dit_rolling.pt", out["indices"])


def vae_inputs(seed=21, h=8, w=12):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(1, 3, 16, h, w, generator=g) for _ in range(3)]


def golden_vae(ref):
    """Streaming VAE decoder (demo_utils/vae_block3.py:177-230), fp32 on CPU, 8x12 latents -> 64x96 px:
    block 0 yields 9 frames (first latent frame skips the temporal upsampling), later blocks 12."""
    from oracle import vae_oracle as vo
    w = vo.make_vae_weights(seed=0)
    dec = ref.vae_block3.VAEDecoderWrapper().eval()
    missing, unexpected = dec.load_state_dict(w, strict=False)
    assert not unexpected and set(missing) <= {"mean", "std"}, (missing, unexpected)
    zs = vae_inputs()
    cache = [None] * 55
    out = {"weights_checksum": float(sum(v.double().abs().sum() for v in w.values())), "pixels": [],
           "cache_shapes": []}
    with torch.inference_mode():
        for z in zs:
            px, cache = dec(z, *cache)
            out["pixels"].append(px.clone())
            out["cache_shapes"].append([None if c is None else tuple(c.shape) for c in cache])
        out["cache_sample"] = [None if c is None else c[0, ::7, :, ::3, ::5].clone() for c in cache]
    torch.save(out, os.path.join(OUT, "vae_decoder.pt"))
    print("vae_decoder.pt", [tuple(p.shape) for p in out["pixels"]])


def golden_vae_single():
    """VAEDecoderWrapperSingle (demo_utils/vae.py:150-195, the one-latent-frame form): first frame on the zero caches of
    demo_utils/constant.py, then two more frames on the returned caches; fp32 on CPU, 8x12 latents -> 4 frames of 64x96 per call."""
    sys.path.insert(0, ref_shim.REF)
    import demo_utils.vae as dv
    from oracle import vae_oracle as vo
    w = vo.make_vae_weights(seed=0)
    dec = dv.VAEDecoderWrapperSingle().eval()
    missing, unexpected = dec.load_state_dict(w, strict=False)
    assert not unexpected and set(missing) <= {"mean", "std"}, (missing, unexpected)
    zs = vae_inputs(seed=23)[0][:, :3]
    cache = [torch.zeros(1, t.shape[1], 2, 8 * t.shape[3] // 60, 12 * t.shape[4] // 104) for t in dv.ZERO_VAE_CACHE]
    assert [tuple(c.shape) for c in cache] == [tuple(c.shape) for c in vo.single_zero_cache(8, 12)]
    out = {"weights_checksum": float(sum(v.double().abs().sum() for v in w.values())), "pixels": [], "cache_sample": []}
    with torch.no_grad():
        for i in range(3):
            px, cache = dec(zs[:, i:i + 1], torch.tensor([1.0 if i == 0 else 0.0]), *cache)
            out["pixels"].append(px.clone())
            out["cache_sample"].append([c[0, ::7, :, ::3, ::5].clone() for c in cache])
    out["cache_shapes"] = [tuple(c.shape) for c in cache]
    torch.save(out, os.path.join(OUT, "vae_decoder_single.pt"))
    print("vae_decoder_single.pt", [tuple(p.shape) for p in out["pixels"]], len(cache))


def golden_vae_encoder(ref):
    """Streaming VAE encoder (VAEEncoderWrapper, demo_utils/vae_block3.py:116-175, over wan/modules/vae.py's
    WanVAE_ encoder + conv1), fp32 on CPU, 64x96 px frames -> 8x12 latents.  Call 1: fresh cache, non-stream, 5 frames
    (chunks 1 + 4 -> 2 latent frames; this is also the single-frame first-frame re-encode of release_server.py:574 for
    its first chunk).  Call 2: stream=True on the returned cache, 8 frames (chunks 4 + 4 -> 2 latent frames)."""
    import types
    from oracle import vae_oracle as vo
    w = vo.make_vae_encoder_weights(seed=1)
    model = ref.vae.WanVAE_(dim=96, z_dim=16, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                            temperal_downsample=[False, True, True], dropout=0.0).eval()
    sd = {k: v for k, v in model.state_dict().items() if k.startswith("encoder.") or k.startswith("conv1.")}
    assert set(sd) == set(w), (sorted(set(sd) ^ set(w))[:8])
    model.load_state_dict(w, strict=False)
    enc = ref.vae_block3.VAEEncoderWrapper(types.SimpleNamespace(model=model)).eval()
    g = torch.Generator().manual_seed(33)
    frames = [torch.rand(1, 3, 5, 64, 96, generator=g) * 2 - 1, torch.rand(1, 3, 8, 64, 96, generator=g) * 2 - 1]
    out = {"weights_checksum": float(sum(v.double().abs().sum() for v in w.values())), "mu": [], "cache_shapes": []}
    with torch.inference_mode():
        mu, cache = enc(frames[0], [None] * 55, stream=False)
        out["mu"].append(mu.clone())
        out["cache_shapes"].append([None if c is None else tuple(c.shape) for c in cache])
        mu, cache = enc(frames[1], cache, stream=True)
        out["mu"].append(mu.clone())
        out["cache_shapes"].append([None if c is None else tuple(c.shape) for c in cache])
        out["cache_sample"] = [None if c is None else c[0, ::7, :, ::3, ::5].clone() for c in cache]
    torch.save(out, os.path.join(OUT, "vae_encoder.pt"))
    print("vae_encoder.pt", [tuple(m.shape) for m in out["mu"]], sum(c is not None for c in cache), "cache slots")


def golden_session():
    """The reference's own GenerationSession (release_server.py:344-736) driven for 3 blocks on the CPU: tiny DiT through the
    reference's CausalWanModel / WanDiffusionWrapper / CausalInferencePipeline, c = 3, 4 steps, keep_first_frame = False (block 2
    takes the first-frame re-encode branch, :572-575), stand-in VAE / text encoder (oracle/standins.py).  Pins the session
    orchestration of oracle.wan_oracle.SessionOracle and of the native session mirror."""
    import types
    from oracle import standins
    rs, CIP = ref_shim.load_release_server()
    ref = ref_shim.load()
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    model = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
    # `.config` is diffusers' register_to_config sugar (stubbed away by the shim); the pipeline reads two fields of it
    model.config = types.SimpleNamespace(num_heads=cfg["num_heads"], dim=cfg["dim"])
    wr = ref_shim.build_reference_wrapper(ref, model)
    g = torch.Generator().manual_seed(5)
    prompt = torch.zeros(1, 512, TEXT_DIM, dtype=torch.bfloat16)
    prompt[0, :64] = torch.randn(64, TEXT_DIM, generator=g).to(torch.bfloat16)
    args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=False, num_frame_per_block=3,
                                 independent_first_frame=False)
    pipe = CIP(args, "cpu", generator=wr, text_encoder=standins.StandinTextEncoder(prompt), vae=object())
    enc_calls = []

    def encoder(frames, cache, stream=False):
        enc_calls.append(frames.float().clone())
        return standins.standin_encoder(frames, cache, stream)

    models = rs.Models(standins.StandinTextEncoder(prompt), wr, pipe, encoder, standins.standin_decoder)
    params = rs.GenerateParams(prompt="synthetic", seed=9, num_blocks=3, num_denoising_steps=4, kv_cache_num_frames=3,
                               keep_first_frame=False)
    sess = rs.GenerationSession(params, types.SimpleNamespace(use_taehv=False), frame_callback=lambda *a, **k: None, models=models)
    out = {"noise": sess.noise.clone(), "steps": sess.denoising_step_list.clone(), "prompt": prompt, "blocks": [], "indices": [],
           "pixels_shape": [], "pixel_sample": []}
    for b in range(3):
        px = sess.generate_block_internal(models)
        out["blocks"].append(sess.last_pred.clone())
        out["indices"].append((int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"]),
                               sess.current_start_frame, sess.block_idx, sess.total_frames_sent))
        out["pixels_shape"].append(tuple(px.shape))
        out["pixel_sample"].append(px[0, :, :, ::40, ::52].clone())
    out["all_latents"] = sess.all_latents.clone()
    out["encoder_inputs"] = [f[..., ::40, ::52].clone() for f in enc_calls]
    out["encoder_input_shapes"] = [tuple(f.shape) for f in enc_calls]
    out["kv_shape"] = tuple(pipe.kv_cache1[0]["k"].shape)
    torch.save(out, os.path.join(OUT, "session_reference.pt"))
    print("session_reference.pt", out["indices"], out["pixels_shape"], out["encoder_input_shapes"])


def start_image(seed=41):
    """The start frame of the image-to-video golden: a smooth 832x480 8-bit RGB image."""
    import numpy as np
    from PIL import Image
    low = torch.rand(1, 3, 15, 26, generator=torch.Generator().manual_seed(seed))
    img = torch.nn.functional.interpolate(low, size=(480, 832), mode="bilinear")[0]
    return Image.fromarray((img.permute(1, 2, 0) * 255).round().to(torch.uint8).numpy(), "RGB")


def golden_session_start_frame():
    """The reference's GenerationSession started from an image (params.start_frame -> setup_start_frame,
    release_server.py:429-431, :578-586; block 0 resumes from the encoded frames, :590-595): 2 blocks behind the 3 encoded
    latent frames, c = 3, 4 steps, stand-in VAE / text encoder."""
    import types
    from oracle import standins
    rs, CIP = ref_shim.load_release_server()
    ref = ref_shim.load()
    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return orig_to(self, *a, **k)
    torch.Tensor.to = to_cpu
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    model = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
    model.config = types.SimpleNamespace(num_heads=cfg["num_heads"], dim=cfg["dim"])
    wr = ref_shim.build_reference_wrapper(ref, model)
    g = torch.Generator().manual_seed(5)
    prompt = torch.zeros(1, 512, TEXT_DIM, dtype=torch.bfloat16)
    prompt[0, :64] = torch.randn(64, TEXT_DIM, generator=g).to(torch.bfloat16)
    text = standins.StandinTextEncoder(prompt)
    args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=False, num_frame_per_block=3,
                                 independent_first_frame=False)
    pipe = CIP(args, "cpu", generator=wr, text_encoder=text, vae=object())
    enc_calls = []

    def encoder(frames, cache, stream=False):
        enc_calls.append((tuple(frames.shape), bool(stream), frames.float()[..., ::40, ::52].clone()))
        return standins.standin_encoder(frames, cache, stream)

    models = rs.Models(text, wr, pipe, encoder, standins.standin_decoder)
    params = rs.GenerateParams(prompt="synthetic", seed=9, num_blocks=3, num_denoising_steps=4, kv_cache_num_frames=3,
                               keep_first_frame=False)
    params.start_frame = start_image()        # the server assigns the decoded PIL image the same way (:944-946)
    sess = rs.GenerationSession(params, types.SimpleNamespace(use_taehv=False), frame_callback=lambda *a, **k: None, models=models)
    out = {"prompt": prompt, "noise": sess.noise.clone(), "resume_latents": sess.resume_latents.clone(), "blocks": [], "indices": []}
    for b in range(2):
        sess.generate_block_internal(models)
        out["blocks"].append(sess.last_pred.clone())
        out["indices"].append((int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"]),
                               sess.current_start_frame, sess.block_idx, sess.total_frames_sent))
    out["all_latents"] = sess.all_latents.clone()
    out["encoder_calls"] = enc_calls
    torch.Tensor.to = orig_to
    torch.save(out, os.path.join(OUT, "session_start_frame_reference.pt"))
    print("session_start_frame_reference.pt", out["indices"], tuple(out["resume_latents"].shape), [(c[0], c[1]) for c in enc_calls])


def v2v_video(seed=51, T=33):
    """The input video of the offline v2v golden: [T, 3, 480, 832] float32 in [-1, 1], smooth in space."""
    low = torch.rand(T, 3, 15, 26, generator=torch.Generator().manual_seed(seed)) * 2 - 1
    return torch.nn.functional.interpolate(low, size=(480, 832), mode="bilinear")


def golden_session_v2v():
    """The reference's GenerationSession with `input_video` (offline video-to-video, release_server.py:417-428, :529-540;
    v2v.py:138-158): the decoded video (33 frames; v2v.load_video_as_rgb - cv2 / ffmpeg file decoding, control plane - is
    replaced by the in-memory frames) is encoded once, noised to the first step's level with the session generator, and
    bounds the block count (9 latent frames -> 2 blocks); strength 0.6, 4 steps."""
    import types
    from oracle import standins
    rs, CIP = ref_shim.load_release_server()
    ref = ref_shim.load()
    video = v2v_video()
    sys.modules["v2v"].load_video_as_rgb = lambda path, **k: video.clone()
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    model = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
    model.config = types.SimpleNamespace(num_heads=cfg["num_heads"], dim=cfg["dim"])
    wr = ref_shim.build_reference_wrapper(ref, model)
    g = torch.Generator().manual_seed(5)
    prompt = torch.zeros(1, 512, TEXT_DIM, dtype=torch.bfloat16)
    prompt[0, :64] = torch.randn(64, TEXT_DIM, generator=g).to(torch.bfloat16)
    text = standins.StandinTextEncoder(prompt)
    args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=False, num_frame_per_block=3,
                                 independent_first_frame=False)
    pipe = CIP(args, "cpu", generator=wr, text_encoder=text, vae=object())
    enc_calls = []

    def encoder(frames, cache, stream=False):
        enc_calls.append((tuple(frames.shape), bool(stream)))
        return standins.standin_encoder(frames, cache, stream)

    models = rs.Models(text, wr, pipe, encoder, standins.standin_decoder)
    params = rs.GenerateParams(prompt="synthetic", seed=9, num_blocks=5, num_denoising_steps=4, kv_cache_num_frames=3,
                               keep_first_frame=True, input_video="memory://video", strength=0.6)
    sess = rs.GenerationSession(params, types.SimpleNamespace(use_taehv=False), frame_callback=lambda *a, **k: None, models=models)
    out = {"prompt": prompt, "noise": sess.noise.clone(), "num_blocks": sess.num_blocks, "steps": sess.denoising_step_list.clone(),
           "blocks": [], "indices": []}
    for b in range(sess.num_blocks):
        sess.generate_block_internal(models)
        out["blocks"].append(sess.last_pred.clone())
        out["indices"].append((int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"]),
                               sess.current_start_frame, sess.block_idx, sess.total_frames_sent))
    assert sess.generate_block_internal(models) is None
    out["encoder_calls"] = enc_calls
    torch.save(out, os.path.join(OUT, "session_v2v_reference.pt"))
    print("session_v2v_reference.pt", out["num_blocks"], out["indices"], tuple(out["noise"].shape), enc_calls, out["steps"])


def webcam_frames(seed=31, counts=(11, 12, 14)):
    """Input frames of the webcam golden: per block a list of [3, 480, 832] fp16 frames in [-1, 1] (smooth in space so that
    the stand-in encoder's pooling is well conditioned)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for n in counts:
        low = torch.rand(n, 3, 15, 26, generator=g) * 2 - 1
        out.append(list(torch.nn.functional.interpolate(low, size=(480, 832), mode="bilinear").half()))
    return out


def golden_session_webcam():
    """The reference's GenerationSession in webcam (streaming v2v) mode with a prompt transition (release_server.py:489-527,
    :402-413, :651-666): 3 blocks, strength 0.8, frames put on the session's frame_queue as tensors (the JPEG decoding of
    push_frame is control plane), 11 / 12 / 14 frames waiting at blocks 0 / 1 / 2 (resampled to 9 / 12 / 12), prompt
    interpolation over 2 steps requested after block 0.  torch.randn_like of the input-noising uses the global generator:
    it is seeded with 100 + block before every block."""
    import types
    from oracle import standins
    rs, CIP = ref_shim.load_release_server()
    ref = ref_shim.load()
    orig_to = torch.Tensor.to

    def to_cpu(self, *a, **k):       # generate_block_internal moves the encoded input with .to("cuda") (:656)
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        return orig_to(self, *a, **k)
    torch.Tensor.to = to_cpu
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    model = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
    model.config = types.SimpleNamespace(num_heads=cfg["num_heads"], dim=cfg["dim"])
    wr = ref_shim.build_reference_wrapper(ref, model)
    g = torch.Generator().manual_seed(5)
    prompts = []
    for _ in range(2):
        p = torch.zeros(1, 512, TEXT_DIM, dtype=torch.bfloat16)
        p[0, :64] = torch.randn(64, TEXT_DIM, generator=g).to(torch.bfloat16)
        prompts.append(p)
    text = standins.StandinTextEncoder(prompts[0], {"second prompt": prompts[1]})
    args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=False, num_frame_per_block=3,
                                 independent_first_frame=False)
    pipe = CIP(args, "cpu", generator=wr, text_encoder=text, vae=object())
    enc_calls = []

    def encoder(frames, cache, stream=False):
        enc_calls.append((tuple(frames.shape), bool(stream), frames.float()[..., ::40, ::52].clone()))
        return standins.standin_encoder(frames, cache, stream)

    models = rs.Models(text, wr, pipe, encoder, standins.standin_decoder)
    params = rs.GenerateParams(prompt="first prompt", seed=9, num_blocks=3, num_denoising_steps=4, kv_cache_num_frames=3,
                               keep_first_frame=False, webcam_mode=True, strength=0.8)
    sess = rs.GenerationSession(params, types.SimpleNamespace(use_taehv=False), frame_callback=lambda *a, **k: None, models=models)
    out = {"steps": sess.denoising_step_list.clone(), "prompts": prompts, "blocks": [], "indices": [], "prompt_used": []}
    for b, frames in enumerate(webcam_frames()):
        for f in frames:
            sess.frame_queue.put(f)
        torch.manual_seed(100 + b)
        sess.generate_block_internal(models)
        out["blocks"].append(sess.last_pred.clone())
        out["prompt_used"].append(sess.current_prompt_embeds.clone())
        out["indices"].append((int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"]),
                               sess.current_start_frame, sess.block_idx, sess.total_frames_sent))
        if b == 0:
            sess.interpolate_prompt_embeds(models, "second prompt", 2)
    out["all_latents"] = sess.all_latents.clone()
    out["encoder_calls"] = enc_calls
    torch.Tensor.to = orig_to
    torch.save(out, os.path.join(OUT, "session_webcam_reference.pt"))
    print("session_webcam_reference.pt", out["indices"], [(c[0], c[1]) for c in enc_calls], out["steps"])


def golden_pipeline_inference():
    """The reference's own CausalInferencePipeline.inference (pipeline/causal_inference.py:48-277, the original Self-Forcing
    driver: per block 4 denoise forwards, then one forward at context_noise that writes the clean K/V): tiny DiT, warped
    denoising steps, 3 input frames (video extension, Step 2) + 2 generated blocks, cache of 32760 rows, stand-in VAE / text
    encoder.  The re-noising uses torch.randn_like on the global generator (seed 77); the strides of its argument are
    recorded so that a test can repeat the very same draws."""
    import types
    from oracle import standins
    rs, CIP = ref_shim.load_release_server()
    ref = ref_shim.load()
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    model = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
    model.config = types.SimpleNamespace(num_heads=cfg["num_heads"], dim=cfg["dim"])
    wr = ref_shim.build_reference_wrapper(ref, model)
    g = torch.Generator().manual_seed(5)
    prompt = torch.zeros(1, 512, TEXT_DIM, dtype=torch.bfloat16)
    prompt[0, :64] = torch.randn(64, TEXT_DIM, generator=g).to(torch.bfloat16)
    noise = torch.randn(1, 6, 16, 60, 104, generator=g).to(torch.bfloat16)
    initial = torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16)
    def run(independent, noise, initial, seed):
        args = types.SimpleNamespace(denoising_step_list=[1000, 750, 500, 250], warp_denoising_step=True, num_frame_per_block=3,
                                     independent_first_frame=independent, context_noise=0)
        pipe = CIP(args, "cpu", generator=wr, text_encoder=standins.StandinTextEncoder(prompt), vae=standins.StandinVAE())
        draws = []
        add_noise = pipe.scheduler.add_noise

        def spy(x, eps, t):
            draws.append((tuple(eps.shape), tuple(eps.stride()), float(eps.float().sum())))
            return add_noise(x, eps, t)
        pipe.scheduler.add_noise = spy
        torch.manual_seed(seed)
        with torch.inference_mode():
            video, latents = pipe.inference(noise, ["a prompt"], initial_latent=initial, return_latents=True)
        pipe.scheduler.add_noise = add_noise
        return {"noise": noise, "initial": initial, "steps": pipe.denoising_step_list.clone(), "latents": latents.clone(),
                "video_shape": tuple(video.shape), "video_sample": video[0, :, :, ::40, ::52].clone(), "draws": draws,
                "indices": (int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"])),
                "kv_shape": tuple(pipe.kv_cache1[0]["k"].shape), "cache": cache_sample(pipe.kv_cache1), "seed": seed,
                "independent_first_frame": independent}

    out = {"prompt": prompt}
    out.update(run(False, noise, initial, 77))                                  # video extension: 3 input frames + 2 blocks
    # independent first frame ([1, 3, ...] block structure, causal_inference.py:80-83, :139-154, :189-191):
    out["t2v_independent"] = run(True, noise[:, :4].contiguous(), None, 78)     # 1 + 3 generated frames
    out["i2v_independent"] = run(True, noise[:, :3].contiguous(), initial[:, :1].contiguous(), 79)   # 1 input frame + 1 block
    torch.save(out, os.path.join(OUT, "pipeline_inference_reference.pt"))
    print("pipeline_inference_reference.pt", out["video_shape"], out["indices"], out["t2v_independent"]["indices"],
          out["i2v_independent"]["indices"], out["t2v_independent"]["video_shape"])


def golden_t5():
    """Text encoder: the reference's own T5Encoder (wan/modules/t5.py:267-313, shared_pos=False like umt5_xxl,
    float32 like WanTextEncoder) at tiny dims with head_dim 64, two prompts of 29 and 48 tokens in a 48-slot window, plus
    the zeroing of the padding rows (wan_wrapper.py:52-53)."""
    from oracle import t5_oracle as to
    t5 = ref_shim.load_t5()
    cfg = dict(to.TINY_T5)
    w = to.make_t5_weights(cfg, seed=0)
    model = t5.T5Encoder(vocab=cfg["vocab"], dim=cfg["dim"], dim_attn=cfg["dim_attn"], dim_ffn=cfg["dim_ffn"],
                         num_heads=cfg["num_heads"], num_layers=cfg["num_layers"], num_buckets=cfg["num_buckets"],
                         shared_pos=False, dropout=0.1).eval()
    assert set(model.state_dict()) == set(w), sorted(set(model.state_dict()) ^ set(w))[:8]
    model.load_state_dict(w)
    ids, mask = to.t5_inputs(cfg)
    with torch.inference_mode():
        ctx = model(ids, mask).clone()
        raw = ctx.clone()
        for u, v in zip(ctx, mask.gt(0).sum(dim=1)):
            u[v:] = 0.0
        bias = model.blocks[1].pos_embedding(48, 48).clone()
    out = {"ids": ids, "mask": mask, "context_raw": raw, "prompt_embeds": ctx, "pos_bias_block1": bias,
           "weights_checksum": float(sum(v.double().abs().sum() for v in w.values()))}
    torch.save(out, os.path.join(OUT, "t5_encoder.pt"))
    print("t5_encoder.pt", tuple(ctx.shape), float(ctx.abs().mean()))


def golden_full_width():
    """BASELINE config 3's width on the REFERENCE ITSELF: the upstream CausalWanModel with dim 5120 / 40 heads / ffn 13824, ONE
    layer, bf16 on the host cores (SDPA fallback), M = 4680 query tokens at cache offset 4680 over a 9360-row window whose
    first half holds earlier K/V (seeded) - flow, x0 and sampled cache rows.  Inputs and weights are regenerated from the
    seeds by the test (tests/test_dit_gpu.py::test_full_width_14b_layer_matches_reference_golden); the file also records how
    long the reference forward and the oracle port took on this container's cores (the bench's cpu_baseline is the port:
    the Python reference cannot travel to the GPU box)."""
    import time
    ref = ref_shim.load()
    cfg = dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=1, freq_dim=256, text_len=512, eps=1e-6,
               num_frame_per_block=3)
    w = wo.make_weights(cfg, seed=5, text_dim=256)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16)
    ctx = torch.randn(40, 256, generator=g).to(torch.bfloat16)
    t = torch.tensor([[713.0, 713.0, 713.0]])
    old_k = torch.randn(1, 4680, 40, 128, generator=g).to(torch.bfloat16)
    old_v = torch.randn(1, 4680, 40, 128, generator=g).to(torch.bfloat16)

    def prefilled():
        kv, ca = fresh_caches(cfg, 9360)
        kv[0]["k"][:, :4680], kv[0]["v"][:, :4680] = old_k, old_v
        kv[0]["global_end_index"] = kv[0]["local_end_index"] = 4680
        return kv, ca

    model = ref_shim.build_reference_model(ref, cfg, w, 256)
    wr = ref_shim.build_reference_wrapper(ref, model)
    out = {"cores": len(os.sched_getaffinity(0)), "threads": torch.get_num_threads()}
    with torch.inference_mode():
        for rep in range(2):                      # second (warm) run timed, like bench.py's cpu_baseline
            kv, ca = prefilled()
            t0 = time.perf_counter()
            flow, x0 = wr(lat, {"prompt_embeds": [ctx]}, t, kv, ca, current_start=4680)
            out["reference_forward_s"] = time.perf_counter() - t0
        for rep in range(2):
            kvp, cap = prefilled()
            t0 = time.perf_counter()
            pflow, _ = wo.wrapper_forward(w, cfg, wo.FlowMatchScheduler(), lat, [ctx], t, kvp, cap, 4680)
            out["port_forward_s"] = time.perf_counter() - t0
    out.update(flow=flow.clone(), x0=x0.clone(), k_new=kv[0]["k"][0, 4680::97].clone(), v_new=kv[0]["v"][0, 4680::97].clone(),
               k_old_checksum=float(kv[0]["k"][0, :4680].double().abs().sum()),
               local_end_index=int(kv[0]["local_end_index"]), global_end_index=int(kv[0]["global_end_index"]),
               port_vs_reference_rel_l2=float((pflow.double() - flow.double()).norm() / flow.double().norm()))
    torch.save(out, os.path.join(OUT, "dit_full_width_layer.pt"))
    print("dit_full_width_layer.pt done: reference %.1f s, port %.1f s on %d threads, port vs reference rel-L2 %.2e" % (
        out["reference_forward_s"], out["port_forward_s"], out["threads"], out["port_vs_reference_rel_l2"]))


def golden_loader_manifest(ref):
    """tests/golden/checkpoint_manifest.json: the state-dict key / shape / dtype set a checkpoint for the reference's
    `WanDiffusionWrapper(model_name=..., is_causal=True)` carries (release_server.py:160-167: `load_file(...)` -> keys prefixed
    `model.` because the wrapper holds the CausalWanModel as `self.model`, utils/wan_wrapper.py:121-151), for both architectures the
    server can load (:162-165), taken from the reference's OWN module tree instantiated on the meta device (no memory, no weights).
    `from_pretrained` needs diffusers + a config.json, neither is here: the constructor is called with the dims of
    wan/configs/wan_t2v_14B.py:21-25 / wan_t2v_1_3B.py:21-25.  Data only: names and shapes."""
    import json
    archs = {"Wan2.1-T2V-14B": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40),
             "Wan2.1-T2V-1.3B": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30)}
    out = {}
    for name, a in archs.items():
        with torch.device("meta"):
            m = ref.cm.CausalWanModel(model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, freq_dim=256, text_dim=4096,
                                      out_dim=16, qk_norm=True, cross_attn_norm=True, eps=1e-6, **a)
        wrapper = torch.nn.Module()
        wrapper.model = m                                   # the wrapper's only nn.Module child (utils/wan_wrapper.py:137-139)
        sd = wrapper.state_dict()
        out[name] = {"arch": a, "unfused": {k: list(v.shape) for k, v in sd.items()}}
        for blk in m.blocks:                                # release_server.py:176-177
            blk.self_attn.fuse_projections()
        out[name]["fused"] = {k: list(v.shape) for k, v in wrapper.state_dict().items()}
        print(name, len(out[name]["unfused"]), "keys unfused,", len(out[name]["fused"]), "fused,",
              sum(int(torch.Size(v).numel()) for v in out[name]["unfused"].values()) / 1e9, "G parameters")
        # stored compactly: the 40 / 30 blocks carry the same names and shapes (asserted here), so one block's entries under the
        # placeholder index "{i}" + the layer count reproduce the full table (tests/test_checkpoint_loader.py: manifest())
        for form in ("unfused", "fused"):
            full = out[name][form]
            pre = "model.blocks."
            per = {k[len(pre):].split(".", 1)[1]: v for k, v in full.items() if k.startswith(pre + "0.")}
            for i in range(a["num_layers"]):
                assert {k[len(pre):].split(".", 1)[1]: v for k, v in full.items() if k.startswith(f"{pre}{i}.")} == per, (name, form, i)
            top = {k: v for k, v in full.items() if not k.startswith(pre)}
            assert len(top) + a["num_layers"] * len(per) == len(full)
            out[name][form] = {"top": top, "per_block": {pre + "{i}." + k: v for k, v in per.items()}, "num_keys": len(full)}
    with open(os.path.join(OUT, "checkpoint_manifest.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    ref = ref_shim.load()
    which = sys.argv[1:] or ["ops", "dit", "rolling", "vae", "vae_single", "vae_enc", "vae_wrapper", "t5", "session", "webcam", "start_frame", "v2v", "pipeline"]
    if "loader_manifest" in which:   # not in the default list (names / shapes only; does not change unless upstream's modules do)
        golden_loader_manifest(ref)
    if "ops" in which:
        golden_ops(ref)
    if "dit" in which:
        golden_dit(ref)
    if "rolling" in which:
        golden_rolling(ref)
    if "vae" in which:
        golden_vae(ref)
    if "vae_single" in which:
        golden_vae_single()
    if "vae_enc" in which:
        golden_vae_encoder(ref)
    if "vae_wrapper" in which:
        golden_wan_vae_wrapper(ref)
    if "t5" in which:
        golden_t5()
    if "full_width" in which:   # not in the default list: minutes of host time (one 14B-width layer on the reference)
        golden_full_width()
    if "session" in which:      # from here on load_release_server() has patched torch.cuda for the rest of the process
        golden_session()
    if "webcam" in which:
        golden_session_webcam()
    if "start_frame" in which:
        golden_session_start_frame()
    if "v2v" in which:
        golden_session_v2v()
    if "pipeline" in which:
        golden_pipeline_inference()
