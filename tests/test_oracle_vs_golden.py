"""Pins the CPU oracle (oracle/wan_oracle.py) against golden vectors minted from the upstream
reference's own modules (oracle/make_golden.py).  CPU only."""
import os
import sys

import torch

from conftest import max_abs, rel_l2

from oracle import wan_oracle as wo

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle.make_golden import TEXT_DIM, TINY, cache_sample, fresh_caches, tiny_inputs  # noqa: E402

BF16_EPS = 2 ** -8


def test_attention_sdpa_matches_reference(golden):
    g = golden("ops.pt")
    out = wo.attention_sdpa(g["attn_q"], g["attn_k"], g["attn_v"])
    assert out.dtype == torch.bfloat16 and out.is_contiguous()
    assert max_abs(out, g["attn_out"]) <= 2e-2
    gold = wo.attention_math(g["attn_q"], g["attn_k"], g["attn_v"])
    assert max_abs(g["attn_out"], gold) <= 2e-2  # the reference itself vs the fp32 definition


def test_rope_matches_reference(golden):
    g = golden("ops.pt")
    freqs = wo.rope_table(128)
    assert torch.equal(torch.view_as_real(freqs[[0, 1, 5, 100, 1023]]), g["freqs_sample"])
    assert torch.equal(wo.rope_apply(g["rope_x"], (2, 6, 8), freqs, start_frame=3), g["rope_causal_s3"])
    assert torch.equal(wo.rope_apply(g["rope_x"], (2, 6, 8), freqs), g["rope_s0"])


def test_norms_match_reference(golden):
    g = golden("ops.pt")
    assert torch.equal(wo.rms_norm(g["norm_x"], g["norm_w"]), g["rms"])
    assert max_abs(wo.layer_norm(g["norm_x"]), g["ln"]) <= 2 * BF16_EPS * 4


def test_scheduler_matches_reference(golden):
    g = golden("ops.pt")
    s = wo.FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    assert torch.equal(s.timesteps, g["sched_timesteps"])
    assert torch.equal(s.sigmas, g["sched_sigmas"])
    zp = torch.cat((s.timesteps, torch.tensor([0], dtype=torch.float32)))
    assert torch.equal(wo.get_denoising_schedule(zp, 1.0, 4), g["schedule_4"])
    assert torch.equal(wo.get_denoising_schedule(zp, 1.0, 5), g["schedule_5"])
    assert torch.equal(wo.get_denoising_schedule(zp, 0.7, 4), g["schedule_4_s07"])
    # headline schedule: 4 steps, shift 5 -> [1000, 908.8, 714.0, 0]
    assert [round(float(v), 1) for v in g["schedule_4"]] == [1000.0, 908.8, 714.0, 0.0]
    assert torch.equal(s.add_noise(g["an_x0"], g["an_noise"], g["an_t"]), g["an_out"])
    assert torch.equal(wo.convert_flow_pred_to_x0(s, g["an_x0"], g["an_noise"], g["an_t"].float()), g["x0_out"])
    assert torch.equal(wo.sinusoidal_embedding_1d(256, g["schedule_4"]), g["sinus"])


def _check_cache(sample, gold, tol):
    for c, gc in zip(sample, gold):
        assert c["global_end_index"] == gc["global_end_index"]
        assert c["local_end_index"] == gc["local_end_index"]
        # rows that must be zero (never written) are exactly zero in both
        assert torch.equal(c["k"] == 0, gc["k"] == 0)
        assert rel_l2(c["k"], gc["k"]) <= tol and rel_l2(c["v"], gc["v"]) <= tol


def test_dit_server_path_matches_reference(golden):
    """recompute + denoise sequence of SURVEY.md Appendix B on the tiny model, bf16 on CPU."""
    g = golden("dit_server_path.pt")
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    assert abs(float(sum(v.double().abs().sum() for v in w.values())) - g["weights_checksum"]) < 1e-6 * g["weights_checksum"]
    lat, ctx = tiny_inputs()
    kv, ca = fresh_caches(cfg, 9360)
    sched = wo.FlowMatchScheduler()
    steps = g["steps"]
    tol = 1e-2

    def ts(v):
        return torch.ones([1, 3], dtype=torch.int64) * v

    flow, x0 = wo.wrapper_forward(w, cfg, sched, lat[0], [ctx], ts(steps[0]), kv, ca, 0)
    assert rel_l2(flow, g["b0s0_flow"]) <= tol and rel_l2(x0, g["b0s0_x0"]) <= tol
    _check_cache(cache_sample(kv), g["b0s0_cache"], tol)
    flow, _ = wo.wrapper_forward(w, cfg, sched, lat[1], [ctx], ts(steps[1]), kv, ca, 0)
    assert rel_l2(flow, g["b0s1_flow"]) <= tol
    _check_cache(cache_sample(kv), g["b0s1_cache"], tol)
    wo.reset_kv_cache(kv)
    flow, _ = wo.wrapper_forward(w, cfg, sched, lat[2], [ctx], torch.zeros([1, 3], dtype=torch.int64), kv, ca,
                                 3 * 1560, recompute=True)
    assert rel_l2(flow, g["rc_flow"]) <= tol
    _check_cache(cache_sample(kv), g["rc_cache"], tol)
    flow, x0 = wo.wrapper_forward(w, cfg, sched, lat[3], [ctx], ts(steps[0]), kv, ca, 4680)
    assert rel_l2(flow, g["b1s0_flow"]) <= tol and rel_l2(x0, g["b1s0_x0"]) <= tol
    _check_cache(cache_sample(kv), g["b1s0_cache"], tol)
    assert kv[0]["local_end_index"] == 9360 and kv[0]["global_end_index"] == 9360


def test_dit_rolling_cache_matches_reference(golden):
    g = golden("dit_rolling.pt")
    cfg = dict(TINY, local_attn_size=6, sink_size=1, num_layers=1)
    w = wo.make_weights(cfg, seed=3, text_dim=TEXT_DIM)
    lat, ctx = tiny_inputs(seed=7)
    kv, ca = fresh_caches(cfg, 6 * 1560)
    sched = wo.FlowMatchScheduler()
    idx = []
    for b in range(4):
        t = torch.ones([1, 3], dtype=torch.int64) * 500
        flow, _ = wo.wrapper_forward(w, cfg, sched, lat[b], [ctx], t, kv, ca, b * 4680)
        idx.append((kv[0]["global_end_index"], kv[0]["local_end_index"]))
        assert rel_l2(flow[0, :, :, ::3, ::4], g["flow_sample"][b]) <= 1e-2
    assert idx == g["indices"] == [(4680, 4680), (9360, 9360), (14040, 9360), (18720, 9360)]
    _check_cache(cache_sample(kv), g["cache"], 1e-2)


def test_gold_fp32_graph_bounds_bf16_error(golden):
    """The fp32 restatement of the same graph is the 'gold' used to state tolerances."""
    g = golden("dit_server_path.pt")
    cfg = dict(TINY)
    w = {k: v.float() for k, v in wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM).items()}
    lat, ctx = tiny_inputs()
    kv, ca = fresh_caches(cfg, 9360, dtype=torch.float32)
    sched = wo.FlowMatchScheduler()
    t = torch.ones([1, 3], dtype=torch.int64) * g["steps"][0]
    flow, _ = wo.wrapper_forward(w, cfg, sched, lat[0].float(), [ctx.float()], t, kv, ca, 0,
                                 attn_fn=lambda q, k, v: wo.attention_sdpa(q, k, v, dtype=None))
    err = rel_l2(g["b0s0_flow"], flow)
    assert err <= 2e-2, err  # reference bf16 eager vs fp32 gold


def test_config1_plumbing_320x192_one_block_one_step():
    """BASELINE config 1 (plumbing, CPU eager): 1.3B widths, 320x192 -> latents [1,3,16,24,40] (240 tokens per frame),
    one block, one denoising step through the oracle's block loop.  frame_seqlen is hard-coded to 1560 in the reference
    (SURVEY trap 3), so only block 0 is meaningful at this size: shapes and finiteness are asserted, as the survey
    prescribes; one layer keeps the CPU run short."""
    from oracle import wan_oracle as wo
    cfg = dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=1, freq_dim=256, text_len=512, eps=1e-6)
    w = wo.make_weights(cfg, seed=0, text_dim=64)
    g = torch.Generator().manual_seed(2)
    noise = torch.randn(1, 3, 16, 24, 40, generator=g).to(torch.bfloat16)
    ctx = torch.randn(16, 64, generator=g).to(torch.bfloat16)
    ora = wo.SessionOracle(w, cfg, [ctx], noise, kv_cache_num_frames=3, num_steps=1, shift=5.0, seed=0)
    out = ora.generate_block()
    assert out.shape == (1, 3, 16, 24, 40) and out.dtype == torch.bfloat16 and torch.isfinite(out.float()).all()
    assert ora.kv_cache[0]["global_end_index"] == 3 * 240 and ora.current_start_frame == 3


def test_fp8_linear_restatement_properties():
    """The fp8 nn.Linear restatement (torchao PerTensor dynamic activation x static weight, oracle/wan_oracle.fp8_linear):
    exact on values that e4m3 represents, saturating scale at the tensor maximum, small error on Gaussian data."""
    from oracle import wan_oracle as wo
    g = torch.Generator().manual_seed(0)
    x = torch.randn(64, 256, generator=g).to(torch.bfloat16)
    w = (torch.randn(128, 256, generator=g) * 256 ** -0.5).to(torch.bfloat16)
    b = torch.randn(128, generator=g).to(torch.bfloat16)
    y8 = wo.fp8_linear(x, w, b)
    y = torch.nn.functional.linear(x.float(), w.float(), b.float())
    assert y8.dtype == torch.bfloat16 and rel_l2(y8, y) <= 6e-2
    # powers of two up to the scale maximum survive e4m3 exactly -> the fp8 product equals the exact one
    xe = torch.tensor([[448.0, 1.0, -2.0, 0.5] + [0.0] * 124]).to(torch.bfloat16)
    we = torch.eye(128).to(torch.bfloat16)
    assert torch.equal(wo.fp8_linear(xe, we, None), xe)
    # the flag routes every Linear of the model through it, q/k/v as ONE fused tensor
    cfg = dict(TINY, num_layers=1)
    wts = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    h = torch.randn(1, 8, cfg["dim"], generator=g).to(torch.bfloat16)
    q0, k0, v0 = wo._qkv(h, wts, "blocks.0.self_attn")
    w8 = dict(wts)
    w8[wo.FP8_FLAG] = True
    q8, k8, v8 = wo._qkv(h, w8, "blocks.0.self_attn")
    assert 0 < rel_l2(q8, q0) <= 8e-2 and 0 < rel_l2(v8, v0) <= 8e-2


def test_t5_encoder_oracle_matches_reference_golden(golden):
    """oracle/t5_oracle (text encoder) vs the golden minted from the reference's T5Encoder + the padding-row
    zeroing of WanTextEncoder: float32 on both sides, same op order -> tight tolerance."""
    from oracle import t5_oracle as to
    gold = golden("t5_encoder.pt")
    cfg = dict(to.TINY_T5)
    w = to.make_t5_weights(cfg, seed=0)
    assert abs(float(sum(v.double().abs().sum() for v in w.values())) - gold["weights_checksum"]) < 1e-6
    ids, mask = to.t5_inputs(cfg)
    assert torch.equal(ids, gold["ids"]) and torch.equal(mask, gold["mask"])
    bias = to.relative_bias(w["blocks.1.pos_embedding.embedding.weight"], 48, 48)
    assert torch.equal(bias, gold["pos_bias_block1"])
    assert torch.allclose(to.encoder(w, ids, mask, cfg), gold["context_raw"], atol=2e-5, rtol=1e-5)
    out = to.text_encoder_forward(w, ids, mask, cfg)["prompt_embeds"]
    assert torch.allclose(out, gold["prompt_embeds"], atol=2e-5, rtol=1e-5)
    assert float(out[0, 29:].abs().max()) == 0.0 and float(out[0, 28].abs().max()) > 0


def test_t5_relative_position_buckets_known_values():
    """Bidirectional 32-bucket rule (t5.py:238-265): 16 buckets per direction, exact below 8, log-spaced up to 128."""
    from oracle import t5_oracle as to
    rel = torch.tensor([0, 1, 7, 8, 11, 12, 15, 16, 64, 127, 128, 500, -1, -7, -8, -127, -128, -511])
    assert to.relative_position_bucket(rel).tolist() == [0, 17, 23, 24, 24, 25, 25, 26, 30, 31, 31, 31, 1, 7, 8, 15, 15, 15]


def _session_harness_first_frame(frames_cache):
    """release_server.py:572-575 + v2v.py:138-158 with the stand-in encoder: the oldest cached pixel frame, as fp16, through
    the (size-preserving) bicubic resize, into the encoder; returns [1, 1, 16, h, w]."""
    from oracle import standins
    frames = frames_cache[0][0].half()                                               # [1, 3, H, W]
    frames = torch.nn.functional.interpolate(frames, size=(480, 832), mode="bicubic").transpose(0, 1).to(torch.float16)
    mu, _ = standins.standin_encoder(frames.unsqueeze(0), [None] * 55, stream=False)
    return mu.squeeze(0).to(torch.float16).transpose(0, 1)[None], frames


def test_session_oracle_matches_reference_generation_session(golden):
    """SessionOracle (restatement of GenerationSession.recompute_kv_cache / generate_block_internal) against a golden minted
    by running the reference's OWN GenerationSession class for 3 blocks on the CPU (oracle/make_golden.py `session`): same
    noise stream (one CPU generator: the latent noise first, then the re-noising draws), same schedule, same context-frame
    selection incl. the first-frame re-encode branch at block 2, same cache bookkeeping."""
    from collections import deque
    from oracle import standins
    gold = golden("session_reference.pt")
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    rnd = torch.Generator().manual_seed(9)
    noise = torch.randn([1, 9, 16, 60, 104], dtype=torch.bfloat16, generator=rnd)
    assert torch.equal(noise, gold["noise"])
    frames_cache, enc_inputs = deque(maxlen=9), []

    def first_frame(block_idx):
        lat, frames = _session_harness_first_frame(frames_cache)
        enc_inputs.append(frames)
        return lat

    ora = wo.SessionOracle(w, cfg, [gold["prompt"][0]], noise, kv_cache_num_frames=3, num_steps=4, shift=5.0, seed=0,
                           first_frame_fn=first_frame)
    ora.rnd = rnd                                                                    # continues after the noise draw
    assert torch.equal(ora.denoising_step_list, gold["steps"])
    vae_cache = [None] * 55
    for b in range(3):
        out = ora.generate_block()
        assert rel_l2(out, gold["blocks"][b]) <= 1e-2, b
        px, vae_cache = standins.standin_decoder(out.half(), *vae_cache)
        frames_cache.extend(px.split(1, dim=1))
        assert (int(ora.kv_cache[0]["global_end_index"]), int(ora.kv_cache[0]["local_end_index"]),
                ora.current_start_frame, ora.block_idx) == gold["indices"][b][:4]
    assert rel_l2(ora.all_latents, gold["all_latents"]) <= 1e-2
    assert len(enc_inputs) == len(gold["encoder_inputs"]) == 1                        # only block 2 re-encodes
    assert tuple(enc_inputs[0].unsqueeze(0).shape) == gold["encoder_input_shapes"][0]
    assert max_abs(enc_inputs[0].unsqueeze(0).float()[..., ::40, ::52], gold["encoder_inputs"][0]) <= 2e-2


def _run_oracle_blocks(ora, gold, n, decode=True, tol=1e-2):
    from collections import deque
    from oracle import standins
    frames_cache, vae_cache = deque(maxlen=9), [None] * 55
    ora.first_frame_fn = lambda b: _session_harness_first_frame(frames_cache)[0]
    for b in range(n):
        yield b
        out = ora.generate_block()
        assert out is not None and rel_l2(out, gold["blocks"][b]) <= tol, b
        px, vae_cache = standins.standin_decoder(out.half(), *vae_cache)
        frames_cache.extend(px.split(1, dim=1))
        assert (int(ora.kv_cache[0]["global_end_index"]), int(ora.kv_cache[0]["local_end_index"]),
                ora.current_start_frame, ora.block_idx) == gold["indices"][b][:4]


def test_session_oracle_webcam_mode_matches_reference_session(golden):
    """Streaming v2v + prompt transition of SessionOracle vs the golden of the reference's GenerationSession in webcam mode."""
    from oracle import standins
    from oracle.make_golden import webcam_frames
    gold = golden("session_webcam_reference.pt")
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    rnd = torch.Generator().manual_seed(9)
    noise = torch.randn([1, 9, 16, 60, 104], dtype=torch.bfloat16, generator=rnd)
    ora = wo.SessionOracle(w, cfg, [gold["prompts"][0][0]], noise, kv_cache_num_frames=3, num_steps=4, strength=0.8,
                           webcam_encoder=standins.standin_encoder)
    ora.rnd = rnd
    assert torch.allclose(ora.denoising_step_list, gold["steps"])
    frames = webcam_frames()
    for b in _run_oracle_blocks(ora, gold, 3):
        ora.frame_queue.extend(frames[b])
        torch.manual_seed(100 + b)                       # the golden script seeds the global generator per block
        if b == 1:
            ora.interpolate_prompt_embeds(gold["prompts"][1][0], 2)
    assert rel_l2(ora.all_latents, gold["all_latents"]) <= 1e-2
    assert torch.equal(ora.prompt_embeds[0][None], gold["prompt_used"][2])


def test_session_oracle_start_frame_matches_reference_session(golden):
    import numpy as np
    from oracle import standins
    from oracle.make_golden import start_image
    gold = golden("session_start_frame_reference.pt")
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    rnd = torch.Generator().manual_seed(9)
    noise = torch.randn([1, 9, 16, 60, 104], dtype=torch.bfloat16, generator=rnd)
    ora = wo.SessionOracle(w, cfg, [gold["prompt"][0]], noise, kv_cache_num_frames=3, num_steps=4)
    ora.rnd = rnd
    img = torch.from_numpy(np.asarray(start_image()).copy()).permute(2, 0, 1).float().div(255)
    ora.setup_start_frame(img, standins.standin_encoder)
    assert torch.allclose(ora.resume_latents.float(), gold["resume_latents"].float(), atol=2e-3)
    for _ in _run_oracle_blocks(ora, gold, 2):
        pass
    assert rel_l2(ora.all_latents, gold["all_latents"]) <= 1e-2


def test_session_oracle_offline_v2v_matches_reference_session(golden):
    from oracle import standins
    from oracle.make_golden import v2v_video
    gold = golden("session_v2v_reference.pt")
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    rnd = torch.Generator().manual_seed(9)
    noise = torch.randn([1, 15, 16, 60, 104], dtype=torch.bfloat16, generator=rnd)
    ora = wo.SessionOracle(w, cfg, [gold["prompt"][0]], noise, kv_cache_num_frames=3, num_steps=4, strength=0.6, num_blocks=5)
    ora.rnd = rnd
    ora.setup_input_video(v2v_video(), standins.standin_encoder)
    assert ora.num_blocks == gold["num_blocks"] and torch.equal(ora.noise, gold["noise"])
    for _ in _run_oracle_blocks(ora, gold, 2):
        pass
    assert ora.generate_block() is None


def test_pipeline_inference_oracle_matches_reference_pipeline(golden):
    """wo.pipeline_inference (restatement of CausalInferencePipeline.inference) vs the goldens of the reference's own class:
    video extension, and the independent-first-frame [1, 3] structure with / without an image latent."""
    top = golden("pipeline_inference_reference.pt")
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    for variant in ("extension", "t2v_independent", "i2v_independent"):
        gold = top if variant == "extension" else top[variant]
        torch.manual_seed(gold.get("seed", 77))          # re-noising = torch.randn_like on the global generator
        lat, kv = wo.pipeline_inference(w, cfg, [top["prompt"][0]], gold["noise"], gold["initial"], warp_denoising_step=True,
                                        independent_first_frame=variant != "extension")
        assert rel_l2(lat, gold["latents"]) <= 1e-2, variant
        assert (int(kv[0]["global_end_index"]), int(kv[0]["local_end_index"])) == gold["indices"], variant
        for c, g in zip(cache_sample(kv), gold["cache"]):
            assert rel_l2(c["k"], g["k"]) <= 1e-2 and rel_l2(c["v"], g["v"]) <= 1e-2
