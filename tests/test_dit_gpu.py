"""Model-level parity on the MI355X: the native DiT forward (rtv_dit_forward through the
CausalWanModel / WanDiffusionWrapper / CausalInferencePipeline / GenerationSession mirrors) against
golden vectors minted from the upstream reference and against the oracle graph (evaluated on the device, see _on_dev).

Stated tolerance: gold = the same graph in fp32 on CPU; accept
max_abs_err(ours, gold) <= 2 x max_abs_err(reference_bf16, gold) (+ small absolute floor) and
rel-L2(ours, reference_bf16) <= 2e-2; KV-cache index bookkeeping must match exactly."""
import pytest
import torch

from conftest import max_abs, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _on_dev(w):
    """Oracle weights on the GPU.  The oracle GRAPH (oracle/wan_oracle.py; its host evaluation is pinned to the reference's
    goldens by tests/test_oracle_vs_golden.py) is evaluated by torch eager on the device in this file: the host evaluation of
    the full-size cases costs minutes per test on a box with few cores (GPUTEST_r02 timed out on exactly that).  Tables
    (sinusoid, RoPE, scheduler) are still built on the host inside the oracle."""
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in w.items()}


def _tiny():
    from oracle.make_golden import TEXT_DIM, TINY, tiny_inputs
    return dict(TINY), TEXT_DIM, tiny_inputs


def _build(cfg, text_dim, weights, **kw):
    from realtime_video_amd.causal_model import CausalWanModel
    from realtime_video_amd.wan_wrapper import WanDiffusionWrapper
    m = CausalWanModel(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                       num_layers=cfg["num_layers"], text_dim=text_dim, freq_dim=cfg.get("freq_dim", 256),
                       local_attn_size=cfg.get("local_attn_size", -1), sink_size=cfg.get("sink_size", 0), **kw)
    m.load_state_dict(weights)
    return m, WanDiffusionWrapper(m, timestep_shift=5.0)


def _caches(cfg, kv_size):
    hd = cfg["dim"] // cfg["num_heads"]
    L, H = cfg["num_layers"], cfg["num_heads"]
    kv = [{"k": torch.zeros(1, kv_size, H, hd, dtype=torch.bfloat16, device=DEV),
           "v": torch.zeros(1, kv_size, H, hd, dtype=torch.bfloat16, device=DEV),
           "global_end_index": 0, "local_end_index": 0} for _ in range(L)]
    ca = [{"k": torch.zeros(1, 512, H, hd, dtype=torch.bfloat16, device=DEV),
           "v": torch.zeros(1, 512, H, hd, dtype=torch.bfloat16, device=DEV), "is_init": False} for _ in range(L)]
    return kv, ca


def _logical_rows(c, name):
    """Cache rows in the reference's (logical) order: the rolling cache is a ring here, not shifted (causal_model.cache_row_map)."""
    from realtime_video_amd.causal_model import cache_row_map
    t = c[name][0]
    if int(c.get("ring_size", 0)) == 0:
        return t
    rows = cache_row_map(c).to(t.device)
    out = t.clone()
    out[:rows.numel()] = t[rows]
    return out


def _check_cache(kv, gold, tol=2e-2):
    for c, g in zip(kv, gold):
        assert int(c["global_end_index"]) == g["global_end_index"]
        assert int(c["local_end_index"]) == g["local_end_index"]
        k, v = _logical_rows(c, "k")[::197].cpu(), _logical_rows(c, "v")[::197].cpu()
        assert torch.equal(k.abs().sum((-1, -2)) == 0, g["k"].abs().sum((-1, -2)) == 0)  # same rows written
        assert rel_l2(k, g["k"]) <= tol and rel_l2(v, g["v"]) <= tol


def test_server_path_sequence_matches_reference_golden(golden):
    from oracle import wan_oracle as wo
    cfg, text_dim, tiny_inputs = _tiny()
    g = golden("dit_server_path.pt")
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    lat, ctx = tiny_inputs()
    lat = [x.to(DEV) for x in lat]
    cond = {"prompt_embeds": [ctx.to(DEV)]}
    model, wr = _build(cfg, text_dim, w)
    kv, ca = _caches(cfg, 9360)
    steps = g["steps"].to(DEV)

    def ts(v):
        return torch.ones([1, 3], dtype=torch.int64, device=DEV) * v

    flow, x0 = wr(lat[0], cond, ts(steps[0]), kv, ca, current_start=0)
    assert flow.shape == (1, 3, 16, 60, 104) and flow.dtype == torch.bfloat16
    assert rel_l2(flow.cpu(), g["b0s0_flow"]) <= 2e-2 and rel_l2(x0.cpu(), g["b0s0_x0"]) <= 2e-2
    _check_cache(kv, g["b0s0_cache"])
    assert all(c["is_init"] for c in ca)
    flow, _ = wr(lat[1], cond, ts(steps[1]), kv, ca, current_start=0)
    assert rel_l2(flow.cpu(), g["b0s1_flow"]) <= 2e-2
    _check_cache(kv, g["b0s1_cache"])
    # KV recompute pass (flex-attention branch of the reference)
    for c in kv:
        c["k"].zero_()
        c["v"].zero_()
        c["global_end_index"] = 0
        c["local_end_index"] = 0
    model.block_mask = model._prepare_blockwise_causal_attn_mask(device=DEV, num_frames=3, frame_seqlen=1560,
                                                                 num_frame_per_block=3, local_attn_size=-1)
    flow, _ = wr(lat[2], cond, torch.zeros([1, 3], dtype=torch.int64, device=DEV), kv, ca, current_start=4680)
    model.block_mask = None
    assert rel_l2(flow.cpu(), g["rc_flow"]) <= 2e-2
    _check_cache(kv, g["rc_cache"])
    flow, x0 = wr(lat[3], cond, ts(steps[0]), kv, ca, current_start=4680)
    assert rel_l2(flow.cpu(), g["b1s0_flow"]) <= 2e-2 and rel_l2(x0.cpu(), g["b1s0_x0"]) <= 2e-2
    _check_cache(kv, g["b1s0_cache"])
    assert kv[0]["local_end_index"] == 9360 and kv[0]["global_end_index"] == 9360


def test_error_vs_fp32_gold_is_within_twice_the_reference_error(golden):
    from oracle import wan_oracle as wo
    cfg, text_dim, tiny_inputs = _tiny()
    g = golden("dit_server_path.pt")
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    lat, ctx = tiny_inputs()
    wf = {k: v.float() for k, v in w.items()}
    kvc, cac = (wo.initialize_kv_cache(cfg["num_layers"], 1, 9360, cfg["num_heads"], 128, torch.float32),
                wo.initialize_crossattn_cache(cfg["num_layers"], 1, cfg["num_heads"], 128, torch.float32))
    t = torch.ones([1, 3], dtype=torch.int64) * g["steps"][0]
    gold, _ = wo.wrapper_forward(wf, cfg, wo.FlowMatchScheduler(), lat[0].float(), [ctx.float()], t, kvc, cac, 0,
                                 attn_fn=lambda q, k, v: wo.attention_sdpa(q, k, v, dtype=None))
    model, wr = _build(cfg, text_dim, w)
    kv, ca = _caches(cfg, 9360)
    flow, _ = wr(lat[0].to(DEV), {"prompt_embeds": [ctx.to(DEV)]}, t.to(DEV), kv, ca, current_start=0)
    err_ours, err_ref = max_abs(flow.cpu(), gold), max_abs(g["b0s0_flow"], gold)
    assert err_ours <= 2 * err_ref + 1e-2, (err_ours, err_ref)
    assert rel_l2(flow.cpu(), gold) <= 2 * rel_l2(g["b0s0_flow"], gold) + 2e-3


def test_rolling_cache_matches_reference_golden(golden):
    from oracle import wan_oracle as wo
    cfg, text_dim, tiny_inputs = _tiny()
    cfg.update(local_attn_size=6, sink_size=1, num_layers=1)
    g = golden("dit_rolling.pt")
    w = wo.make_weights(cfg, seed=3, text_dim=text_dim)
    lat, ctx = tiny_inputs(seed=7)
    model, wr = _build(cfg, text_dim, w)
    kv, ca = _caches(cfg, 6 * 1560)
    idx = []
    for b in range(4):
        t = torch.ones([1, 3], dtype=torch.int64, device=DEV) * 500
        flow, _ = wr(lat[b].to(DEV), {"prompt_embeds": [ctx.to(DEV)]}, t, kv, ca, current_start=b * 4680)
        idx.append((kv[0]["global_end_index"], kv[0]["local_end_index"]))
        assert rel_l2(flow[0, :, :, ::3, ::4].cpu(), g["flow_sample"][b]) <= 2e-2
    assert idx == g["indices"]
    assert kv[0]["ring_size"] == 5 * 1560 and kv[0]["ring_start"] > 0      # evictions advanced the ring: zero shift copies
    _check_cache(kv, g["cache"])


def test_production_width_layer_matches_oracle():
    """One 1.3B-width layer stack (d=1536, H=12, ffn=8960, L=2) at the real token count against the CPU
    oracle graph (bf16 eager restatement of the reference, evaluated by torch on the device)."""
    from oracle import wan_oracle as wo
    cfg = dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=2, freq_dim=256, text_len=512, eps=1e-6,
               num_frame_per_block=3)
    w = wo.make_weights(cfg, seed=5, text_dim=256)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16)
    ctx = torch.randn(40, 256, generator=g).to(torch.bfloat16)
    t = torch.tensor([[713.0, 713.0, 713.0]])
    kvc = wo.initialize_kv_cache(2, 1, 9360, 12, 128, torch.bfloat16, DEV)
    cac = wo.initialize_crossattn_cache(2, 1, 12, 128, torch.bfloat16, device=DEV)
    ref, ref_x0 = wo.wrapper_forward(_on_dev(w), cfg, wo.FlowMatchScheduler(), lat.to(DEV), [ctx.to(DEV)], t.to(DEV), kvc, cac, 0)
    model, wr = _build(cfg, 256, w)
    kv, ca = _caches(cfg, 9360)
    flow, x0 = wr(lat.to(DEV), {"prompt_embeds": [ctx.to(DEV)]}, t.to(DEV), kv, ca, current_start=0)
    assert rel_l2(flow.cpu(), ref) <= 2e-2
    assert rel_l2(x0.cpu(), ref_x0) <= 2e-2
    assert rel_l2(kv[1]["k"][0, :4680].cpu(), kvc[1]["k"][0, :4680]) <= 2e-2


def test_full_width_14b_layer_matches_reference_golden(golden):
    """BASELINE config 3 at production width against the REFERENCE ITSELF: tests/golden/dit_full_width_layer.pt is the upstream
    CausalWanModel (dim 5120, 40 heads, ffn 13824, one layer; bf16 on the host cores, SDPA fallback) run by
    oracle/make_golden.py full_width on the same seeded inputs: M = 4680 tokens at cache offset 4680 over a 9360-row window
    whose first half holds earlier K/V.  rel-L2 <= 2e-2 on flow / x0 / the new K and V rows (sampled every 97th row),
    indices exact, earlier rows untouched.  (The oracle port is bit-identical to the reference on this case: the golden
    records port_vs_reference_rel_l2 = 0.)"""
    from oracle import wan_oracle as wo
    gold = golden("dit_full_width_layer.pt")
    assert gold["port_vs_reference_rel_l2"] <= 1e-6
    cfg = dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=1, freq_dim=256, text_len=512, eps=1e-6,
               num_frame_per_block=3)
    w = wo.make_weights(cfg, seed=5, text_dim=256)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16)
    ctx = torch.randn(40, 256, generator=g).to(torch.bfloat16)
    t = torch.tensor([[713.0, 713.0, 713.0]])
    old_k = torch.randn(1, 4680, 40, 128, generator=g).to(torch.bfloat16)
    old_v = torch.randn(1, 4680, 40, 128, generator=g).to(torch.bfloat16)
    model, wr = _build(cfg, 256, w)
    kv, ca = _caches(cfg, 9360)
    kv[0]["k"][:, :4680], kv[0]["v"][:, :4680] = old_k.to(DEV), old_v.to(DEV)
    kv[0]["global_end_index"] = kv[0]["local_end_index"] = 4680
    flow, x0 = wr(lat.to(DEV), {"prompt_embeds": [ctx.to(DEV)]}, t.to(DEV), kv, ca, current_start=4680)
    assert int(kv[0]["local_end_index"]) == gold["local_end_index"] == 9360
    assert int(kv[0]["global_end_index"]) == gold["global_end_index"] == 9360
    assert rel_l2(flow.cpu(), gold["flow"]) <= 2e-2 and rel_l2(x0.cpu(), gold["x0"]) <= 2e-2
    assert rel_l2(kv[0]["k"][0, 4680::97].cpu(), gold["k_new"]) <= 2e-2
    assert rel_l2(kv[0]["v"][0, 4680::97].cpu(), gold["v_new"]) <= 2e-2
    assert torch.equal(kv[0]["k"][0, :4680].cpu(), old_k[0])
    assert abs(float(kv[0]["k"][0, :4680].double().abs().sum()) - gold["k_old_checksum"]) <= 1e-6 * gold["k_old_checksum"]


def test_full_width_14b_layer_matches_oracle_and_fp32_gold():
    """BASELINE config 3 at production width against the ORACLE (not against itself): one layer of the 14B architecture
    (d 5120, 40 heads, ffn 13824) inside the full forward (patch / time / text embeddings, head), M = 4680 query tokens at
    cache offset 4680 over a 9360-row window whose first half holds earlier K/V - vs the bf16 eager oracle graph (rel-L2 <= 2e-2)
    and vs the fp32 gold graph (error within 2x the bf16 oracle's own error).  Both oracle graphs are evaluated by torch eager on
    the device (see _on_dev); the test above holds the same case to the reference's own host evaluation."""
    from oracle import wan_oracle as wo
    cfg = dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=1, freq_dim=256, text_len=512, eps=1e-6,
               num_frame_per_block=3)
    w = wo.make_weights(cfg, seed=5, text_dim=256)
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16)
    ctx = torch.randn(40, 256, generator=g).to(torch.bfloat16)
    t = torch.tensor([[713.0, 713.0, 713.0]])
    old_k = torch.randn(1, 4680, 40, 128, generator=g).to(torch.bfloat16)
    old_v = torch.randn(1, 4680, 40, 128, generator=g).to(torch.bfloat16)

    def prefilled(dtype):
        kv = wo.initialize_kv_cache(1, 1, 9360, 40, 128, dtype, DEV)
        kv[0]["k"][:, :4680], kv[0]["v"][:, :4680] = old_k.to(DEV, dtype), old_v.to(DEV, dtype)
        kv[0]["global_end_index"] = kv[0]["local_end_index"] = 4680
        return kv, wo.initialize_crossattn_cache(1, 1, 40, 128, dtype, device=DEV)

    with torch.inference_mode():
        kvr, car = prefilled(torch.bfloat16)
        wd = _on_dev(w)
        ref, ref_x0 = wo.wrapper_forward(wd, cfg, wo.FlowMatchScheduler(), lat.to(DEV), [ctx.to(DEV)], t.to(DEV), kvr, car, 4680)
        kvg, cag = prefilled(torch.float32)
        gold, _ = wo.wrapper_forward({k: v.float() for k, v in wd.items()}, cfg, wo.FlowMatchScheduler(), lat.float().to(DEV),
                                     [ctx.float().to(DEV)], t.to(DEV), kvg, cag, 4680,
                                     attn_fn=lambda q, k, v: wo.attention_sdpa(q, k, v, dtype=None))
        del wd, kvg, cag
    assert kvr[0]["local_end_index"] == 9360
    model, wr = _build(cfg, 256, w)
    kv, ca = _caches(cfg, 9360)
    kv[0]["k"][:, :4680], kv[0]["v"][:, :4680] = old_k.to(DEV), old_v.to(DEV)
    kv[0]["global_end_index"] = kv[0]["local_end_index"] = 4680
    flow, x0 = wr(lat.to(DEV), {"prompt_embeds": [ctx.to(DEV)]}, t.to(DEV), kv, ca, current_start=4680)
    assert kv[0]["local_end_index"] == 9360 and kv[0]["global_end_index"] == 9360
    assert rel_l2(flow.cpu(), ref) <= 2e-2 and rel_l2(x0.cpu(), ref_x0) <= 2e-2
    assert rel_l2(kv[0]["k"][0, 4680:].cpu(), kvr[0]["k"][0, 4680:]) <= 2e-2
    assert rel_l2(kv[0]["v"][0, 4680:].cpu(), kvr[0]["v"][0, 4680:]) <= 2e-2
    assert torch.equal(kv[0]["k"][0, :4680].cpu(), old_k[0])              # earlier rows untouched
    err_ours, err_ref = max_abs(flow.cpu(), gold), max_abs(ref, gold)
    assert err_ours <= 2 * err_ref + 1e-2, (err_ours, err_ref)
    assert rel_l2(flow.cpu(), gold) <= 2 * rel_l2(ref, gold) + 2e-3


def test_session_block_loop_matches_oracle():
    """GenerationSession mirror (recompute + 4 denoise steps per block) vs the oracle's restatement of
    release_server.py:588-708 on the tiny model, fed the same noise stream; 2 blocks."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    g = torch.Generator().manual_seed(5)
    ctx = torch.randn(64, text_dim, generator=g).to(torch.bfloat16)
    noise = torch.randn(1, 6, 16, 60, 104, generator=g).to(torch.bfloat16)
    ora = wo.SessionOracle(_on_dev(w), cfg, [ctx.to(DEV)], noise.to(DEV), kv_cache_num_frames=3, num_steps=4, shift=5.0, seed=9)
    ref_blocks = [ora.generate_block().clone() for _ in range(2)]

    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]),
                                   DEV, generator=wr, text_encoder=None, vae=None)
    padded = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16)
    padded[0, :64] = ctx
    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded.to(DEV)))
    sess = GenerationSession(GenerateParams(seed=9, num_blocks=2, num_denoising_steps=4, keep_first_frame=True),
                             models, device=DEV)
    sess.noise = noise.to(DEV)
    cpu_rnd = torch.Generator().manual_seed(9)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    assert torch.equal(sess.denoising_step_list.cpu(), ora.denoising_step_list)
    for b in range(2):
        out = sess.generate_block()
        assert rel_l2(out.cpu(), ref_blocks[b]) <= 5e-2, b
    assert sess.current_start_frame == 6 and pipe.kv_cache1[0]["local_end_index"] == 9360
    assert pipe.kv_cache1[0]["k"].shape == (1, 9360, cfg["num_heads"], 128)


def test_session_at_another_resolution_uses_its_own_frame_length(monkeypatch):
    """416 x 240 (latent 30 x 52 -> 390 tokens per frame): the session derives frame_seq_length from the latent grid - cache
    of (c + 3) * 390 rows, current_start in units of 390, RoPE frame = current_start // 390 - and matches the oracle run with
    the reference's hard-coded 1560 replaced by the same 390 (at 1560 the reference leaves gaps in the window, SURVEY 8a trap 3);
    with the index-only cache reset every row of the window is written before it is read."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    monkeypatch.setattr(wo, "FRAME_SEQLEN", 390)
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    g = torch.Generator().manual_seed(6)
    ctx = torch.randn(64, text_dim, generator=g).to(torch.bfloat16)
    noise = torch.randn(1, 9, 16, 30, 52, generator=g).to(torch.bfloat16)
    ora = wo.SessionOracle(_on_dev(w), cfg, [ctx.to(DEV)], noise.to(DEV), kv_cache_num_frames=3, num_steps=4, shift=5.0, seed=9)
    ref_blocks = [ora.generate_block().clone() for _ in range(3)]
    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]),
                                   DEV, generator=wr, text_encoder=None, vae=None)
    padded = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16)
    padded[0, :64] = ctx
    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded.to(DEV)))
    sess = GenerationSession(GenerateParams(seed=9, num_blocks=3, num_denoising_steps=4, keep_first_frame=True,
                                            width=416, height=240), models, device=DEV)
    assert pipe.frame_seq_length == 390 and pipe.kv_cache1[0]["k"].shape == (1, 6 * 390, cfg["num_heads"], 128)
    sess.noise = noise.to(DEV)
    cpu_rnd = torch.Generator().manual_seed(9)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    pipe._kv_arena.fill_(float("nan"))          # a row read before it is written would poison the output
    for b in range(3):
        out = sess.generate_block()
        assert torch.isfinite(out.float()).all(), b
        assert rel_l2(out.cpu(), ref_blocks[b]) <= 5e-2, b
    assert pipe.kv_cache1[0]["local_end_index"] == 6 * 390


def test_attention_window_is_counted_in_tokens_at_every_resolution(monkeypatch):
    """the self-attention window is max_attention_size TOKENS (32760, or local_attn_size * 1560) at every
    resolution, as in the reference (causal_model.py:192, :388-389) - not a number of frames.  416 x 240 (390 tokens per frame)
    with the pipeline's 32760-row cache: after 8 blocks 9360 rows are live, more than 21 frames' worth (8190) - a frame-counted
    window would drop the oldest three frames, the reference (and the oracle with the same frame length) attends them all."""
    from oracle import wan_oracle as wo
    monkeypatch.setattr(wo, "FRAME_SEQLEN", 390)
    cfg, text_dim, _ = _tiny()
    cfg["num_layers"] = 1
    w = wo.make_weights(cfg, seed=2, text_dim=text_dim)
    g = torch.Generator().manual_seed(8)
    ctx = torch.randn(32, text_dim, generator=g).to(torch.bfloat16)
    lat = torch.randn(8, 1, 3, 16, 30, 52, generator=g).to(torch.bfloat16)
    kvc = wo.initialize_kv_cache(1, 1, 32760, cfg["num_heads"], 128, torch.bfloat16, DEV)
    cac = wo.initialize_crossattn_cache(1, 1, cfg["num_heads"], 128, torch.bfloat16, device=DEV)
    model, wr = _build(cfg, text_dim, w)
    kv, ca = _caches(cfg, 32760)
    t = torch.ones([1, 3], dtype=torch.int64, device=DEV) * 500
    wd = _on_dev(w)
    for b in range(8):
        ref, _ = wo.wrapper_forward(wd, cfg, wo.FlowMatchScheduler(), lat[b].to(DEV), [ctx.to(DEV)], t, kvc, cac, b * 1170)
        flow, _ = wr(lat[b].to(DEV), {"prompt_embeds": [ctx.to(DEV)]}, t, kv, ca, current_start=b * 1170)
        assert rel_l2(flow, ref) <= 2e-2, b
    assert kv[0]["local_end_index"] == kvc[0]["local_end_index"] == 9360


@pytest.mark.parametrize("world,exchange,heads", [(2, "rows", 2), (8, "rows", 2), (2, "heads", 2), (4, "heads", 8),
                                                  (8, "heads", 8)])
def test_context_parallel_phase_api_equals_unsharded(world, exchange, heads):
    """Token-axis sharding (rtv_dit_begin / layer_qkv / layer_rest / head / finish with row ranges) with either exchange
    around self-attention - "rows": K/V all-gather into a replicated cache; "heads": the all-to-all pair of
    rtv_dit_layer_{qkv,attn,rest}_hp, every rank attending all rows for its own heads.  All shards run in lockstep on
    this one GPU and must reproduce the unsharded forward bit for bit - denoise pass at a non-zero cache offset and the
    block-causal recompute pass.  Tile config 4 (no split-K): the default dispatch cuts K by a launch's tile count, i.e. by the
    shard's row count and by which projections share a launch (r05: the unsharded forward runs the V projection as a launch of its own,
    straight into the cache), and a K-split tile sums its halves in another order (profiles/r04_gemm_shard_identity.log)."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.parallel import SimulatedContextParallel
    cfg, text_dim, tiny_inputs = _tiny()
    cfg.update(num_heads=heads, dim=128 * heads)
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    lat, ctx = tiny_inputs()
    cond = {"prompt_embeds": [ctx.to(DEV)]}
    outs = []
    for cp in (None, SimulatedContextParallel(world, exchange)):
        model, wr = _build(cfg, text_dim, w)
        model.context_parallel = cp
        model.gemm_tile_cfg = 4
        assert model.kv_cache_heads() == heads          # a simulation shares one full-head cache
        kv, ca = _caches(cfg, 9360)
        t = torch.ones([1, 3], dtype=torch.int64, device=DEV) * 700
        model.block_mask = model._prepare_blockwise_causal_attn_mask(device=DEV, num_frames=3, frame_seqlen=1560,
                                                                     num_frame_per_block=3)
        f_rc, _ = wr(lat[2].to(DEV), cond, torch.zeros([1, 3], dtype=torch.int64, device=DEV), kv, ca, current_start=4680)
        model.block_mask = None
        f_dn, _ = wr(lat[3].to(DEV), cond, t, kv, ca, current_start=4680)
        outs.append((f_rc.clone(), f_dn.clone(), kv[1]["k"].clone(), kv[1]["v"].clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("world,exchange,heads,splits", [(8, "heads", 8, 2), (4, "heads", 8, 3), (8, "rows", 2, 2)])
def test_context_parallel_kv_split_attention_within_tolerance(world, exchange, heads, splits):
    """ContextParallel(attn_kv_splits=S): every rank's dense self-attention launch cut along the keys (rtv_attn_fwd_split) - the
    launch of a rank has `world` times fewer workgroups than the unsharded one.  Not bit-identical with the unsharded forward
    (fp32 summation order of the softmax sums), hence off by default; stated tolerance: rel-L2 <= 3e-3 on the flow of a denoise
    pass and of the block-causal recompute pass, K / V rows written to the cache bit-identical in layer 0 (they do not depend on
    attention) and within 3e-3 in layer 1."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.parallel import SimulatedContextParallel
    cfg, text_dim, tiny_inputs = _tiny()
    cfg.update(num_heads=heads, dim=128 * heads)
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    lat, ctx = tiny_inputs()
    cond = {"prompt_embeds": [ctx.to(DEV)]}
    outs = []
    for cp in (None, SimulatedContextParallel(world, exchange, attn_kv_splits=splits)):
        model, wr = _build(cfg, text_dim, w)
        model.context_parallel = cp
        model.gemm_tile_cfg = 4     # shard-invariant GEMMs (see test_context_parallel_phase_api_equals_unsharded): layer 0's K is compared bitwise
        kv, ca = _caches(cfg, 9360)
        t = torch.ones([1, 3], dtype=torch.int64, device=DEV) * 700
        model.block_mask = model._prepare_blockwise_causal_attn_mask(device=DEV, num_frames=3, frame_seqlen=1560,
                                                                     num_frame_per_block=3)
        f_rc, _ = wr(lat[2].to(DEV), cond, torch.zeros([1, 3], dtype=torch.int64, device=DEV), kv, ca, current_start=4680)
        rc_k = kv[1]["k"].clone()
        model.block_mask = None
        f_dn, _ = wr(lat[3].to(DEV), cond, t, kv, ca, current_start=4680)
        outs.append((f_rc.clone(), rc_k, f_dn.clone(), kv[0]["k"].clone(), kv[1]["k"].clone(), kv[1]["v"].clone()))
    (rc0, rck0, dn0, k00, k10, v10), (rc1, rck1, dn1, k01, k11, v11) = outs
    assert 0 < rel_l2(rc1, rc0) <= 3e-3 and rel_l2(rck1, rck0) <= 3e-3
    assert torch.equal(k00, k01)
    assert 0 < rel_l2(dn1, dn0) <= 3e-3
    assert rel_l2(k11, k10) <= 3e-3 and rel_l2(v11, v10) <= 3e-3


def test_session_first_frame_reencode_path():
    """keep_first_frame=False (the reference default): from block 2 on, get_clean_context_frames re-encodes the oldest
    pixel frame of the context window through the VAE encoder (release_server.py:572-575).  The re-encoded latent is
    checked against the encoder oracle (eager fp16 on this GPU) on the same pixel frame; the DiT blocks are checked
    against the session oracle fed with that latent."""
    from oracle import vae_oracle as vo
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    g = torch.Generator().manual_seed(5)
    ctx = torch.randn(64, text_dim, generator=g).to(torch.bfloat16)
    noise = torch.randn(1, 9, 16, 60, 104, generator=g).to(torch.bfloat16)

    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]),
                                   DEV, generator=wr, text_encoder=None, vae=None)
    padded = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16)
    padded[0, :64] = ctx
    enc_w = vo.make_vae_encoder_weights(seed=1)
    enc = VAEEncoderWrapper(device=DEV)
    enc.load_state_dict(enc_w)
    calls = []

    def recording_encoder(frames, cache, stream=False):
        mu, c = enc(frames, cache, stream=stream)
        calls.append((frames.clone(), mu.clone()))
        return mu, c

    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded.to(DEV)),
                    vae_decoder=VAEDecoderWrapper(DEV).init_random_weights(), vae_encoder=recording_encoder)
    sess = GenerationSession(GenerateParams(seed=9, num_blocks=3, num_denoising_steps=4, keep_first_frame=False),
                             models, device=DEV)
    sess.noise = noise.to(DEV)
    cpu_rnd = torch.Generator().manual_seed(9)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    outs, oldest = [], None
    for b in range(3):
        if b == 2:
            oldest = sess.frame_context_cache[0].clone()       # [1, 1, 3, 480, 832]: what block 2 must re-encode
        outs.append(sess.generate_block())
    assert [o.shape[1] for o in outs] == [6, 12, 12]            # 9 - 3 dropped, then 12 per block
    assert len(calls) == 1                                       # blocks 0, 1 keep the first latent; block 2 re-encodes
    frames, mu = calls[0]
    assert frames.shape == (1, 3, 1, 480, 832) and mu.shape == (1, 16, 1, 60, 104)
    assert torch.equal(frames[0, :, 0], oldest[0, 0].half())
    w16 = {k: v.half().to(DEV) for k, v in enc_w.items()}
    mu_ref, _ = vo.encoder_wrapper_forward(w16, frames, [None] * 55, stream=False)
    assert rel_l2(mu.float(), mu_ref.float()) <= 2e-2

    ora = wo.SessionOracle(_on_dev(w), cfg, [ctx.to(DEV)], noise.to(DEV), kv_cache_num_frames=3, num_steps=4, shift=5.0, seed=9,
                           first_frame_fn=lambda idx: mu.permute(0, 2, 1, 3, 4).to(torch.bfloat16))
    for b in range(3):
        ref = ora.generate_block()
        assert rel_l2(sess.all_latents[:, 3 * b:3 * b + 3].cpu(), ref) <= 5e-2, b


def test_session_long_context_kv_cache_num_frames_9():
    """BASELINE config 5's context length: kv_cache_num_frames = 9 (Lkv = 18720, recompute over up to 9 context frames
    = 14040 tokens with the block-causal mask over three 3-frame blocks).  Four blocks on the tiny model with
    keep_first_frame=True vs the session oracle: block 3 recomputes over all nine earlier frames."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    cfg, text_dim, _ = _tiny()
    cfg["num_layers"] = 1
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    g = torch.Generator().manual_seed(15)
    ctx = torch.randn(64, text_dim, generator=g).to(torch.bfloat16)
    noise = torch.randn(1, 12, 16, 60, 104, generator=g).to(torch.bfloat16)
    ora = wo.SessionOracle(_on_dev(w), cfg, [ctx.to(DEV)], noise.to(DEV), kv_cache_num_frames=9, num_steps=2, shift=5.0, seed=4)
    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 500]), DEV, generator=wr,
                                   text_encoder=None, vae=None)
    padded = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16)
    padded[0, :64] = ctx
    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded.to(DEV)))
    sess = GenerationSession(GenerateParams(seed=4, num_blocks=4, num_denoising_steps=2, kv_cache_num_frames=9,
                                            keep_first_frame=True), models, device=DEV)
    sess.noise = noise.to(DEV)
    cpu_rnd = torch.Generator().manual_seed(4)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    for b in range(4):
        out = sess.generate_block()
        ref = ora.generate_block()
        assert rel_l2(out.cpu(), ref) <= 5e-2, b
    kv = pipe.kv_cache1[0]
    assert kv["k"].shape[1] == 12 * 1560 and kv["local_end_index"] == 12 * 1560 and kv["global_end_index"] == 12 * 1560


def test_session_long_form_fp8_window_slides_at_kv_cache_num_frames_9():
    """BASELINE config 5 put together on one GPU: long-form generation (more blocks than the context window holds, so the
    recompute pass runs over first frame + the last 8 frames from block 4 on: release_server.py:563-576, :588-633), kv_cache_num_frames
    = 9, the fp8 weight path (:179-182).  Eight blocks, two layers, tiny width, latents 60 x 104, against the session oracle
    with the fp8 restatement, and the bf16 session oracle beside it as the scale of the quantisation noise.
    Tolerance: generation is autoregressive (block b's context is the output of blocks < b), so the two fp8 implementations
    drift apart by their per-forward difference compounded - stated per block as <= 1.5e-2 and <= the fp8 oracle's own distance
    from the bf16 oracle at that block (measured: 4.7e-3 .. 7.7e-3 against 1.2e-2 of quantisation noise, flat over the eight
    blocks); the cache bookkeeping must be exact."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    cfg, text_dim, _ = _tiny()
    cfg["num_layers"] = 2
    blocks = 8
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    w8 = dict(w)
    w8[wo.FP8_FLAG] = True
    g = torch.Generator().manual_seed(21)
    ctx = torch.randn(64, text_dim, generator=g).to(torch.bfloat16)
    noise = torch.randn(1, 3 * blocks, 16, 60, 104, generator=g).to(torch.bfloat16)
    oras = [wo.SessionOracle(_on_dev(ww), cfg, [ctx.to(DEV)], noise.to(DEV), kv_cache_num_frames=9, num_steps=2, shift=5.0, seed=4)
            for ww in (w8, w)]
    model, wr = _build(cfg, text_dim, w)
    model.enable_fp8()
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 500]), DEV, generator=wr,
                                   text_encoder=None, vae=None)
    padded = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16)
    padded[0, :64] = ctx
    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded.to(DEV)))
    sess = GenerationSession(GenerateParams(seed=4, num_blocks=blocks, num_denoising_steps=2, kv_cache_num_frames=9,
                                            keep_first_frame=True), models, device=DEV)
    sess.noise = noise.to(DEV)
    cpu_rnd = torch.Generator().manual_seed(4)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    report = []
    for b in range(blocks):
        out = sess.generate_block().cpu()
        ref8, ref16 = (o.generate_block().cpu() for o in oras)
        noise_b = rel_l2(ref8, ref16)
        report.append((b, rel_l2(out, ref8), noise_b, rel_l2(out, ref16)))
        print("long-form fp8 c=9 block %d: rel_l2(ours, fp8 oracle) %.3e  (fp8 oracle, bf16 oracle) %.3e  (ours, bf16 oracle) %.3e"
              % report[-1])
    for b, d, noise_b, d16 in report:
        assert d <= 1.5e-2 and d <= noise_b, report
        assert d16 <= 1.5 * noise_b, report
    kv = pipe.kv_cache1[0]
    assert kv["k"].shape[1] == 12 * 1560 and kv["local_end_index"] == 12 * 1560 and kv["global_end_index"] == 12 * 1560
    assert sess.current_start_frame == 3 * blocks and oras[0].current_start_frame == 3 * blocks


def test_session_webcam_v2v_and_prompt_interpolation():
    """Streaming video-to-video (release_server.py:489-527, :651-657): block 0 encodes 9 pushed frames on fresh encoder
    caches (chunks 1+4+4), block 1 encodes 12 with stream=True; denoising starts from latents noised to the first step's
    level.  Checked: the encoded latents vs the encoder oracle (eager fp16 on this GPU, same cache continuation), the DiT
    blocks vs the session oracle started from the same noisy latents, and a prompt interpolation that re-initialises
    the cross-attention cache (:459-468, :662-666)."""
    from oracle import vae_oracle as vo
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder, resample_array
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    assert resample_array(list(range(15)), 12) == [0, 1, 3, 4, 5, 6, 8, 9, 10, 11, 13, 14]
    cfg, text_dim, _ = _tiny()
    cfg["num_layers"] = 1
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    g = torch.Generator().manual_seed(7)
    ctx = [torch.randn(64, text_dim, generator=g).to(torch.bfloat16) for _ in range(2)]
    frames = torch.rand(21, 3, 480, 832, generator=g) * 2 - 1
    eps = torch.randn(2, 1, 3, 16, 60, 104, generator=g).to(torch.bfloat16)

    class TwoPrompts:   # text encoder stub: prompt "a" / "b" -> padded embeddings
        def __call__(self, text_prompts=None):
            e = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16)
            e[0, :64] = ctx[0 if text_prompts[0] == "a" else 1]
            return {"prompt_embeds": e.to(DEV)}

    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 500]), DEV, generator=wr)
    enc_w = vo.make_vae_encoder_weights(seed=1)
    enc = VAEEncoderWrapper(device=DEV)
    enc.load_state_dict(enc_w)
    models = Models(transformer=wr, pipeline=pipe, text_encoder=TwoPrompts(), vae_encoder=enc)
    sess = GenerationSession(GenerateParams(prompt="a", seed=1, num_blocks=2, num_denoising_steps=2, strength=0.7,
                                            webcam_mode=True, keep_first_frame=True), models, device=DEV)
    cpu_rnd = torch.Generator().manual_seed(1)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    seen = []

    def fixed_randn_like(t):
        seen.append(t.clone())
        return eps[len(seen) - 1].to(DEV)
    sess._randn_like = fixed_randn_like

    assert sess.generate_block_internal(models) is None          # no frames queued yet
    w16 = {k: v.half().to(DEV) for k, v in enc_w.items()}
    cache16 = [None] * 55
    noisy = []
    strength = float(sess.denoising_step_list[0]) / 1000.0
    assert abs(strength - float(wo.get_denoising_schedule(sess.zero_padded_timesteps.cpu(), 0.7, steps=2)[0]) / 1000.0) < 1e-6
    for b, n in enumerate((9, 12)):
        chunk = frames[:9] if b == 0 else frames[9:21]
        for f in chunk:
            sess.push_frame(f.half())
        if b == 1:
            sess.interpolate_prompt_embeds(models, "b", 2)        # lerp weights linspace(0, 1, 2) = [0, 1]: this block still
            assert len(sess.interpolated_prompt_embeds) == 2      # sees prompt "a" (fresh cross-attn cache), the next one "b"
        out = sess.generate_block()
        assert out.shape == (1, 3, 16, 60, 104) and len(seen) == b + 1
        mu16, cache16 = vo.encoder_wrapper_forward(w16, chunk.transpose(0, 1).unsqueeze(0).half().to(DEV), cache16, stream=b > 0)
        lat_ref = mu16.movedim(1, 2).to(torch.bfloat16)
        assert rel_l2(seen[b].float(), lat_ref.float()) <= 2e-2, b
        noisy.append((seen[b] * (1.0 - strength) + eps[b].to(DEV) * strength).cpu())
    assert torch.equal(sess.current_prompt_embeds[0, :64].cpu(), ctx[0])
    assert len(sess.interpolated_prompt_embeds) == 1 and torch.equal(sess.interpolated_prompt_embeds[0][0, :64].cpu(), ctx[1])
    assert all(c["is_init"] for c in pipe.crossattn_cache)

    # DiT side vs the oracle, started from the same noisy latents (the oracle reads them from its noise tensor)
    ora = wo.SessionOracle(_on_dev(w), cfg, [ctx[0].to(DEV)], torch.cat(noisy, dim=1).to(DEV), kv_cache_num_frames=3, num_steps=2,
                           shift=5.0, seed=1)
    ora.denoising_step_list = sess.denoising_step_list.cpu()
    for b in range(2):
        if b == 1:   # prompt switch AFTER the KV recompute (which still sees the old prompt's cross-attention cache)
            orig_recompute = ora.recompute_kv_cache

            def recompute_then_switch():
                start = orig_recompute()
                ora.prompt_embeds = [ctx[0].to(DEV)]  # lerp weight 0: same prompt, but the cache is rebuilt
                for c in ora.crossattn_cache:
                    c["is_init"] = False
                return start
            ora.recompute_kv_cache = recompute_then_switch
        ref = ora.generate_block()
        assert rel_l2(sess.all_latents[:, 3 * b:3 * b + 3].cpu(), ref) <= 5e-2, b


def test_config1_320x192_native_block_and_vae():
    """BASELINE config 1's size on the native path: 320x192 (latents 24x40, 240 tokens per frame), 1.3B widths (one
    layer), one block, one denoising step, VAE decode of the block (9 - 3 = 6 frames of 192x320).  Block 0 vs the
    oracle (the reference's hard-coded frame_seqlen = 1560 only matters from block 1 on, SURVEY trap 3)."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.causal_model import CausalWanModel
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    from realtime_video_amd.wan_wrapper import WanDiffusionWrapper
    cfg = dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=1, freq_dim=256, text_len=512, eps=1e-6)
    w = wo.make_weights(cfg, seed=0, text_dim=64)
    g = torch.Generator().manual_seed(2)
    noise = torch.randn(1, 3, 16, 24, 40, generator=g).to(torch.bfloat16)
    ctx = torch.randn(16, 64, generator=g).to(torch.bfloat16)
    ref = wo.SessionOracle(_on_dev(w), cfg, [ctx.to(DEV)], noise.to(DEV), kv_cache_num_frames=3, num_steps=1, shift=5.0,
                           seed=0).generate_block()
    m = CausalWanModel(dim=1536, ffn_dim=8960, num_heads=12, num_layers=1, text_dim=64, freq_dim=256, device=DEV)
    m.load_state_dict(w)
    wr = WanDiffusionWrapper(m, timestep_shift=5.0)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000]), DEV, generator=wr)
    padded = torch.zeros(1, 512, 64, dtype=torch.bfloat16)
    padded[0, :16] = ctx
    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded.to(DEV)),
                    vae_decoder=VAEDecoderWrapper(DEV).init_random_weights(seed=1))
    sess = GenerationSession(GenerateParams(width=320, height=192, seed=0, num_blocks=1, num_denoising_steps=1,
                                            keep_first_frame=True), models, device=DEV)
    sess.noise = noise.to(DEV)
    px = sess.generate_block()
    assert px.shape == (1, 6, 3, 192, 320) and torch.isfinite(px).all() and float(px.abs().max()) <= 1.0
    assert rel_l2(sess.all_latents.cpu(), ref) <= 5e-2


def test_v_cache_rows_written_by_the_projection_gemm_equal_the_copied_ones():
    """r05: the V third of the fused QKV projection runs as a GEMM launch of its own whose output matrix IS the call's rows of the V cache (dit_layer_proj, csrc/dit_forward.hip)
    whenever the call's cache rows are one physical range; the RoPE / cache kernel then skips its V copy.  Pure data movement: flow
    prediction and the whole K / V cache must equal the r04 form (rtv_dit_set_direct_v(0): V copied out of the projection output) bit
    for bit - a denoise step at cache offset 0, the block-causal recompute pass, a second-block step at offset 4680."""
    from oracle import wan_oracle as wo
    from realtime_video_amd import _lib
    lib = _lib.load()
    cfg, text_dim, tiny_inputs = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    lat, ctx = tiny_inputs()
    t = torch.ones([1, 3], dtype=torch.int64) * 700
    t0 = torch.zeros([1, 3], dtype=torch.int64)
    res = {}
    try:
        for mode in (0, 1):
            lib.rtv_dit_set_direct_v(mode)
            model, wr = _build(cfg, text_dim, w)
            kv, ca = _caches(cfg, 9360)
            for c in kv:                      # poison: a row the GEMM fails to write would show
                c["v"].fill_(7.0)
                c["k"].fill_(7.0)
            cond = {"prompt_embeds": [ctx.to(DEV)]}
            a, _ = wr(lat[0].to(DEV), cond, t.to(DEV), kv, ca, current_start=0)
            for c in kv:
                c["global_end_index"] = 0
                c["local_end_index"] = 0
            model.block_mask = model._prepare_blockwise_causal_attn_mask(device=DEV, num_frames=3, frame_seqlen=1560,
                                                                         num_frame_per_block=3)
            rc, _ = wr(lat[2].to(DEV), cond, t0.to(DEV), kv, ca, current_start=4680)
            model.block_mask = None
            b, _ = wr(lat[3].to(DEV), cond, t.to(DEV), kv, ca, current_start=4680)
            res[mode] = ([x.clone() for x in (a, rc, b)], [(c["k"].clone(), c["v"].clone()) for c in kv])
    finally:
        lib.rtv_dit_set_direct_v(1)
    for x0, x1 in zip(res[0][0], res[1][0]):
        assert torch.equal(x0, x1)
    for (k0, v0), (k1, v1) in zip(res[0][1], res[1][1]):
        assert torch.equal(k0, k1) and torch.equal(v0, v1)
        assert not (v1[:, :9360] == 7.0).all()


def test_fp8_forward_matches_fp8_oracle():
    """BASELINE config 5's weight path (release_server.py:179-182, torchao Float8DynamicActivationFloat8WeightConfig
    PerTensor over every nn.Linear): native enable_fp8() forward vs the oracle's restatement of the same arithmetic, on a
    denoise step at cache offset 0, the block-causal recompute pass and a second-block denoise step.  Tolerance: both
    sides take identical quantisation decisions except where upstream bf16 rounding differs, so rel-L2 <= 3e-2 against
    the fp8 oracle; the fp8 result must also stay within 10 % rel-L2 of the bf16 model (quantisation noise, not a bug)."""
    from oracle import wan_oracle as wo
    cfg, text_dim, tiny_inputs = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    w8 = dict(w)
    w8[wo.FP8_FLAG] = True
    lat, ctx = tiny_inputs()
    sched = wo.FlowMatchScheduler()
    kvc = wo.initialize_kv_cache(cfg["num_layers"], 1, 9360, cfg["num_heads"], 128, torch.bfloat16, DEV)
    cac = wo.initialize_crossattn_cache(cfg["num_layers"], 1, cfg["num_heads"], 128, torch.bfloat16, device=DEV)
    w8, ctx_d, lat_d = _on_dev(w8), ctx.to(DEV), [x.to(DEV) for x in lat]
    t = torch.ones([1, 3], dtype=torch.int64) * 700
    t0 = torch.zeros([1, 3], dtype=torch.int64)
    ref_a, _ = wo.wrapper_forward(w8, cfg, sched, lat_d[0], [ctx_d], t.to(DEV), kvc, cac, 0)
    wo.reset_kv_cache(kvc)
    ref_rc, _ = wo.wrapper_forward(w8, cfg, sched, lat_d[2], [ctx_d], t0.to(DEV), kvc, cac, 4680, recompute=True)
    ref_b, _ = wo.wrapper_forward(w8, cfg, sched, lat_d[3], [ctx_d], t.to(DEV), kvc, cac, 4680)

    outs = {}
    for mode in ("bf16", "fp8"):
        model, wr = _build(cfg, text_dim, w)
        if mode == "fp8":
            model.enable_fp8()
        kv, ca = _caches(cfg, 9360)
        cond = {"prompt_embeds": [ctx.to(DEV)]}
        a, _ = wr(lat[0].to(DEV), cond, t.to(DEV), kv, ca, current_start=0)
        for c in kv:
            c["global_end_index"] = 0
            c["local_end_index"] = 0
        model.block_mask = model._prepare_blockwise_causal_attn_mask(device=DEV, num_frames=3, frame_seqlen=1560,
                                                                     num_frame_per_block=3)
        rc, _ = wr(lat[2].to(DEV), cond, t0.to(DEV), kv, ca, current_start=4680)
        model.block_mask = None
        b, _ = wr(lat[3].to(DEV), cond, t.to(DEV), kv, ca, current_start=4680)
        outs[mode] = [x.cpu() for x in (a, rc, b)]
    for ours, ref in zip(outs["fp8"], (ref_a, ref_rc, ref_b)):
        assert rel_l2(ours, ref) <= 3e-2
    for f8, bf in zip(outs["fp8"], outs["bf16"]):
        assert 0 < rel_l2(f8, bf) <= 0.1


@pytest.mark.parametrize("exchange", ["heads", "rows"])
def test_fp8_context_parallel_matches_row_sharded_fp8_oracle(exchange):
    """BASELINE config 5 is fp8 on 8 GPUs: under context parallelism every rank's linears quantise the token rows that rank
    holds with their own dynamic scale (what a per-rank torchao linear does).  Two shards in lockstep on this GPU vs the
    oracle with FP8_ROW_SHARDS = (2, 4680) - recompute pass and a denoise step at cache offset 4680 - and vs the unsharded
    fp8 forward (differs by quantisation noise only).  Tolerance 4e-2: a dynamic scale is max|x|/448, so wherever the
    upstream bf16 rounding of the two implementations moves that one maximum by an ulp, every rounding boundary of the
    tensor moves with it and ~5-10 % of its e4m3 codes change - as much error power as the quantisation itself.  Two valid
    scale choices (oracle sharded vs oracle whole) are 1.8e-2 apart on these inputs; with two shards twice as many scales
    can differ (measured 2.5e-2 / 3.2e-2; unsharded 1.3e-2 / 2.2e-2)."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.parallel import SimulatedContextParallel
    cfg, text_dim, tiny_inputs = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    w8 = dict(w)
    w8[wo.FP8_FLAG] = True
    w8[wo.FP8_ROW_SHARDS] = (2, 4680)
    lat, ctx = tiny_inputs()
    sched = wo.FlowMatchScheduler()
    kvc = wo.initialize_kv_cache(cfg["num_layers"], 1, 9360, cfg["num_heads"], 128, torch.bfloat16, DEV)
    cac = wo.initialize_crossattn_cache(cfg["num_layers"], 1, cfg["num_heads"], 128, torch.bfloat16, device=DEV)
    w8, ctx_d, lat_d = _on_dev(w8), ctx.to(DEV), [x.to(DEV) for x in lat]
    t = torch.ones([1, 3], dtype=torch.int64) * 700
    t0 = torch.zeros([1, 3], dtype=torch.int64)
    ref_rc, _ = wo.wrapper_forward(w8, cfg, sched, lat_d[2], [ctx_d], t0.to(DEV), kvc, cac, 4680, recompute=True)
    ref_b, _ = wo.wrapper_forward(w8, cfg, sched, lat_d[3], [ctx_d], t.to(DEV), kvc, cac, 4680)
    outs = []
    for cp in (SimulatedContextParallel(2, exchange), None):
        model, wr = _build(cfg, text_dim, w)
        model.context_parallel = cp
        model.enable_fp8()
        kv, ca = _caches(cfg, 9360)
        cond = {"prompt_embeds": [ctx.to(DEV)]}
        model.block_mask = model._prepare_blockwise_causal_attn_mask(device=DEV, num_frames=3, frame_seqlen=1560,
                                                                     num_frame_per_block=3)
        rc, _ = wr(lat[2].to(DEV), cond, t0.to(DEV), kv, ca, current_start=4680)
        model.block_mask = None
        b, _ = wr(lat[3].to(DEV), cond, t.to(DEV), kv, ca, current_start=4680)
        outs.append((rc.cpu(), b.cpu()))
    for ours, ref in zip(outs[0], (ref_rc, ref_b)):
        assert rel_l2(ours, ref) <= 4e-2
    for sharded, whole in zip(*outs):
        assert 0 < rel_l2(sharded, whole) <= 5e-2


def test_cross_attention_padding_fold_matches_full_text_window(golden):
    """`fold_text_padding` (default): the cross-attention attends the real prompt rows plus ONE of the zero-padding rows weighted by
    their count (all padding rows have the same cached K / V row) instead of all 512 text rows.  Against the same model with the
    switch off - server-path sequence of the reference golden (64 prompt rows of 512): rel-L2 <= 3e-3 on every output, the text
    K / V caches are bit-identical, the padding rows of the cache really are identical rows - and both are within the golden
    tolerance of the reference."""
    from oracle import wan_oracle as wo
    g = golden("dit_server_path.pt")
    cfg, text_dim, tiny_inputs = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    lat, ctx = tiny_inputs()
    t = torch.ones([1, 3], dtype=torch.int64) * g["steps"][0]
    outs = {}
    for fold in (True, False):
        model, wr = _build(cfg, text_dim, w)
        model.fold_text_padding = fold
        kv, ca = _caches(cfg, 9360)
        cond = {"prompt_embeds": [ctx.to(DEV)]}
        a, _ = wr(lat[0].to(DEV), cond, t.to(DEV), kv, ca, current_start=0)
        b, _ = wr(lat[3].to(DEV), cond, t.to(DEV), kv, ca, current_start=4680)
        assert ca[0].get("text_rows") == ctx.shape[0]
        outs[fold] = (a, b, [c["k"].clone() for c in ca], [c["v"].clone() for c in ca])
    for i in range(2):
        assert rel_l2(outs[True][i], outs[False][i]) <= 3e-3
    assert rel_l2(outs[True][0].cpu(), g["b0s0_flow"]) <= 2e-2 and rel_l2(outs[False][0].cpu(), g["b0s0_flow"]) <= 2e-2
    for x, y in zip(outs[True][2] + outs[True][3], outs[False][2] + outs[False][3]):
        assert torch.equal(x, y)
    n = ctx.shape[0]
    for kc in outs[True][2] + outs[True][3]:
        assert torch.equal(kc[0, n:], kc[0, n:n + 1].expand_as(kc[0, n:]))      # every padding row is the same row


@pytest.mark.parametrize("cp_world", [0, 2])
def test_kv_cache_only_forward_fills_the_same_cache(cp_world):
    """`kv_cache_only=True` (a per-call argument; what the session passes for its KV-recompute pass, whose output the reference discards,
    release_server.py:611-632): the forward stops behind the last layer's cache write - the KV caches of EVERY layer are bit-identical
    to those of the full recompute forward, the indices advance the same way, the returned tensor is zeros; with the
    cross-attention caches still uninitialised the switch is ignored (the last layer's text K / V are computed in its rest
    phase).  Also through the context-parallel phase API (two shards in lockstep)."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.parallel import SimulatedContextParallel
    cfg, text_dim, tiny_inputs = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    lat, ctx = tiny_inputs()
    t, t0 = torch.ones([1, 3], dtype=torch.int64) * 700, torch.zeros([1, 3], dtype=torch.int64)
    res = {}
    for only in (False, True):
        model, wr = _build(cfg, text_dim, w)
        if cp_world:
            model.context_parallel = SimulatedContextParallel(cp_world, "heads")
        kv, ca = _caches(cfg, 9360)
        cond = {"prompt_embeds": [ctx.to(DEV)]}
        model.block_mask = model._prepare_blockwise_causal_attn_mask(device=DEV, num_frames=3, frame_seqlen=1560, num_frame_per_block=3)
        first, _ = wr(lat[2].to(DEV), cond, t0.to(DEV), kv, ca, current_start=4680, kv_cache_only=only)   # cross caches uninitialised: full forward
        assert all(c["is_init"] for c in ca)
        for c in kv:
            c["global_end_index"] = c["local_end_index"] = 0
        rc, _ = wr(lat[1].to(DEV), cond, t0.to(DEV), kv, ca, current_start=4680, kv_cache_only=only)      # the switch takes effect here
        model.block_mask = None
        nxt, _ = wr(lat[3].to(DEV), cond, t.to(DEV), kv, ca, current_start=4680)          # a denoise step on the recomputed cache
        res[only] = (first, rc, nxt, [c["k"].clone() for c in kv], [c["v"].clone() for c in kv],
                     [(int(c["global_end_index"]), int(c["local_end_index"])) for c in kv])
    full, part = res[False], res[True]
    assert torch.equal(full[0], part[0]) and float(full[0].float().abs().max()) > 0     # ignored while the text caches fill
    assert float(part[1].float().abs().max()) == 0 and float(full[1].float().abs().max()) > 0
    assert torch.equal(full[2], part[2])                                                 # the next step sees the same cache
    for a, b in zip(full[3] + full[4], part[3] + part[4]):
        assert torch.equal(a, b)
    assert full[5] == part[5]


def test_hip_graph_replay_equals_eager():
    """with use_hip_graphs the recompute pass and the denoise step of the steady state are captured into
    hipGraphs (second sighting) and replayed afterwards; four blocks must equal the eager session bit for bit."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    g = torch.Generator().manual_seed(5)
    padded = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16)
    padded[0, :64] = torch.randn(64, text_dim, generator=g).to(torch.bfloat16)
    noise = torch.randn(1, 12, 16, 60, 104, generator=g).to(torch.bfloat16)
    outs = {}
    for graphs in (False, True):
        model, wr = _build(cfg, text_dim, w)
        model.use_hip_graphs = graphs
        pipe = CausalInferencePipeline(make_args(num_frame_per_block=3), DEV, generator=wr)
        models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded.to(DEV)))
        sess = GenerationSession(GenerateParams(seed=9, num_blocks=4, num_denoising_steps=4, keep_first_frame=True),
                                 models, device=DEV)
        sess.noise = noise.to(DEV)
        cpu_rnd = torch.Generator().manual_seed(9)
        sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
        outs[graphs] = [sess.generate_block().clone().cpu() for _ in range(4)]
        if graphs:
            assert sum(isinstance(v, dict) for v in model._graphs.values()) >= 2    # recompute + denoise step captured
    for a, b in zip(outs[False], outs[True]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("exchange", ["heads", "rows"])
def test_context_parallel_hip_graph_replay_equals_eager(exchange):
    """r05: the context-parallel forward - per-layer phase calls with the exchanges between them - is captured
    into hipGraphs like the single-GPU forward (causal_model.py; under RCCL the collectives are recorded by the capture as well:
    `bench.py --cp-host-probe --hipgraph` on a one-rank NCCL group, profiles/r05_cp_host_probe.txt).  Here: two shards in lockstep
    in this process (SimulatedContextParallel: the exchanges are block copies), four blocks of the session loop, eager vs graph
    replay vs the unsharded forward - all bit-identical (tile config 4: no K split, see the test below)."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.parallel import SimulatedContextParallel
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    g = torch.Generator().manual_seed(5)
    padded = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16)
    padded[0, :64] = torch.randn(64, text_dim, generator=g).to(torch.bfloat16)
    noise = torch.randn(1, 12, 16, 60, 104, generator=g).to(torch.bfloat16)
    outs = {}
    for mode in ("plain", "cp", "cp_graph"):
        model, wr = _build(cfg, text_dim, w)
        model.gemm_tile_cfg = 4
        if mode != "plain":
            model.context_parallel = SimulatedContextParallel(2, exchange)
        model.use_hip_graphs = mode == "cp_graph"
        pipe = CausalInferencePipeline(make_args(num_frame_per_block=3), DEV, generator=wr)
        models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded.to(DEV)))
        sess = GenerationSession(GenerateParams(seed=9, num_blocks=4, num_denoising_steps=4, keep_first_frame=True),
                                 models, device=DEV)
        sess.noise = noise.to(DEV)
        cpu_rnd = torch.Generator().manual_seed(9)
        sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
        outs[mode] = [sess.generate_block().clone().cpu() for _ in range(4)]
        if mode == "cp_graph":
            assert sum(isinstance(v, dict) for v in model._graphs.values()) >= 2    # recompute + denoise step captured
            assert model.cp_forwards_issued < 4 * 5                                  # the rest were replays
    for a, b, c in zip(outs["plain"], outs["cp"], outs["cp_graph"]):
        assert torch.equal(a, b) and torch.equal(b, c)


@pytest.mark.parametrize("exchange,tile_cfg", [("heads", 4), ("rows", 4), ("heads", 0)])
def test_full_width_layer_context_parallel_equals_unsharded(exchange, tile_cfg):
    """BASELINE config 4 at full width: ONE layer of the 14B architecture (d 5120, 40 heads, ffn 13824), 4680 tokens, cache of
    9360 rows - the 8-way token-sharded forward (585 rows per rank, 5 heads per rank under the head exchange) against the
    unsharded one, recompute pass and denoise step at cache offset 4680.  Size-independent property at the production
    shapes (tile edges at 585 rows, 128-row attention workgroups).  With a shape-independent K summation order (tile
    config 4: the ping-pong GEMM without split-K) the two are bit-identical; the default config splits K where a launch
    leaves CUs idle - which tiles that hits depends on the row count - so there the fp32 partial sums associate
    differently and the outputs agree to bf16 rounding (rel-L2 <= 8e-3: one-ulp flips, 2^-8 relative, carried through the layer)."""
    from realtime_video_amd.causal_model import CausalWanModel
    from realtime_video_amd.parallel import SimulatedContextParallel
    from realtime_video_amd.wan_wrapper import WanDiffusionWrapper
    cfg = dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=1)
    g = torch.Generator().manual_seed(3)
    lat = [torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV) for _ in range(2)]
    ctx = torch.zeros(512, 4096, dtype=torch.bfloat16)
    ctx[:64] = torch.randn(64, 4096, generator=g).to(torch.bfloat16)
    cond = {"prompt_embeds": [ctx.to(DEV)]}
    outs = []
    for cp in (None, SimulatedContextParallel(8, exchange)):
        model = CausalWanModel(text_dim=4096, freq_dim=256, device=DEV, **cfg).init_random_weights(seed=0)
        model.context_parallel = cp
        model.gemm_tile_cfg = tile_cfg
        wr = WanDiffusionWrapper(model, timestep_shift=5.0)
        kv, ca = _caches(cfg, 9360)
        model.block_mask = model._prepare_blockwise_causal_attn_mask(device=DEV, num_frames=3, frame_seqlen=1560,
                                                                     num_frame_per_block=3)
        f_rc, _ = wr(lat[0], cond, torch.zeros([1, 3], dtype=torch.int64, device=DEV), kv, ca, current_start=4680)
        model.block_mask = None
        f_dn, _ = wr(lat[1], cond, torch.ones([1, 3], dtype=torch.int64, device=DEV) * 700, kv, ca, current_start=4680)
        assert torch.isfinite(f_dn.float()).all() and float(f_dn.float().abs().mean()) > 0
        outs.append((f_rc.clone(), f_dn.clone(), kv[0]["k"].clone(), kv[0]["v"].clone()))
        del model, wr
    for a, b in zip(outs[0], outs[1]):
        if tile_cfg == 4:
            assert torch.equal(a, b)
        else:
            assert rel_l2(a, b) <= 8e-3


def test_native_session_matches_reference_generation_session(golden):
    """The native GenerationSession mirror (HIP DiT forward, native cache manager) against the golden minted by running the
    reference's OWN GenerationSession for 3 blocks (oracle/make_golden.py `session`; stand-in VAE / text encoder of
    oracle/standins.py on both sides): latents per block, cache indices, frame accounting (block 0 sends 6 of its 9
    frames), and the pixel frame handed to the encoder by the first-frame re-encode branch of block 2."""
    from oracle import standins
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models
    gold = golden("session_reference.pt")
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]),
                                   DEV, generator=wr, text_encoder=None, vae=None)
    enc_inputs = []

    def encoder(frames, cache, stream=False):
        enc_inputs.append(frames.float().cpu())
        return standins.standin_encoder(frames, cache, stream)

    models = Models(transformer=wr, pipeline=pipe, text_encoder=standins.StandinTextEncoder(gold["prompt"].to(DEV)),
                    vae_decoder=standins.standin_decoder, vae_encoder=encoder)
    sent = []
    sess = GenerationSession(GenerateParams(seed=9, num_blocks=3, num_denoising_steps=4, kv_cache_num_frames=3,
                                            keep_first_frame=False), models,
                             frame_callback=lambda px, ids, ev: sent.append(tuple(px.shape)), device=DEV)
    cpu_rnd = torch.Generator().manual_seed(9)          # the reference draws the latent noise, then the re-noising, from ONE generator
    noise = torch.randn([1, 9, 16, 60, 104], dtype=torch.bfloat16, generator=cpu_rnd)
    assert torch.equal(noise, gold["noise"])
    sess.noise = noise.to(DEV)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    assert torch.equal(sess.denoising_step_list.cpu(), gold["steps"])
    for b in range(3):
        px = sess.generate_block()
        assert rel_l2(sess.last_pred.cpu(), gold["blocks"][b]) <= 5e-2, b
        assert (int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"]),
                sess.current_start_frame, sess.block_idx, sess.total_frames_sent) == gold["indices"][b]
        assert tuple(px.shape) == gold["pixels_shape"][b] == sent[b]
        assert max_abs(px[0, :, :, ::40, ::52].cpu(), gold["pixel_sample"][b]) <= 0.1
    assert rel_l2(sess.all_latents.cpu(), gold["all_latents"]) <= 5e-2
    assert tuple(pipe.kv_cache1[0]["k"].shape) == gold["kv_shape"]
    assert [tuple(f.shape) for f in enc_inputs] == gold["encoder_input_shapes"]
    assert max_abs(enc_inputs[0][..., ::40, ::52], gold["encoder_inputs"][0]) <= 0.1


def test_native_webcam_session_matches_reference_generation_session(golden):
    """Streaming video-to-video + prompt transition against the golden minted by the reference's own GenerationSession in
    webcam mode (oracle/make_golden.py `webcam`: 3 blocks, strength 0.8, 11 / 12 / 14 queued frames resampled to 9 / 12 / 12,
    prompt interpolation over 2 steps requested after block 0, first-frame re-encode at block 2; stand-in VAE / text
    encoder on both sides).  Checked: latents per block, the schedule, cache indices and frame accounting, the sequence of
    encoder calls (shapes, stream flags, sampled input pixels) and the prompt embedding every block used."""
    from oracle import standins
    from oracle import wan_oracle as wo
    from oracle.make_golden import webcam_frames
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models
    gold = golden("session_webcam_reference.pt")
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]),
                                   DEV, generator=wr, text_encoder=None, vae=None)
    calls = []

    def encoder(frames, cache, stream=False):
        calls.append((tuple(frames.shape), bool(stream), frames.float()[..., ::40, ::52].cpu()))
        return standins.standin_encoder(frames, cache, stream)

    prompts = [p.to(DEV) for p in gold["prompts"]]
    text = standins.StandinTextEncoder(prompts[0], {"second prompt": prompts[1]})
    models = Models(transformer=wr, pipeline=pipe, text_encoder=text, vae_decoder=standins.standin_decoder, vae_encoder=encoder)
    sess = GenerationSession(GenerateParams(prompt="first prompt", seed=9, num_blocks=3, num_denoising_steps=4,
                                            kv_cache_num_frames=3, keep_first_frame=False, webcam_mode=True, strength=0.8),
                             models, device=DEV)
    assert torch.allclose(sess.denoising_step_list.cpu().float(), gold["steps"].float())
    cpu_rnd = torch.Generator().manual_seed(9)
    torch.randn([1, 9, 16, 60, 104], dtype=torch.bfloat16, generator=cpu_rnd)      # the session's (unused) latent noise comes first
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    for b, frames in enumerate(webcam_frames()):
        for f in frames:
            sess.push_frame(f.to(DEV))
        # the reference noises the input with torch.randn_like on the GLOBAL CPU generator (seeded 100 + b by the golden
        # script); its argument is the movedim(1, 2) view of the [1, 16, 3, h, w] encoder output, and normal_ on such a
        # non-contiguous tensor takes torch's element-wise path - so the draw is reproduced by the very same call
        def like(t, b=b):
            torch.manual_seed(100 + b)
            return torch.randn_like(torch.empty([1, 16, 3, 60, 104], dtype=torch.bfloat16).movedim(1, 2)).to(t.device)
        sess._randn_like = like
        sess.generate_block()
        assert rel_l2(sess.last_pred.cpu(), gold["blocks"][b]) <= 5e-2, b
        assert torch.equal(sess.current_prompt_embeds.cpu(), gold["prompt_used"][b])
        assert (int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"]),
                sess.current_start_frame, sess.block_idx, sess.total_frames_sent) == gold["indices"][b]
        if b == 0:
            sess.interpolate_prompt_embeds(models, "second prompt", 2)
    assert rel_l2(sess.all_latents.cpu(), gold["all_latents"]) <= 5e-2
    assert [(c[0], c[1]) for c in calls] == [(c[0], c[1]) for c in gold["encoder_calls"]]
    for ours, ref in zip(calls, gold["encoder_calls"]):
        assert max_abs(ours[2], ref[2]) <= 0.1


@pytest.mark.parametrize("variant", ["extension", "t2v_independent", "i2v_independent"])
def test_pipeline_inference_matches_reference_pipeline(golden, variant):
    """CausalInferencePipeline.inference - the drop-in boundary named by the north star (pipeline/causal_inference.py:48-277) -
    against goldens minted by the reference's own class: "extension" = 3 input frames cached at t = 0, then 2 blocks of 4 warped
    denoising steps + the clean-context forward; "t2v_independent" / "i2v_independent" = the independent-first-frame block
    structure [1, 3] without / with a 1-frame image latent.  KV cache of 32760 rows, stand-in VAE / text encoder.  Latents,
    decoded video (in [0, 1]), schedule, cache indices and sampled cache rows."""
    from oracle import standins
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    top = golden("pipeline_inference_reference.pt")
    gold = top if variant == "extension" else top[variant]
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250],
                                             warp_denoising_step=True, independent_first_frame=variant != "extension",
                                             context_noise=0),
                                   DEV, generator=wr, text_encoder=standins.StandinTextEncoder(top["prompt"].to(DEV)),
                                   vae=standins.StandinVAE())
    assert torch.allclose(pipe.denoising_step_list.float(), gold["steps"].float())
    torch.manual_seed(gold.get("seed", 77))   # the reference re-noises with torch.randn_like on the global CPU generator
    draws = []

    def like(t):
        eps = torch.randn_like(torch.empty(t.shape, dtype=torch.bfloat16))
        draws.append((tuple(eps.shape), tuple(eps.stride()), float(eps.float().sum())))
        return eps.to(t.device)
    pipe._randn_like = like
    initial = gold["initial"].to(DEV) if gold["initial"] is not None else None
    video, latents = pipe.inference(gold["noise"].to(DEV), ["a prompt"], initial_latent=initial, return_latents=True)
    assert [d[:2] for d in draws] == [d[:2] for d in gold["draws"]]      # same noise stream (checksums: thread-count dependent sums)
    assert all(abs(a[2] - b[2]) <= 1e-2 for a, b in zip(draws, gold["draws"]))
    if initial is not None:
        assert torch.equal(latents[:, :initial.shape[1]].cpu(), gold["initial"])
    assert rel_l2(latents.cpu(), gold["latents"]) <= 5e-2
    assert tuple(video.shape) == gold["video_shape"] and float(video.min()) >= 0 and float(video.max()) <= 1
    assert max_abs(video[0, :, :, ::40, ::52].cpu(), gold["video_sample"]) <= 0.1
    assert (int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"])) == gold["indices"]
    assert tuple(pipe.kv_cache1[0]["k"].shape) == gold["kv_shape"]
    _check_cache(pipe.kv_cache1, gold["cache"], tol=5e-2)


def test_native_session_start_frame_matches_reference_generation_session(golden):
    """Image-to-video start (params.start_frame -> setup_start_frame -> resume_latents, release_server.py:429-431, :578-595)
    against the golden minted by the reference's GenerationSession: the 9-frame encoder call, the resumed latents, two
    generated blocks behind them, indices and frame accounting."""
    from oracle import standins
    from oracle import wan_oracle as wo
    from oracle.make_golden import start_image
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models
    gold = golden("session_start_frame_reference.pt")
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]),
                                   DEV, generator=wr, text_encoder=None, vae=None)
    calls = []

    def encoder(frames, cache, stream=False):
        calls.append((tuple(frames.shape), bool(stream), frames.float()[..., ::40, ::52].cpu()))
        return standins.standin_encoder(frames, cache, stream)

    models = Models(transformer=wr, pipeline=pipe, text_encoder=standins.StandinTextEncoder(gold["prompt"].to(DEV)),
                    vae_decoder=standins.standin_decoder, vae_encoder=encoder)
    sess = GenerationSession(GenerateParams(seed=9, num_blocks=3, num_denoising_steps=4, kv_cache_num_frames=3,
                                            keep_first_frame=False, start_frame=start_image()), models, device=DEV)
    assert max_abs(sess.resume_latents.float().cpu(), gold["resume_latents"].float()) <= 2e-2
    cpu_rnd = torch.Generator().manual_seed(9)
    noise = torch.randn([1, 9, 16, 60, 104], dtype=torch.bfloat16, generator=cpu_rnd)
    assert torch.equal(noise, gold["noise"])
    sess.noise = noise.to(DEV)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    for b in range(2):
        sess.generate_block()
        assert rel_l2(sess.last_pred.cpu(), gold["blocks"][b]) <= 5e-2, b
        assert (int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"]),
                sess.current_start_frame, sess.block_idx, sess.total_frames_sent) == gold["indices"][b]
    assert rel_l2(sess.all_latents.cpu(), gold["all_latents"]) <= 5e-2
    assert [(c[0], c[1]) for c in calls] == [(c[0], c[1]) for c in gold["encoder_calls"]]
    assert max_abs(calls[0][2], gold["encoder_calls"][0][2]) <= 2e-2


def test_native_session_offline_v2v_matches_reference_generation_session(golden):
    """Offline video-to-video (`input_video`, release_server.py:417-428) against the golden minted by the reference's
    GenerationSession: one non-streamed encoder call over the 33-frame video, noise = latents noised to the first step's
    level from the session generator, block count bounded by the video (2), two generated blocks."""
    from oracle import standins
    from oracle import wan_oracle as wo
    from oracle.make_golden import v2v_video
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models
    gold = golden("session_v2v_reference.pt")
    cfg, text_dim, _ = _tiny()
    w = wo.make_weights(cfg, seed=0, text_dim=text_dim)
    model, wr = _build(cfg, text_dim, w)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]),
                                   DEV, generator=wr, text_encoder=None, vae=None)
    calls = []

    def encoder(frames, cache, stream=False):
        calls.append((tuple(frames.shape), bool(stream)))
        return standins.standin_encoder(frames, cache, stream)

    models = Models(transformer=wr, pipeline=pipe, text_encoder=standins.StandinTextEncoder(gold["prompt"].to(DEV)),
                    vae_decoder=standins.standin_decoder, vae_encoder=encoder)
    sess = GenerationSession(GenerateParams(seed=9, num_blocks=5, num_denoising_steps=4, kv_cache_num_frames=3,
                                            keep_first_frame=True, strength=0.6), models, device=DEV)
    assert torch.allclose(sess.denoising_step_list.cpu().float(), gold["steps"].float())
    cpu_rnd = torch.Generator().manual_seed(9)
    torch.randn([1, 15, 16, 60, 104], dtype=torch.bfloat16, generator=cpu_rnd)       # the session's initial noise comes first
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    sess.setup_input_video(v2v_video(), models)       # what GenerateParams(input_frames=...) runs inside __init__
    assert sess.num_blocks == gold["num_blocks"] == 2 and calls == gold["encoder_calls"]
    # same draws, same formula; the stand-in encoder's fp32 sums differ in the last bit between CPU and GPU -> single bf16 ulps
    assert rel_l2(sess.noise.cpu(), gold["noise"]) <= 2e-3 and max_abs(sess.noise.float().cpu(), gold["noise"].float()) <= 0.07
    for b in range(2):
        sess.generate_block()
        assert rel_l2(sess.last_pred.cpu(), gold["blocks"][b]) <= 5e-2, b
        assert (int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"]),
                sess.current_start_frame, sess.block_idx, sess.total_frames_sent) == gold["indices"][b]
    assert sess.generate_block_internal(models) is None
