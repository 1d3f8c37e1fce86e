"""CPU-only tests: the C-ABI library builds/loads and exports every symbol of include/rtv_hip.h (no compute),
and the host-side mirrors (scheduler, cache bookkeeping, weight packing, plugin argument checks) behave like
the reference."""
import ctypes
import os

import pytest
import torch

from realtime_video_amd import _lib


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = _lib.declared_symbols()
    assert len(syms) >= 20 and "rtv_attn_fwd" in syms and "rtv_dit_forward" in syms and "rtv_vae_decode" in syms
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.rtv_version() >= 100
    # the drop-in boundary (include/rtv_hip.h) carries no variant switches: those live in include/rtv_hip_lab.h, and the product
    # library holds no experimental kernels
    boundary = _lib.declared_symbols(lab=False)
    assert not [s_ for s_ in boundary if "_set_" in s_ and s_ not in ("rtv_gemm_set_workspace", "rtv_gemm_set_stream_workspace",
                                                                      "rtv_prof_set_stride")]   # resources / measurement, not variants
    assert "rtv_attn_set_waves" in syms and "rtv_attn_set_waves" not in boundary
    if not os.environ.get("RTV_LIB_PATH"):
        assert lib.rtv_lab_build() == 0
    lib.rtv_dit_workspace_bytes.restype = ctypes.c_size_t
    lib.rtv_vae_arena_bytes.restype = ctypes.c_size_t
    lib.rtv_vae_arena_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    # sized for 288 GB HBM: 14.6 GB per decode stream since r06 (concat buffers hold three frames' worth of new slices so that the
    # [cache | new] window slides instead of being copied back per frame; 7.4 GB before)
    assert 5e9 < lib.rtv_vae_arena_bytes(60, 104) < 16e9
    assert "rtv_vae_encode" in syms and "rtv_vae_enc_cache_slot" in syms
    lib.rtv_vae_enc_arena_bytes.restype = ctypes.c_size_t
    lib.rtv_vae_enc_arena_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    assert 3e9 < lib.rtv_vae_enc_arena_bytes(480, 832) < 8e9 and lib.rtv_vae_enc_arena_bytes(481, 832) == 0


def test_product_path_refuses_cpu_tensors():
    from realtime_video_amd import ops
    from realtime_video_amd.attention import attention
    q = torch.zeros(1, 8, 2, 128, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        attention(q, q, q)
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16))
    with pytest.raises(NotImplementedError):
        attention(q, q, q, causal=True)


def test_scheduler_mirror_matches_reference_golden(golden):
    from realtime_video_amd.scheduler import FlowMatchScheduler, get_denoising_schedule
    g = golden("ops.pt")
    s = FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    s.set_timesteps(1000, training=True)
    assert torch.equal(s.timesteps, g["sched_timesteps"]) and torch.equal(s.sigmas, g["sched_sigmas"])
    zp = torch.cat((s.timesteps, torch.tensor([0], dtype=torch.float32)))
    assert torch.equal(get_denoising_schedule(zp, 1.0, 4), g["schedule_4"])
    assert torch.equal(get_denoising_schedule(zp, 0.7, 4), g["schedule_4_s07"])
    assert torch.equal(s.add_noise(g["an_x0"], g["an_noise"], g["an_t"]), g["an_out"])


def test_wrapper_x0_conversion_matches_reference_golden(golden):
    from realtime_video_amd.wan_wrapper import WanDiffusionWrapper
    g = golden("ops.pt")
    wr = WanDiffusionWrapper.__new__(WanDiffusionWrapper)
    from realtime_video_amd.scheduler import FlowMatchScheduler
    wr.scheduler = FlowMatchScheduler(shift=5.0, sigma_min=0.0, extra_one_step=True)
    wr.scheduler.set_timesteps(1000, training=True)   # utils/wan_wrapper.py:148-151
    assert torch.equal(wr._convert_flow_pred_to_x0(g["an_x0"], g["an_noise"], g["an_t"].float()), g["x0_out"])


def _fake_cache(L, kv_size):
    return [{"k": torch.empty(1, kv_size, 2, 128, dtype=torch.bfloat16), "v": torch.empty(1, kv_size, 2, 128, dtype=torch.bfloat16),
             "global_end_index": 0, "local_end_index": 0} for _ in range(L)]


def _window(m, kv, num_new, cur, fs=1560, **kw):
    """_cache_window + commit: (cache_row0, kv_lo, kv_hi, start_frame, causal_block), ring = (ring_lo, ring_size, ring_shift)."""
    row0, lo, hi, sf, cb, ring, commit = m._cache_window(kv, num_new, cur, fs, **kw)
    commit()
    return (row0, lo, hi, sf, cb), ring


def test_cache_window_bookkeeping_server_path():
    """recompute then denoise steps at c = 3."""
    from realtime_video_amd.causal_model import CausalWanModel
    m = CausalWanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, device="cpu")
    kv = _fake_cache(2, 9360)
    # block 0: denoise at current_start 0, twice (second call overwrites the same rows)
    assert _window(m, kv, 4680, 0) == ((0, 0, 4680, 0, 0), (0, 0, 0))
    assert _window(m, kv, 4680, 0) == ((0, 0, 4680, 0, 0), (0, 0, 0))
    for c in kv:
        c["global_end_index"] = c["local_end_index"] = 0
    m.block_mask = m._prepare_blockwise_causal_attn_mask("cpu", num_frames=3, frame_seqlen=1560, num_frame_per_block=3)
    assert _window(m, kv, 4680, 4680)[0] == (0, 0, 4680, 0, 4680)   # recompute ignores current_start
    with pytest.raises(RuntimeError):
        m._cache_window(kv, 9361, 0, 1560)                           # recompute context larger than the cache
    m.block_mask = None
    assert _window(m, kv, 4680, 4680)[0] == (4680, 0, 9360, 3, 0)
    assert _window(m, kv, 4680, 4680)[0] == (4680, 0, 9360, 3, 0)
    assert all(c["global_end_index"] == 9360 and c["local_end_index"] == 9360 for c in kv)
    with pytest.raises(RuntimeError):
        m._cache_window(kv, 4680, 9360, 1560)   # would run past the (c+3)-frame cache
    # the bookkeeping is committed only once the forward has been issued: a call that is not committed changes nothing
    m._cache_window(kv, 4680, 4680, 1560)
    assert all(c["global_end_index"] == 9360 and c["local_end_index"] == 9360 for c in kv)
    # the attention window is fixed at construction (causal_model.py:192), whatever init_models writes afterwards
    m2 = CausalWanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=64, local_attn_size=4, device="cpu")
    m2.blocks[0].self_attn.local_attn_size = -1
    assert m2.blocks[0].self_attn.max_attention_size == 4 * 1560


def test_cache_window_rolling_eviction():
    """causal_model.py:359-385 with local_attn_size=6, sink_size=1: indices of SURVEY Appendix B.  The eviction is a ring
    advance (no copy): the cache tensors are never touched by the bookkeeping, the live rows in logical order
    (`cache_row_map`) are what the reference's shifted cache holds."""
    from realtime_video_amd.causal_model import CausalWanModel, cache_row_map
    m = CausalWanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=64, local_attn_size=6, sink_size=1,
                       device="cpu")
    kv = _fake_cache(1, 6 * 1560)
    kv[0]["k"][:] = torch.arange(6 * 1560).view(1, -1, 1, 1).to(torch.bfloat16)
    before = kv[0]["k"].clone()
    seen, rings = [], []
    for b in range(4):
        (row0, lo, hi, sf, cb), ring = _window(m, kv, 4680, b * 4680)
        seen.append((kv[0]["global_end_index"], kv[0]["local_end_index"], row0, lo, hi, sf))
        rings.append(ring)
    assert seen == [(4680, 4680, 0, 0, 4680, 0), (9360, 9360, 4680, 0, 9360, 3),
                    (14040, 9360, 4680, 0, 9360, 6), (18720, 9360, 4680, 0, 9360, 9)]
    assert rings == [(0, 0, 0), (0, 0, 0), (1560, 7800, 4680), (1560, 7800, 1560)]
    assert torch.equal(kv[0]["k"], before)                     # zero eviction copies
    rows = cache_row_map(kv[0])
    assert rows[:1560].tolist() == list(range(1560))           # the sink frame stays in place
    # after two evictions of 4680 rows the logical rows behind the sink start 9360 rows further along the ring
    assert rows[1560].item() == 1560 + (9360 % 7800) and sorted(rows.tolist()) == list(range(9360))
    # the reference's shift copy is still available (context-parallel exchanges move contiguous row blocks)
    kv2 = _fake_cache(1, 6 * 1560)
    kv2[0]["k"][:] = torch.arange(6 * 1560).view(1, -1, 1, 1).to(torch.bfloat16)
    for b in range(3):
        _, ring = _window(m, kv2, 4680, b * 4680, ring=False)
        assert ring == (0, 0, 0)
    assert float(kv2[0]["k"][0, 0, 0, 0]) == 0.0 and float(kv2[0]["k"][0, 1560, 0, 0]) == float(torch.tensor(6240.).to(torch.bfloat16))


def test_conv_weight_packing_is_im2col_order():
    from realtime_video_amd.vae_decoder import pack_conv_weight
    w = torch.arange(4 * 3 * 27, dtype=torch.float32).view(4, 3, 3, 3, 3)
    p = pack_conv_weight(w, cin_pad=8, cout_pad=8)
    assert p.shape == (8, 27, 8)
    assert float(p[2, 5, 1]) == float(w[2, 1].flatten()[5]) and float(p[2, 5, 3]) == 0 and float(p[5].abs().sum()) == 0


def test_vae_state_dict_spec_matches_oracle_weights():
    from oracle import vae_oracle as vo
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    spec = dict(VAEDecoderWrapper.state_dict_spec())
    w = vo.make_vae_weights(0)
    assert set(spec) == set(w) and all(tuple(w[k].shape) == tuple(v) for k, v in spec.items())


def test_vae_encoder_state_dict_spec_and_cache_slots():
    from oracle import vae_oracle as vo
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    import realtime_video_amd.vae_encoder  # noqa: F401  (registers signatures)
    spec = dict(VAEEncoderWrapper.state_dict_spec())
    w = vo.make_vae_encoder_weights(1)
    assert set(spec) == set(w) and all(tuple(w[k].shape) == tuple(v) for k, v in spec.items())
    # cache-slot geometry = the shapes the reference leaves in feat_cache (golden cache_shapes), channels 3 -> 32 padded
    off, C, h, wd, ns = ctypes.c_size_t(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    shapes = []
    for i in range(24):
        _lib.call("rtv_vae_enc_cache_slot", 64, 96, i, ctypes.byref(off), ctypes.byref(C), ctypes.byref(h), ctypes.byref(wd),
                  ctypes.byref(ns))
        shapes.append((C.value, ns.value, h.value, wd.value))
        assert off.value % 2 == 0
    g = torch.load(os.path.join(os.path.dirname(__file__), "golden", "vae_encoder.pt"))
    want = [(32 if s[1] == 3 else s[1], s[2], s[3], s[4]) for s in g["cache_shapes"][-1] if s is not None]
    assert shapes == want
    with pytest.raises(RuntimeError):
        _lib.call("rtv_vae_enc_cache_slot", 64, 96, 24, ctypes.byref(off), ctypes.byref(C), ctypes.byref(h),
                  ctypes.byref(wd), ctypes.byref(ns))


def test_rope_table_matches_reference_freqs(golden):
    from realtime_video_amd.rope import rope_cos_sin_table
    g = golden("ops.pt")
    t = rope_cos_sin_table(128)
    assert t.shape == (1024, 64, 2)
    assert torch.allclose(t[[0, 1, 5, 100, 1023]].double(), g["freqs_sample"], atol=1e-7)


def test_resample_array_matches_reference_rule():
    """release_server.py:59-64: np.round(np.linspace(0, len-1, target)) index pick."""
    from realtime_video_amd.session import resample_array
    assert resample_array([1, 2, 3], 3) == [1, 2, 3]
    assert resample_array(list(range(24)), 12) == [0, 2, 4, 6, 8, 10, 13, 15, 17, 19, 21, 23]
    assert resample_array(list(range(5)), 9) == [0, 0, 1, 2, 2, 2, 3, 4, 4]


def test_text_encoder_host_logic_matches_oracle():
    """The product's bias-table bucket rule (text_encoder.relative_position_bucket) against the oracle's restatement of
    t5.py:238-265 over every distance of a 512-token window; prompt cleaning; CPU refusal."""
    import pytest
    from oracle import t5_oracle as to
    from realtime_video_amd.text_encoder import WanTextEncoder, relative_position_bucket, whitespace_clean
    rel = torch.arange(-511, 512)
    assert torch.equal(relative_position_bucket(rel), to.relative_position_bucket(rel))
    assert int(relative_position_bucket(rel).max()) == 31 and int(relative_position_bucket(rel).min()) == 0
    assert whitespace_clean("  a &amp;amp; b \n\t c ") == "a & b c"
    enc = WanTextEncoder(device="cpu", text_len=48, **to.TINY_T5)
    with pytest.raises(RuntimeError):
        enc.encode_ids(torch.zeros(1, 4, dtype=torch.long), torch.ones(1, 4, dtype=torch.long))   # weights not loaded


def test_context_parallel_exchange_choice_and_cache_heads():
    """"auto" picks the all-to-all head exchange whenever the head count divides; the KV cache of a real rank then holds
    num_heads / world heads (a single-process simulation shares one full-head cache)."""
    import pytest
    from realtime_video_amd.causal_model import CausalWanModel
    from realtime_video_amd.parallel import SimulatedContextParallel
    cp = SimulatedContextParallel(8)
    assert cp.head_exchange(40) and not cp.head_exchange(12)
    assert not SimulatedContextParallel(8, "rows").head_exchange(40)
    assert not SimulatedContextParallel(1).head_exchange(40)
    with pytest.raises(ValueError):
        SimulatedContextParallel(8, "heads").head_exchange(12)
    m = CausalWanModel(dim=1024, ffn_dim=2048, num_heads=8, num_layers=1, device="cpu")
    assert m.kv_cache_heads() == 8
    m.context_parallel = cp
    assert m.kv_cache_heads() == 8                     # simulation: all ranks in one process, one shared cache

    class OneRank(SimulatedContextParallel):
        def local_ranks(self):
            return [3]
    m.context_parallel = OneRank(8)
    assert m.kv_cache_heads() == 1
    m.context_parallel = OneRank(8, "rows")
    assert m.kv_cache_heads() == 8


def test_attention_plugin_installs_into_the_reference_modules():
    """INTEGRATION.md section 1 executed against the real reference (authoring container only; skipped where
    /root/reference does not exist): after install() the reference's own CausalWanModel reaches this backend from its
    self-attention and cross-attention call sites - shown by the backend's refusal of CPU tensors surfacing from the
    reference's forward."""
    import pytest
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    from oracle import wan_oracle as wo
    from oracle.make_golden import TEXT_DIM, TINY, fresh_caches, tiny_inputs
    from realtime_video_amd import attention as plug
    ref = ref_shim.load()
    mods = (ref.attention, ref.model, ref.cm)
    saved = [{n: getattr(m, n) for n in ("attention", "sageattn_func", "SAGEATTN_AVAILABLE") if hasattr(m, n)} for m in mods]
    plug.install(*mods)
    try:
        assert ref.attention.attention is plug.attention and ref.cm.attention is plug.attention
        assert ref.model.SAGEATTN_AVAILABLE is True and ref.model.sageattn_func is plug.sageattn_func
        cfg = dict(TINY, num_layers=1)
        w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
        model = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
        wr = ref_shim.build_reference_wrapper(ref, model)
        lat, ctx = tiny_inputs()
        kv, ca = fresh_caches(cfg, 4680)
        with pytest.raises(RuntimeError, match="GPU|cuda|CUDA"):
            with torch.inference_mode():
                wr(lat[0], {"prompt_embeds": [ctx]}, torch.ones([1, 3], dtype=torch.int64) * 500, kv, ca, current_start=0)
    finally:
        for m, s in zip(mods, saved):
            for n, v in s.items():
                setattr(m, n, v)


def _key_window_rows(lo, hi, S, R, shift):
    """Python restatement of dit_forward.hip key_window(): the <= 2 physical row ranges of the logical window [lo, hi)."""
    if R <= 0 or hi <= S:
        segs = [(lo, hi - lo)]
    else:
        rl = max(lo, S)
        n = hi - rl
        p0 = S + (rl - S + shift) % R
        first = n if p0 + n <= S + R else S + R - p0
        sink = S - lo if lo < S else 0
        if first == n:
            segs = [(p0, n)] if sink == 0 else ([(lo, sink + n)] if p0 == S else [(lo, sink), (p0, n)])
        else:
            segs = [(lo if sink else S, sink + n - first), (p0, first)]
    assert len(segs) <= 2
    rows = [r for a, n in segs for r in range(a, a + n)]
    assert len(rows) == len(set(rows))
    return set(rows)


def test_cache_window_random_sequences_match_the_reference_rule():
    """Property test of the KV bookkeeping (must match the reference EXACTLY): 150 random call sequences - window
    sizes with and without attention sinks, 1- and 3-frame calls, repeated calls on the same block (denoising steps) - run
    through CausalWanModel._cache_window on CPU cache tensors whose rows carry their write stamp, against the index
    arithmetic and the eviction copy of causal_model.py:349-392 restated inline."""
    import random
    from realtime_video_amd.causal_model import CausalWanModel, cache_row_map
    rng = random.Random(7)
    fs = 1560
    for trial in range(150):
        las = rng.choice([-1, 4, 6, 9])
        sink = rng.choice([0, 1, 2]) if las != -1 else 0
        kv_size = 32760 if las == -1 else las * fs
        m = CausalWanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=64, local_attn_size=las, sink_size=sink,
                           device="cpu")
        k = torch.zeros(1, kv_size, 1, 1)
        kv = [{"k": k, "v": k.clone(), "global_end_index": 0, "local_end_index": 0}]
        ref_k = torch.zeros(kv_size)                    # reference model of the cache: one stamp per row
        g_end = l_end = 0
        cur, stamp = 0, 0
        for call in range(rng.randint(3, 14)):
            num_new = fs * rng.choice([1, 3, 3])
            if rng.random() < 0.4 and call > 0:
                cur = prev_cur                            # another denoising step on the same block
                num_new = prev_new
            current_end = cur + num_new
            max_att = 32760 if las == -1 else las * fs
            sink_tokens = sink * fs
            if las != -1 and current_end > g_end and num_new + l_end > kv_size:        # causal_model.py:363-379
                evicted = num_new + l_end - kv_size
                rolled = l_end - evicted - sink_tokens
                ref_k[sink_tokens:sink_tokens + rolled] = ref_k[sink_tokens + evicted:sink_tokens + evicted + rolled].clone()
                local_end = l_end + current_end - g_end - evicted
            else:
                local_end = l_end + current_end - g_end                                # :380-381
            local_start = local_end - num_new
            if local_start < 0 or local_end > kv_size:
                break                                       # the reference would index out of range here; skip the tail
            use_ring = trial % 3 != 0                       # every third trial: the shift-copy mode of the CP path
            (row0, lo, hi, start_frame, cb), (r_lo, r_size, r_shift) = _window(m, kv, num_new, cur, fs, ring=use_ring)
            stamp += 1
            # what the forward's cache write does (rtv_qk_norm_rope_cache_ring): logical row -> physical row
            rr = torch.arange(row0, row0 + num_new)
            phys = torch.where(rr >= r_lo, r_lo + (rr - r_lo + r_shift) % r_size, rr) if r_size else rr
            assert use_ring or r_size == 0
            kv[0]["k"][0, phys] = stamp
            ref_k[local_start:local_end] = stamp
            g_end, l_end = current_end, local_end
            assert (row0, lo, hi, start_frame, cb) == (local_start, max(0, local_end - max_att), local_end, cur // fs, 0), trial
            assert (kv[0]["global_end_index"], kv[0]["local_end_index"]) == (g_end, l_end), trial
            # the live rows in logical order are exactly the reference's (shifted) cache rows [0, local_end)
            assert torch.equal(kv[0]["k"][0, cache_row_map(kv[0]), 0, 0], ref_k[:l_end]), trial
            # and the two physical ranges the attention kernel walks cover exactly the window [lo, hi)
            live = set(cache_row_map(kv[0])[lo:hi].tolist())
            assert _key_window_rows(lo, hi, r_lo, r_size, r_shift) == live, trial
            prev_cur, prev_new = cur, num_new
            cur = current_end


def test_cache_window_matches_the_real_reference_model_on_random_sequences():
    """The same property against the REAL reference (authoring container only): the reference's CausalWanModel (1 layer, tiny
    dims) is driven through random call sequences with local attention windows and sinks; after every call its cache
    indices must equal those of CausalWanModel._cache_window fed the same sequence, and the set of cache rows it left
    non-zero must be the window our bookkeeping reports."""
    import random
    import pytest
    from oracle import ref_shim
    if not ref_shim.available():
        pytest.skip("reference tree not present")
    from oracle import wan_oracle as wo
    from oracle.make_golden import TEXT_DIM, TINY
    from realtime_video_amd.causal_model import CausalWanModel
    ref = ref_shim.load()
    rng = random.Random(11)
    fs = 1560
    g = torch.Generator().manual_seed(2)
    ctx = torch.randn(64, TEXT_DIM, generator=g).to(torch.bfloat16)
    for trial, (las, sink) in enumerate([(4, 0), (6, 1), (6, 2), (9, 1)]):
        cfg = dict(TINY, num_layers=1, local_attn_size=las, sink_size=sink)
        w = wo.make_weights(cfg, seed=trial, text_dim=TEXT_DIM)
        rmodel = ref_shim.build_reference_model(ref, cfg, w, TEXT_DIM)
        wr = ref_shim.build_reference_wrapper(ref, rmodel)
        kv_size = las * fs
        rkv = [{"k": torch.zeros(1, kv_size, 2, 128, dtype=torch.bfloat16), "v": torch.zeros(1, kv_size, 2, 128, dtype=torch.bfloat16),
                "global_end_index": torch.tensor([0]), "local_end_index": torch.tensor([0])}]
        rca = [{"k": torch.zeros(1, 512, 2, 128, dtype=torch.bfloat16), "v": torch.zeros(1, 512, 2, 128, dtype=torch.bfloat16),
                "is_init": False}]
        ours = CausalWanModel(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=2, num_layers=1, text_dim=TEXT_DIM,
                              local_attn_size=las, sink_size=sink, device="cpu")
        okv = [{"k": torch.zeros(1, kv_size, 1, 1), "v": torch.zeros(1, kv_size, 1, 1), "global_end_index": 0, "local_end_index": 0}]
        cur = 0
        for call in range(7):
            frames = rng.choice([1, 3])
            if call and rng.random() < 0.35:
                cur, frames = prev
            lat = torch.randn(1, frames, 16, 60, 104, generator=g).to(torch.bfloat16)
            with torch.inference_mode():
                wr(lat, {"prompt_embeds": [ctx]}, torch.ones([1, frames], dtype=torch.int64) * 500, rkv, rca, current_start=cur)
            (row0, lo, hi, sf, cb), _ = _window(ours, okv, frames * fs, cur, fs)
            assert (int(rkv[0]["global_end_index"]), int(rkv[0]["local_end_index"])) == \
                (okv[0]["global_end_index"], okv[0]["local_end_index"]), (trial, call)
            written = rkv[0]["k"][0].float().abs().sum((-1, -2)) > 0
            assert bool(written[:hi].all()) and not bool(written[hi:].any()), (trial, call)     # rows [0, local_end) are live
            prev = (cur, frames)
            cur += frames * fs


# ------------------------------------------------------------------------------------------------ VAE cache arenas
def test_vae_cache_arena_registry_finds_rebuilds_and_refuses():
    """vae_decoder.CacheArenas (host logic, no kernels): a cache list is recognised by the address of its first slot, a list
    of cloned slots is copied into a new arena (contents preserved, list updated in place), a list of another frame size
    or with a mismatching slot is refused, and only the last few arenas are kept alive by the registry."""
    from realtime_video_amd.vae_decoder import CacheArenas

    def make_views(arena, base, reg, size):
        views = [arena[base + 16 * i: base + 16 * (i + 1)].view(torch.float16) for i in range(3)] + [None]
        reg.register(views, arena, base, size)
        return views

    reg = CacheArenas()
    arena = torch.zeros(64 + 256, dtype=torch.uint8)
    views = make_views(arena, 4, reg, (8, 12))
    for i, v in enumerate(views[:3]):
        v.fill_(i + 1)
    got_arena, got_base = reg.lookup(list(views), (8, 12), None, None)
    assert got_arena is arena and got_base == 4
    with pytest.raises(ValueError):
        reg.lookup(list(views), (8, 20), None, None)
    # cloned slots: a new arena, contents carried over, the caller's list now holds views of it
    made = []

    def new_arena():
        made.append(torch.zeros(64 + 256, dtype=torch.uint8))
        return made[-1]

    snap = [None if v is None else v.clone() for v in views]
    a2, b2 = reg.lookup(snap, (8, 12), new_arena, lambda a, b: make_views(a, b, reg, (8, 12)))
    assert a2 is made[0] and a2 is not arena and b2 == (-a2.data_ptr()) % 256
    assert all(torch.equal(s, v) for s, v in zip(snap[:3], views[:3])) and snap[3] is None
    assert snap[0].data_ptr() == a2.data_ptr() + b2          # a view of the new arena
    assert reg.lookup(snap, (8, 12), None, None)[0] is a2     # ... which is registered
    # a slot that should be a tensor is None (or has another shape): refused
    bad = [v.clone() for v in views[:3]] + [None]
    bad[1] = None
    with pytest.raises(ValueError):
        reg.lookup(bad, (8, 12), new_arena, lambda a, b: make_views(a, b, reg, (8, 12)))
    bad = [v.clone() for v in views[:3]] + [None]
    bad[2] = torch.zeros(5, dtype=torch.float16)
    with pytest.raises(ValueError):
        reg.lookup(bad, (8, 12), new_arena, lambda a, b: make_views(a, b, reg, (8, 12)))
    # a restored snapshot that replaced SOME slots (slot 0 still a view of the old arena): the replaced contents must not be
    # ignored -> copy-in path, not a hit on the old arena
    mixed = list(views)
    mixed[1] = torch.full_like(views[1], 7.0)
    n_made = len(made)
    a3, _ = reg.lookup(mixed, (8, 12), new_arena, lambda a, b: make_views(a, b, reg, (8, 12)))
    assert len(made) == n_made + 1 and a3 is made[-1]
    assert float(mixed[1][0]) == 7.0 and torch.equal(mixed[0], views[0]) and mixed[0].data_ptr() != views[0].data_ptr()
    # eviction: only `keep` arenas stay registered, least recently USED first (a hit refreshes an entry), with a warning
    reg = CacheArenas(keep=3)
    lists = []
    for i in range(3):
        a = torch.zeros(64 + 256, dtype=torch.uint8)
        lists.append(make_views(a, 0, reg, (8, 12)))
    reg.lookup(list(lists[0]), (8, 12), None, None)               # stream 0 is now the most recently used
    with pytest.warns(UserWarning, match="concurrent feature-cache streams"):
        lists.append(make_views(torch.zeros(64 + 256, dtype=torch.uint8), 0, reg, (8, 12)))
    assert len(reg._by_ptr) == 3
    assert lists[0][0].data_ptr() in reg._by_ptr and lists[1][0].data_ptr() not in reg._by_ptr


def test_attn_w4_audit_compiler_stays_out_of_the_accumulation_registers():
    """csrc/attn_w4.hip names all 256 accumulation registers literally in inline asm (O^T, Q^T, K / V^T fragments): correct only
    as long as hipcc never places a value of its own in one.  scripts/micro/w4_audit.sh compiles the product variant to ISA (no
    GPU needed) and reports every accumulation register named outside ASMSTART / ASMEND, scratch use and spills: all must be 0,
    and the tile loop must hold its 4 x 64 matrix instructions."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, W4_AUDIT_DIR=os.path.join(root, "realtime_video_amd", "csrc", "build", "w4_audit"))
    out = subprocess.run([os.path.join(root, "scripts", "micro", "w4_audit.sh"), "600"], capture_output=True, text=True, env=env,
                         timeout=600).stdout
    assert re.search(r"instructions naming an accumulation register outside asm: 0\b", out), out
    assert re.search(r"v_accvgpr outside asm: 0\b", out), out
    assert re.search(r"\bscratch 0\b", out) and re.search(r"vgpr_spill_count: 0\b", out) and re.search(r"sgpr_spill_count: 0\b", out), out
    assert re.search(r"private_segment_fixed_size: 0\b", out), out
    assert len(re.findall(r"barrier  mfma=64 ", out)) == 4, out


def test_gemm5_audit_compiler_stays_out_of_the_live_accumulation_registers():
    """csrc/gemm5.hip (160-row GEMM of the context-parallel token shards) keeps its accumulators and fragments in accumulation registers
    named literally in inline asm: scripts/micro/g5_audit.sh compiles it to ISA and counts compiler references to accumulation
    registers in front of the kernel's last accumulator read-out (behind it they are dead), scratch use and spills - all zero, for
    the bf16 and the f16 instantiation - and the 3 x 40 matrix instructions of the unrolled K loop."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, G5_AUDIT_DIR=os.path.join(root, "realtime_video_amd", "csrc", "build", "g5_audit"))
    out = subprocess.run([os.path.join(root, "scripts", "micro", "g5_audit.sh")], capture_output=True, text=True, env=env, timeout=600).stdout
    assert len(re.findall(r"AUDIT gemm5 F16=\d: vgpr_spills 0 private_segment 0 scratch_ops 0 mfma 120 compiler_acc_refs_before_last_accumulator_read 0 ", out)) == 2, out


def test_conv_halo4_audit_compiler_stays_out_of_the_accumulation_registers():
    """csrc/vae_conv.hip: the one-wave-per-SIMD halo-tile convolution kernels (conv_halo4_kernel, and conv_halo4p_kernel - the
    persistent form, the VAE's default) keep 12 accumulator blocks and two fragment sets in accumulation registers named literally
    in inline asm.  scripts/micro/h4_audit.sh compiles the file to ISA and counts, per kernel, compiler references to accumulation
    registers (the persistent form has a tile loop around its K loop: everything per-lane is rebuilt per tile from laundered ids so
    that hipcc has no loop-carried values to park there), scratch use and vector-register spills - all zero - and the 216 matrix
    instructions of a group of three tap rows."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, H4_AUDIT_DIR=os.path.join(root, "realtime_video_amd", "csrc", "build", "h4_audit"))
    out = subprocess.run([os.path.join(root, "scripts", "micro", "h4_audit.sh")], capture_output=True, text=True, env=env, timeout=600).stdout
    pat = (r"AUDIT _ZN3rtv1[78]conv_halo4p?_kernel\S*: arch_vgprs \d+ vgpr_spills 0 sgpr_spills \d+ private_segment 0 scratch_ops 0 "
           r"mfma 216 compiler_acc_refs 0 \(a0-a7: 0\)")
    assert len(re.findall(pat, out)) == 2, out
