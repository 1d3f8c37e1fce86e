"""Full-DEPTH parity on the MI355X at BASELINE width: the whole layer stack of the 14B architecture (40 layers, d 5120,
wan/configs/wan_t2v_14B.py:21-25) and of the 1.3B architecture (30 layers, d 1536, wan_t2v_1_3B.py:21-25) driven through
whole blocks of the session loop (KV-recompute forward + 4 denoise forwards, release_server.py:588-736) against the oracle
graph (oracle/wan_oracle.SessionOracle; its host evaluation is pinned to the reference's goldens by the CPU suite) evaluated
by torch eager on the device over THE SAME weight tensors, in bf16 (the reference's arithmetic) and in fp32 (gold).

Stated tolerance: end-of-block latents rel-L2(ours, bf16 oracle) <= 2e-2, the cache indices exact,
err(ours, fp32 gold) <= 2 x err(bf16 oracle, fp32 gold) (+ a small floor), K / V cache rows of the first, a middle and the last
layer rel-L2 <= 2e-2.  `scripts/depth_error_curve.py` writes the error-vs-depth curve of the same setup to profiles/."""
import pytest
import torch

from conftest import max_abs, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"

ARCH = {
    "14b": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40),
    "1.3b": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30),
}


def native_model(arch, text_dim=4096, seed=0, num_layers=None, share=None, **model_kw):
    """The native model with synthetic weights generated on the device (`init_random_weights`), or - `share` - a model of
    fewer layers over the SAME tensors as `share` (load_state_dict keeps device bf16 tensors as they are).  `model_kw`:
    further constructor arguments of CausalWanModel (local_attn_size, sink_size)."""
    from realtime_video_amd.causal_model import CausalWanModel
    from realtime_video_amd.wan_wrapper import WanDiffusionWrapper
    a = dict(ARCH[arch])
    if num_layers is not None:
        a["num_layers"] = num_layers
    m = CausalWanModel(dim=a["dim"], ffn_dim=a["ffn_dim"], num_heads=a["num_heads"], num_layers=a["num_layers"],
                       text_dim=text_dim, freq_dim=256, device=DEV, **model_kw)
    if share is None:
        m.init_random_weights(seed=seed)
    else:
        m.load_state_dict(reference_state_dict(share))
    cfg = dict(a, freq_dim=256, text_len=512, eps=1e-6, num_frame_per_block=3, **model_kw)
    return m, WanDiffusionWrapper(m, timestep_shift=5.0), cfg


def reference_state_dict(model):
    """The native model's weights under the reference's state_dict names (wan/modules/causal_model.py module tree), as VIEWS of
    the tensors the kernels read (q / k / v = row blocks of the fused to_qkv): the oracle graph and the native forward share one
    copy of the 28.6 GB."""
    t, d = model._tensors, model.dim
    sd = {"patch_embedding.weight": t["patch_w"].view(d, model.in_dim, 1, 2, 2), "patch_embedding.bias": t["patch_b"],
          "head.modulation": t["head_modulation"].view(1, 2, d)}
    for dst, src in (("text0", "text_embedding.0"), ("text2", "text_embedding.2"), ("time0", "time_embedding.0"),
                     ("time2", "time_embedding.2"), ("tproj", "time_projection.1"), ("head", "head.head")):
        sd[src + ".weight"], sd[src + ".bias"] = t[dst + "_w"], t[dst + "_b"]
    for i in range(model.num_layers):
        p, g = f"blocks.{i}", lambda n, i=i: t[f"L{i}.{n}"]
        for j, m in enumerate(("q", "k", "v")):
            sd[f"{p}.self_attn.{m}.weight"] = g("qkv_w")[j * d:(j + 1) * d]
            sd[f"{p}.self_attn.{m}.bias"] = g("qkv_b")[j * d:(j + 1) * d]
        sd[f"{p}.self_attn.o.weight"], sd[f"{p}.self_attn.o.bias"] = g("o_w"), g("o_b")
        sd[f"{p}.self_attn.norm_q.weight"], sd[f"{p}.self_attn.norm_k.weight"] = g("norm_q_w"), g("norm_k_w")
        for m in ("q", "k", "v", "o"):
            sd[f"{p}.cross_attn.{m}.weight"], sd[f"{p}.cross_attn.{m}.bias"] = g(f"c{m}_w"), g(f"c{m}_b")
        sd[f"{p}.cross_attn.norm_q.weight"], sd[f"{p}.cross_attn.norm_k.weight"] = g("cnorm_q_w"), g("cnorm_k_w")
        sd[f"{p}.norm3.weight"], sd[f"{p}.norm3.bias"] = g("norm3_w"), g("norm3_b")
        sd[f"{p}.ffn.0.weight"], sd[f"{p}.ffn.0.bias"] = g("ffn0_w"), g("ffn0_b")
        sd[f"{p}.ffn.2.weight"], sd[f"{p}.ffn.2.bias"] = g("ffn2_w"), g("ffn2_b")
        sd[f"{p}.modulation"] = t["modulation"][i].view(1, 6, d)
    return sd


def fp32_attention(q, k, v):
    """The gold graph's attention: softmax(q k^T / sqrt(d)) v in fp32, one head group at a time (the full fp32 score tensor of
    40 heads x 14040^2 does not have to exist at once)."""
    from oracle import wan_oracle as wo
    H = q.shape[2]
    step = max(1, min(H, int(2e9 // max(1, q.shape[1] * k.shape[1]))))
    return torch.cat([wo.attention_math(q[:, :, h:h + step], k[:, :, h:h + step], v[:, :, h:h + step])
                      for h in range(0, H, step)], dim=2)


def native_session(model, wr, text_dim, ctx, noise, blocks, seed, c=3):
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]),
                                   DEV, generator=wr, text_encoder=None, vae=None)
    padded = torch.zeros(1, 512, text_dim, dtype=torch.bfloat16, device=DEV)
    padded[0, :ctx.shape[0]] = ctx
    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(padded))
    sess = GenerationSession(GenerateParams(seed=seed, num_blocks=blocks, num_denoising_steps=4, keep_first_frame=True,
                                            kv_cache_num_frames=c), models, device=DEV)
    sess.noise = noise
    cpu_rnd = torch.Generator().manual_seed(seed)
    sess._randn = lambda shape: torch.randn(*shape, generator=cpu_rnd, dtype=torch.bfloat16).to(DEV)
    return sess, pipe


def run_depth_case(arch, blocks=2, text_dim=4096, gold=True, seed=9):
    """-> dict of per-block errors; used by the tests below and by scripts/depth_error_curve.py."""
    from oracle import wan_oracle as wo
    model, wr, cfg = native_model(arch, text_dim=text_dim, seed=0)
    sd = reference_state_dict(model)
    g = torch.Generator().manual_seed(5)
    ctx = torch.randn(64, text_dim, generator=g).to(torch.bfloat16).to(DEV)
    noise = torch.randn(1, 3 * blocks, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV)
    with torch.inference_mode():
        ora = wo.SessionOracle(sd, cfg, [ctx], noise, kv_cache_num_frames=3, num_steps=4, shift=5.0, seed=seed)
        ref = [ora.generate_block().clone() for _ in range(blocks)]
        ref_kv = {l: (ora.kv_cache[l]["k"][0, ::197].clone(), ora.kv_cache[l]["v"][0, ::197].clone())
                  for l in (0, cfg["num_layers"] // 2, cfg["num_layers"] - 1)}
        ref_idx = [(c["global_end_index"], c["local_end_index"]) for c in ora.kv_cache]
        del ora
        gold_blocks = None
        if gold:
            sd32 = {k: v.float() for k, v in sd.items()}
            og = wo.SessionOracle(sd32, cfg, [ctx.float()], noise.float(), kv_cache_num_frames=3, num_steps=4, shift=5.0,
                                  seed=seed, attn_fn=fp32_attention)
            gold_blocks = [og.generate_block().clone() for _ in range(blocks)]
            del og, sd32
            torch.cuda.empty_cache()
    sess, pipe = native_session(model, wr, text_dim, ctx, noise, blocks, seed)
    ours = [sess.generate_block().clone() for _ in range(blocks)]
    res = {"cfg": cfg, "blocks": []}
    for b in range(blocks):
        e = {"rel_l2_vs_oracle": rel_l2(ours[b], ref[b]), "max_abs_vs_oracle": max_abs(ours[b], ref[b])}
        if gold_blocks is not None:
            e.update(rel_l2_vs_gold=rel_l2(ours[b], gold_blocks[b]), oracle_rel_l2_vs_gold=rel_l2(ref[b], gold_blocks[b]),
                     max_abs_vs_gold=max_abs(ours[b], gold_blocks[b]), oracle_max_abs_vs_gold=max_abs(ref[b], gold_blocks[b]))
        res["blocks"].append(e)
    res["indices"] = [(int(c["global_end_index"]), int(c["local_end_index"])) for c in pipe.kv_cache1]
    res["ref_indices"] = ref_idx
    res["kv"] = {l: (rel_l2(pipe.kv_cache1[l]["k"][0, ::197], rk), rel_l2(pipe.kv_cache1[l]["v"][0, ::197], rv))
                 for l, (rk, rv) in ref_kv.items()}
    res["finite"] = all(bool(torch.isfinite(o.float()).all()) for o in ours)
    return res


@pytest.mark.timeout(600)
@pytest.mark.parametrize("arch", ["1.3b", "14b"])
def test_full_depth_two_blocks_match_oracle_and_fp32_gold(arch):
    """Block 0 (4 denoise forwards over a growing 4680-row window) and block 1 (KV-recompute over the 3 context frames with the
    block-causal mask + 4 denoise forwards over 9360 rows) of the session loop on the FULL layer stack at production width."""
    r = run_depth_case(arch, blocks=2, gold=True)
    assert r["finite"]
    assert r["indices"] == r["ref_indices"] and r["indices"][0] == (9360, 9360)      # bookkeeping exact, every layer
    for b, e in enumerate(r["blocks"]):
        assert e["rel_l2_vs_oracle"] <= 2e-2, (arch, b, e)
        assert e["rel_l2_vs_gold"] <= 2 * e["oracle_rel_l2_vs_gold"] + 2e-3, (arch, b, e)
        assert e["max_abs_vs_gold"] <= 2 * e["oracle_max_abs_vs_gold"] + 1e-2, (arch, b, e)
    for l, (ek, ev) in r["kv"].items():
        assert ek <= 2e-2 and ev <= 2e-2, (arch, l, ek, ev)


@pytest.mark.timeout(600)
def test_full_width_layer_long_context_c9_matches_oracle():
    """BASELINE config 5's context length at production WIDTH: kv_cache_num_frames = 9 (18720-row window; the recompute pass
    of block 3 covers nine context frames = 14040 tokens under the block-causal mask of three 3-frame blocks) on one layer of
    the 14B architecture, four blocks, against the bf16 oracle graph on the device."""
    from oracle import wan_oracle as wo
    model, wr, cfg = native_model("14b", text_dim=256, seed=3, num_layers=1)
    sd = reference_state_dict(model)
    g = torch.Generator().manual_seed(6)
    ctx = torch.randn(40, 256, generator=g).to(torch.bfloat16).to(DEV)
    noise = torch.randn(1, 12, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV)
    with torch.inference_mode():
        ora = wo.SessionOracle(sd, cfg, [ctx], noise, kv_cache_num_frames=9, num_steps=4, shift=5.0, seed=4)
        ref = [ora.generate_block().clone() for _ in range(4)]
    sess, pipe = native_session(model, wr, 256, ctx, noise, 4, seed=4, c=9)
    for b in range(4):
        out = sess.generate_block()
        assert rel_l2(out, ref[b]) <= 2e-2, b
    assert pipe.kv_cache1[0]["k"].shape[1] == 12 * 1560
    assert (int(pipe.kv_cache1[0]["global_end_index"]), int(pipe.kv_cache1[0]["local_end_index"])) == \
        (ora.kv_cache[0]["global_end_index"], ora.kv_cache[0]["local_end_index"]) == (18720, 18720)
    assert rel_l2(pipe.kv_cache1[0]["k"][0, :18720:97], ora.kv_cache[0]["k"][0, :18720:97]) <= 2e-2


# ------------------------------------------------------------------------------------ r05: the corners VERDICT r04 named
@pytest.mark.timeout(600)
def test_full_width_layer_rolling_cache_wrapped_ring_matches_oracle():
    """The rolling / sink cache branch (wan/modules/causal_model.py:359-385) at production WIDTH: local_attn_size = 6,
    sink_size = 1 on one layer of the 14B architecture (40 heads), four 3-frame blocks.  Blocks 2 and 3 evict 4680 rows each:
    the reference clones and shifts the cache down, the native cache advances its ring (ring of 5 x 1560 rows behind the
    1560 sink rows: shift 4680, then 1560), so the attention window of those blocks is TWO physical row ranges read by the
    four-phase kernel at 40 heads.  Against the bf16 oracle graph on the device: flow of every block rel-L2 <= 2e-2, cache
    indices exact, the cache rows in the reference's logical order rel-L2 <= 2e-2."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.causal_model import cache_row_map
    model, wr, cfg = native_model("14b", text_dim=256, seed=4, num_layers=1, local_attn_size=6, sink_size=1)
    sd = reference_state_dict(model)
    g = torch.Generator().manual_seed(8)
    ctx = torch.randn(40, 256, generator=g).to(torch.bfloat16).to(DEV)
    lat = [torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV) for _ in range(4)]
    H, rows = cfg["num_heads"], 6 * 1560
    kvc = wo.initialize_kv_cache(1, 1, rows, H, 128, torch.bfloat16, DEV)
    cac = wo.initialize_crossattn_cache(1, 1, H, 128, torch.bfloat16, device=DEV)
    kv = [{"k": torch.zeros(1, rows, H, 128, dtype=torch.bfloat16, device=DEV),
           "v": torch.zeros(1, rows, H, 128, dtype=torch.bfloat16, device=DEV), "global_end_index": 0, "local_end_index": 0}]
    ca = [{"k": torch.zeros(1, 512, H, 128, dtype=torch.bfloat16, device=DEV),
           "v": torch.zeros(1, 512, H, 128, dtype=torch.bfloat16, device=DEV), "is_init": False}]
    t = torch.ones([1, 3], dtype=torch.int64, device=DEV) * 500
    shifts = []
    for b in range(4):
        with torch.inference_mode():
            ref, _ = wo.wrapper_forward(sd, cfg, wo.FlowMatchScheduler(), lat[b], [ctx], t, kvc, cac, b * 4680)
        flow, _ = wr(lat[b], {"prompt_embeds": [ctx]}, t, kv, ca, current_start=b * 4680)
        assert rel_l2(flow, ref) <= 2e-2, (b, rel_l2(flow, ref))
        assert (int(kv[0]["global_end_index"]), int(kv[0]["local_end_index"])) == \
            (kvc[0]["global_end_index"], kvc[0]["local_end_index"]), b
        shifts.append(int(kv[0].get("ring_start", 0)))
    assert int(kv[0]["local_end_index"]) == rows and int(kv[0]["global_end_index"]) == 4 * 4680
    assert kv[0]["ring_size"] == 5 * 1560 and shifts[2] > 0 and shifts[3] != shifts[2]      # the ring wrapped twice, no shift copy
    order = cache_row_map(kv[0]).to(DEV)
    for name in ("k", "v"):
        ours = kv[0][name][0][order]                                                        # the reference's logical row order
        assert rel_l2(ours[::7], kvc[0][name][0, :order.numel():7]) <= 2e-2, name


@pytest.mark.timeout(900)
def test_long_context_c9_eight_layers_four_blocks_match_oracle():
    """kv_cache_num_frames = 9 (BASELINE config 5's context: 18720-row window, block 3 recomputes nine context frames under
    the three-block causal mask) at DEPTH: eight layers of the 14B architecture, four blocks of the session loop, against the
    bf16 oracle graph on the device (the one-layer case above pins the width; this one lets the error of the long window pass
    through a stack).  rel-L2 <= 2e-2 on every block's latents, indices exact on every layer, K rows of the first / last layer."""
    from oracle import wan_oracle as wo
    model, wr, cfg = native_model("14b", text_dim=256, seed=3, num_layers=8)
    sd = reference_state_dict(model)
    g = torch.Generator().manual_seed(6)
    ctx = torch.randn(40, 256, generator=g).to(torch.bfloat16).to(DEV)
    noise = torch.randn(1, 12, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV)
    with torch.inference_mode():
        ora = wo.SessionOracle(sd, cfg, [ctx], noise, kv_cache_num_frames=9, num_steps=4, shift=5.0, seed=4)
        ref = [ora.generate_block().clone() for _ in range(4)]
    sess, pipe = native_session(model, wr, 256, ctx, noise, 4, seed=4, c=9)
    for b in range(4):
        out = sess.generate_block()
        assert rel_l2(out, ref[b]) <= 2e-2, (b, rel_l2(out, ref[b]))
    for l in range(8):
        assert (int(pipe.kv_cache1[l]["global_end_index"]), int(pipe.kv_cache1[l]["local_end_index"])) == \
            (ora.kv_cache[l]["global_end_index"], ora.kv_cache[l]["local_end_index"]) == (18720, 18720)
    for l in (0, 7):
        assert rel_l2(pipe.kv_cache1[l]["k"][0, :18720:97], ora.kv_cache[l]["k"][0, :18720:97]) <= 2e-2, l


def run_fp8_depth_case(num_layers, seed=9):
    """One block (four denoise forwards over the growing 4680-row window, scheduler steps in between) of the session loop in
    the fp8 weight mode at 14B width: native `enable_fp8()` vs the fp8 oracle (oracle/wan_oracle.fp8_linear: the restated
    torchao arithmetic, evaluated on the device) and both against the bf16 oracle.  -> dict of rel-L2 / max-abs figures."""
    from oracle import wan_oracle as wo
    model, wr, cfg = native_model("14b", text_dim=4096, seed=0, num_layers=num_layers)
    sd = reference_state_dict(model)
    g = torch.Generator().manual_seed(5)
    ctx = torch.randn(64, 4096, generator=g).to(torch.bfloat16).to(DEV)
    noise = torch.randn(1, 3, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV)
    with torch.inference_mode():
        sd8 = dict(sd)
        sd8[wo.FP8_FLAG] = True
        o8 = wo.SessionOracle(sd8, cfg, [ctx], noise, kv_cache_num_frames=3, num_steps=4, shift=5.0, seed=seed)
        ref8 = o8.generate_block().clone()
        ref_k = o8.kv_cache[num_layers - 1]["k"][0, :4680:97].clone()
        del o8
        obf = wo.SessionOracle(sd, cfg, [ctx], noise, kv_cache_num_frames=3, num_steps=4, shift=5.0, seed=seed)
        bf = obf.generate_block().clone()
        bf_k = obf.kv_cache[num_layers - 1]["k"][0, :4680:97].clone()
        del obf
    torch.cuda.empty_cache()
    model.enable_fp8()
    sess, pipe = native_session(model, wr, 4096, ctx, noise, 1, seed)
    ours = sess.generate_block().clone()
    return {"layers": num_layers, "rel_l2_vs_fp8_oracle": rel_l2(ours, ref8), "max_abs_vs_fp8_oracle": max_abs(ours, ref8),
            "rel_l2_vs_bf16_oracle": rel_l2(ours, bf), "fp8_oracle_rel_l2_vs_bf16_oracle": rel_l2(ref8, bf),
            "k_last_layer_rel_l2": rel_l2(pipe.kv_cache1[num_layers - 1]["k"][0, :4680:97], ref_k),
            "k_last_layer_fp8_oracle_vs_bf16_oracle": rel_l2(ref_k, bf_k),
            "finite": bool(torch.isfinite(ours.float()).all())}


@pytest.mark.timeout(900)
@pytest.mark.parametrize("num_layers", [1, 40])
def test_fp8_weight_path_at_14b_width_matches_fp8_oracle(num_layers):
    """BASELINE config 5's weight path (release_server.py:179-182) at production WIDTH and DEPTH: dynamic per-tensor activation
    scales over 5120- / 13824-wide rows, one layer and the whole 40-layer stack, one block of the session loop (four chained
    denoise forwards).  What can be asked of two implementations of a DYNAMIC per-tensor scale: the scale is max|x| / 448, so
    wherever upstream bf16 rounding moves that one maximum by an ulp (2^-8), every element of the tensor is divided by another
    number and ~1 in 20 of its e4m3 codes (3 mantissa bits) re-rounds - a relative perturbation of ~2e-2 on that linear's
    output, i.e. of the size of the fp8 quantisation noise itself (measured, profiles/r05_fp8_depth_error_14b.txt: ours vs
    the fp8 oracle 3.5e-2 / 4.0e-2 / 4.9e-2 at 1 / 8 / 40 layers where fp8 vs bf16 is 5.6e-2 / 6.6e-2 / 8.4e-2; the kernel
    itself is pinned on identical quantised bytes, test_kernels_gpu.py: GEMM rel-L2 <= 4e-3, quantisation bit-exact).
    Stated tolerance therefore: (a) ours is as far from the bf16 graph as the oracle's fp8 is, within 10 %; (b) the two fp8
    implementations differ by less than 0.8 x that quantisation noise and by <= 6e-2 absolute; (c) the same for the last
    layer's cached K rows (<= 1.0 x, <= 9e-2)."""
    r = run_fp8_depth_case(num_layers)
    noise = r["fp8_oracle_rel_l2_vs_bf16_oracle"]
    assert r["finite"], r
    assert abs(r["rel_l2_vs_bf16_oracle"] - noise) <= 0.1 * noise, r
    assert r["rel_l2_vs_fp8_oracle"] <= 0.8 * noise and r["rel_l2_vs_fp8_oracle"] <= 6e-2, r
    assert r["k_last_layer_rel_l2"] <= 1.0 * r["k_last_layer_fp8_oracle_vs_bf16_oracle"] and r["k_last_layer_rel_l2"] <= 9e-2, r


# ------------------------------------------------------------------------------------ r06: the configurations VERDICT r05 named
@pytest.mark.timeout(900)
def test_full_width_layer_pipeline_inference_fills_the_32760_row_cache():
    """north_star's "~25 GB KV cache" at production WIDTH through the drop-in boundary itself: `CausalInferencePipeline.inference`
    (pipeline/causal_inference.py:48-277) on one layer of the 14B architecture (40 heads), `local_attn_size = -1`, the 32760-row
    cache `_initialize_kv_cache` allocates (:284-289), 21 latent frames = 7 blocks of 3: every block runs four denoise forwards
    and one clean-context forward over a window that grows by 4680 rows per block, and the last block attends ALL 32760 rows =
    `max_attention_size` (wan/modules/causal_model.py:192, :388-389) with the one-wave-per-SIMD kernel on the arena's strided
    rows.  Against `oracle.wan_oracle.pipeline_inference` (pinned to the reference pipeline's golden by the CPU suite) evaluated
    in bf16 by torch eager on the device over the same weights and the same re-noising draws.  Stated tolerance: latents of every
    block rel-L2 <= 2e-2 (the blocks are autoregressive: block b's context is the output of blocks < b), cache indices exact
    (32760, 32760), sampled K / V rows of the whole cache rel-L2 <= 2e-2."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    model, wr, cfg = native_model("14b", text_dim=256, seed=5, num_layers=1)
    sd = reference_state_dict(model)
    g = torch.Generator().manual_seed(11)
    ctx = torch.randn(40, 256, generator=g).to(torch.bfloat16).to(DEV)
    noise = torch.randn(1, 21, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV)

    def draws(seed):
        rnd = torch.Generator().manual_seed(seed)
        return lambda t: torch.randn(t.shape, generator=rnd, dtype=torch.bfloat16).to(t.device)

    with torch.inference_mode():
        ref, ref_kv = wo.pipeline_inference(sd, cfg, [ctx], noise, kv_size=32760, randn_like=draws(3))

    class Text:
        def __call__(self, text_prompts):
            return {"prompt_embeds": [ctx]}

    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250], context_noise=0),
                                   DEV, generator=wr, text_encoder=Text(), vae=None)
    pipe._randn_like = draws(3)
    latents = pipe.inference(noise, ["a prompt"], return_latents=True)[1]
    c = pipe.kv_cache1[0]
    assert tuple(c["k"].shape) == (1, 32760, 40, 128)
    assert (int(c["global_end_index"]), int(c["local_end_index"])) == \
        (ref_kv[0]["global_end_index"], ref_kv[0]["local_end_index"]) == (32760, 32760)
    errs = [rel_l2(latents[:, 3 * b:3 * b + 3], ref[:, 3 * b:3 * b + 3]) for b in range(7)]
    print("32760-row pipeline.inference, per-block rel-L2 vs the bf16 oracle:", " ".join(f"{e:.2e}" for e in errs))
    assert max(errs) <= 2e-2, errs
    for name in ("k", "v"):
        assert rel_l2(c[name][0, ::61], ref_kv[0][name][0, ::61]) <= 2e-2, name


def run_config5_case(num_layers=2, blocks=64, world=8, exchange="heads", seed=4):
    """BASELINE.json configs[4] as written - long-form: `num_blocks = 64`, `kv_cache_num_frames = 9`, the fp8 weight path, 8-way
    context parallelism - at 14B width on `num_layers` layers: the native session with `enable_fp8()` under
    `SimulatedContextParallel(8)` (all eight token shards in lockstep on this GPU: 585 rows per rank in the denoise forwards,
    585 / 1170 / 1755 in the recompute forwards, 5 heads per rank under the head exchange) against the session oracle in the
    row-sharded fp8 restatement (FP8_ROW_SHARDS: every rank quantises the rows it holds with its own dynamic scale), with the
    bf16 session oracle beside it as the scale of the quantisation noise.  -> per-block rel-L2 figures."""
    from oracle import wan_oracle as wo
    from realtime_video_amd.parallel import SimulatedContextParallel
    model, wr, cfg = native_model("14b", text_dim=256, seed=7, num_layers=num_layers)
    sd = reference_state_dict(model)
    sd8 = dict(sd)
    sd8[wo.FP8_FLAG] = True
    sd8[wo.FP8_ROW_SHARDS] = (world, (4680, 9360, 14040))
    g = torch.Generator().manual_seed(13)
    ctx = torch.randn(48, 256, generator=g).to(torch.bfloat16).to(DEV)
    noise = torch.randn(1, 3 * blocks, 16, 60, 104, generator=g).to(torch.bfloat16).to(DEV)
    oras = [wo.SessionOracle(w, cfg, [ctx], noise, kv_cache_num_frames=9, num_steps=4, shift=5.0, seed=seed) for w in (sd8, sd)]
    model.context_parallel = SimulatedContextParallel(world, exchange)
    model.enable_fp8()
    sess, pipe = native_session(model, wr, 256, ctx, noise, blocks, seed, c=9)
    rows = []
    with torch.inference_mode():
        for b in range(blocks):
            out = sess.generate_block()
            ref8, ref16 = (o.generate_block() for o in oras)
            rows.append({"block": b, "ours_vs_fp8_oracle": rel_l2(out, ref8), "fp8_oracle_vs_bf16_oracle": rel_l2(ref8, ref16),
                         "ours_vs_bf16_oracle": rel_l2(out, ref16), "finite": bool(torch.isfinite(out.float()).all())})
    idx = [(int(c["global_end_index"]), int(c["local_end_index"])) for c in pipe.kv_cache1]
    ref_idx = [(c["global_end_index"], c["local_end_index"]) for c in oras[0].kv_cache]
    return {"rows": rows, "indices": idx, "ref_indices": ref_idx, "frames": (sess.current_start_frame, oras[0].current_start_frame),
            "kv_rows": pipe.kv_cache1[0]["k"].shape[1]}


@pytest.mark.timeout(1100)
def test_config5_literal_64_blocks_c9_fp8_eight_way_context_parallel():
    """BASELINE.json configs[4] literally (release_server.py:179-182 fp8, :563-576 / :588-633 the sliding recompute context,
    `num_blocks = 64`, `kv_cache_num_frames = 9`, context parallel 8): two 14B-width layers, 64 blocks = 192 latent frames = 320
    forwards per side.  From block 4 on the window slides (first frame + the last eight).  Stated tolerance (the fp8 rule of
    `test_fp8_weight_path_at_14b_width_matches_fp8_oracle`, per block): ours is as far from the bf16 graph as the oracle's fp8 is
    (within 15 %); the two fp8 implementations are closer to each other than 0.9 x that quantisation noise and than 6e-2 (the
    unsharded rule says 0.8 x: here every linear has EIGHT rank-local dynamic scales instead of one, each of which re-rounds its
    585 rows' e4m3 codes when upstream bf16 rounding moves that shard's maximum by an ulp - measured 0.75-0.83 x, flat:
    3.2e-2 .. 3.5e-2 against 4.26e-2 of quantisation noise at blocks 0 / 8 / .. / 63, profiles/r06_config5_literal.log); and the
    error does NOT GROW over the 64 blocks: the mean over the last 16 blocks is <= 1.25 x the mean over blocks 4..19 (the first
    blocks with a full window).  Cache bookkeeping exact on every layer."""
    r = run_config5_case()
    rows = r["rows"]
    for x in rows[::8] + rows[-1:]:
        print("config 5 block %(block)2d: ours vs fp8 oracle %(ours_vs_fp8_oracle).3e   fp8 oracle vs bf16 oracle "
              "%(fp8_oracle_vs_bf16_oracle).3e   ours vs bf16 oracle %(ours_vs_bf16_oracle).3e" % x)
    assert all(x["finite"] for x in rows)
    assert r["indices"] == r["ref_indices"] and r["indices"][0] == (18720, 18720) and r["kv_rows"] == 18720
    assert r["frames"] == (192, 192)
    for x in rows:
        noise = x["fp8_oracle_vs_bf16_oracle"]
        assert abs(x["ours_vs_bf16_oracle"] - noise) <= 0.15 * noise, x
        assert x["ours_vs_fp8_oracle"] <= 0.9 * noise and x["ours_vs_fp8_oracle"] <= 6e-2, x
    early = sum(x["ours_vs_fp8_oracle"] for x in rows[4:20]) / 16
    late = sum(x["ours_vs_fp8_oracle"] for x in rows[-16:]) / 16
    assert late <= 1.25 * early, (early, late)
