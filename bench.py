"""Headline benchmark: frames/sec at 832x480, 4 denoising steps, Krea-Realtime-14B-shaped causal Wan DiT
(random-init weights of the 14B architecture, synthetic latents / prompt embeddings) + streaming VAE
decode, driven by the GenerationSession block loop — BASELINE.json `configs[2]`.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

One "step" = one generated block = KV-recompute forward + 4 denoise forwards + VAE decode of 3 latent
frames = 12 output frames.  Prints ONE JSON line on rank 0 (see README / DESIGN.md §Measurement).
"""
import argparse
import json
import os
import sys
import time

# RCCL between processes needs dmabuf IPC on this platform (the driver's environment exports it; keep it if somebody does not)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODELS = {
    "14b": dict(dim=5120, ffn_dim=13824, num_heads=40, num_layers=40, name="Krea-Realtime-14B (Wan2.1-T2V-14B arch)"),
    "1.3b": dict(dim=1536, ffn_dim=8960, num_heads=12, num_layers=30, name="Wan2.1-T2V-1.3B arch"),
    # test rig only (tests/test_context_parallel_gpu.py drives the launcher path with it): 8 heads -> one head per rank at 8 ranks
    "tiny": dict(dim=1024, ffn_dim=2048, num_heads=8, num_layers=2, name="TEST RIG (INVALID as a result): 2-layer d=1024 H=8"),
    # --cp-host-probe: the 14B's 40 layers (= its count of host operations per forward) at a width whose kernels take microseconds,
    # so that the launch queue never pushes back and the loop's wall time is the host's own cost
    "hostrig": dict(dim=256, ffn_dim=512, num_heads=2, num_layers=40, name="HOST-COST RIG (INVALID as a result): 40-layer d=256 H=2"),
}
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X (MI355X_MICROARCH.md)
HBM_PEAK_TBPS = 8.0        # HBM3E, MI355X (MI355X_MICROARCH.md)
MFMA_SUSTAINED_FP8_TFLOPS = 4350.0  # v_mfma_f32_32x32x64_f8f6f4 loop, random e4m3 operands (scripts/micro/fp8_mfma.hip)
MFMA_SUSTAINED_TFLOPS = 1770.0  # measured: MFMA-only loop, random operands, clock settles at 1.78 GHz (profiles/r01_mfma_peak.txt)


def gemm_algorithmic_bytes(mc, M=4680):
    """Average algorithmic bytes per DiT-layer projection GEMM launch: operands read once + output written once
    (A[M,K] + W[N,K] + C[M,N] in bf16, plus the residual read where the epilogue fuses it)."""
    d, f = mc["dim"], mc["ffn_dim"]
    # (N, K, fused residual): q|k, v, o, cross-q, cross-o, ffn-in, ffn-out - the seven projection launches of a layer since r05
    shapes = [(2 * d, d, 0), (d, d, 0), (d, d, 1), (d, d, 0), (d, d, 1), (f, d, 0), (d, f, 1)]
    tot = sum(2 * (M * K + N * K + M * N + res * M * N) for N, K, res in shapes)
    return tot / len(shapes)


def measured_traffic(model):
    """HBM-side bytes per launch of the dominant kernel class from the committed rocprofv3 PMC passes over the six
    projection shapes of a layer (scripts/profile_gemm_traffic.sh: FETCH_SIZE x2 + WRITE_SIZE per MI355X_MICROARCH.md, separate
    --pmc passes).  rocprofv3 --pmc cannot run inside this process, so the committed summary of the newest round is read;
    its sample size and file travel in the JSON line."""
    root = os.path.dirname(os.path.abspath(__file__))
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(root, "profiles", f"{rnd}_traffic_{model}.json")
        try:
            with open(path) as f:
                g = json.load(f)["classes"]["gemm"]
            return g["hbm_bytes_per_launch"], os.path.relpath(path, root), g.get("launches_sampled")
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def cpu_baseline(model_cfg, with_vae=True):
    """The CPU oracle (a port of the reference's eager path: oracle/wan_oracle.py, oracle/vae_oracle.py) timed on this host's
    cores on a bounded sample of the benchmarked block: ONE DiT layer of the benchmarked width at the real token counts
    (M = 4680 queries, 9360 cached keys; run twice, the second - warm - run is the sample) extrapolated to the block's
    (4 denoise + 0.88 recompute-equivalent) x L layer-forwards, plus ONE latent frame of the streaming VAE decode at
    480 x 832 on warm caches (fp32; 4 of the block's 12 pixel frames) extrapolated x3."""
    from oracle import wan_oracle as wo
    try:
        cores = len(os.sched_getaffinity(0))   # the cores this process may use (a cgroup / cpuset can be narrower than the host)
    except AttributeError:
        cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    d, ffn, H = model_cfg["dim"], model_cfg["ffn_dim"], model_cfg["num_heads"]
    g = torch.Generator().manual_seed(0)
    w = {}
    for a in ("self_attn", "cross_attn"):
        for m in ("q", "k", "v", "o"):
            w[f"blocks.0.{a}.{m}.weight"] = (torch.randn(d, d, generator=g) * d ** -0.5).to(torch.bfloat16)
            w[f"blocks.0.{a}.{m}.bias"] = torch.zeros(d, dtype=torch.bfloat16)
        w[f"blocks.0.{a}.norm_q.weight"] = torch.ones(d, dtype=torch.bfloat16)
        w[f"blocks.0.{a}.norm_k.weight"] = torch.ones(d, dtype=torch.bfloat16)
    w["blocks.0.norm3.weight"], w["blocks.0.norm3.bias"] = torch.ones(d, dtype=torch.bfloat16), torch.zeros(d, dtype=torch.bfloat16)
    w["blocks.0.ffn.0.weight"] = (torch.randn(ffn, d, generator=g) * d ** -0.5).to(torch.bfloat16)
    w["blocks.0.ffn.0.bias"] = torch.zeros(ffn, dtype=torch.bfloat16)
    w["blocks.0.ffn.2.weight"] = (torch.randn(d, ffn, generator=g) * ffn ** -0.5).to(torch.bfloat16)
    w["blocks.0.ffn.2.bias"] = torch.zeros(d, dtype=torch.bfloat16)
    w["blocks.0.modulation"] = (torch.randn(1, 6, d, generator=g) * d ** -0.5).to(torch.bfloat16)
    x = torch.randn(1, 4680, d, generator=g).to(torch.bfloat16)
    e = torch.randn(1, 3, 6, d, generator=g).to(torch.bfloat16) * 0.1
    ctx = torch.randn(1, 512, d, generator=g).to(torch.bfloat16)
    freqs = wo.rope_table(128)
    t_layer = None
    for _ in range(2):   # the second (warm: thread pool, allocator, weights paged in) iteration is the sample
        kv = wo.initialize_kv_cache(1, 1, 9360, H, 128, torch.bfloat16)[0]
        kv["k"][:, :4680].normal_(generator=g)
        kv["v"][:, :4680].normal_(generator=g)
        kv["global_end_index"] = kv["local_end_index"] = 4680
        ca = wo.initialize_crossattn_cache(1, 1, H, 128, torch.bfloat16)[0]
        t0 = time.time()
        with torch.inference_mode():
            wo.attention_block(w, "blocks.0", x, e, (3, 30, 52), freqs, ctx, H, kv, ca, 4680, False)
        t_layer = time.time() - t0
    del w, x, kv
    L = model_cfg["num_layers"]
    t_dit = t_layer * L * 4.88
    t_vae = None
    if with_vae:
        from oracle import vae_oracle as vo
        wv = vo.make_vae_weights(seed=0)
        cache = [None] * 55
        with torch.inference_mode():
            _, cache = vo.decoder_wrapper_forward(wv, torch.randn(1, 1, 16, 60, 104, generator=g), cache)   # fills the caches
            t0 = time.time()
            vo.decoder_wrapper_forward(wv, torch.randn(1, 1, 16, 60, 104, generator=g), cache)               # steady state: 4 frames
            t_vae = (time.time() - t0) * 3
    t_block = t_dit + (t_vae or 0.0)
    return {"value": 12.0 / t_block, "unit": "frames/s", "cores": cores, "kind": "port",
            "port_vs_reference": "the Python reference cannot travel to the GPU box; the same sample on the upstream modules in "
                                 "the authoring container costs what the port costs (port / reference = 1.03 DiT layer, 0.92 VAE "
                                 "frame, outputs identical: profiles/r03_cpu_baseline_reference_vs_port.txt)",
            "sample": f"DiT: 1 layer at the benchmarked width (d={d}, ffn={ffn}, H={H}), M=4680 query tokens, 9360 cached keys, "
                      f"bf16 eager oracle, second of two runs {t_layer:.2f} s, extrapolated x{L} layers x 4.88 forwards per 12-frame "
                      f"block = {t_dit:.0f} s" + (f"; VAE: 1 latent frame (4 of 12 pixel frames) of the streaming decode at 480x832 "
                                                  f"on warm caches, fp32 eager oracle, {t_vae / 3:.2f} s, x3 = {t_vae:.0f} s"
                                                  if t_vae is not None else "; VAE excluded")}


def rccl_graph_preflight(dev, world, rank):
    """Capture one all-reduce and one all-to-all of the default process group into a hipGraph on a side stream, replay it once and
    check the numbers: -> (ok, reason).  The collectives run once eagerly first (communicator set-up must not fall into a capture)."""
    import torch.distributed as dist
    try:
        x = torch.ones(4096, device=dev)
        a_in = torch.full((world * 256,), float(rank), device=dev)
        a_out = torch.empty_like(a_in)
        dist.all_reduce(x)
        dist.all_to_all_single(a_out, a_in)
        torch.cuda.synchronize()
        x.fill_(1.0)
        a_out.fill_(-1.0)
        g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream(device=dev)
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            dist.all_reduce(x)
            dist.all_to_all_single(a_out, a_in)
        g.replay()
        torch.cuda.synchronize()
        want = torch.arange(world, device=dev, dtype=torch.float32).repeat_interleave(256)
        if abs(float(x[0]) - world) > 1e-3 or not torch.equal(a_out, want):
            return False, f"wrong result after replay (all-reduce {float(x[0])}, expected {world})"
        return True, None
    except Exception as e:      # noqa: BLE001 - any failure means "do not capture collectives on this box"
        return False, repr(e)[:300]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="14b", choices=sorted(MODELS))
    ap.add_argument("--kv-cache-num-frames", type=int, default=3)
    ap.add_argument("--denoising-steps", type=int, default=4)
    ap.add_argument("--no-vae", action="store_true", help="diagnostic only: skips the VAE (result is flagged invalid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-classes", default="gemm,attn,conv,layernorm,rope",
                    help="kernel classes bracketed with hipEvents inside the timed region: 'gemm,attn,conv,layernorm,rope' (the roofline "
                         "kernel, the attention kernels, the VAE convolutions and the HBM-bound row kernels, default), 'all' "
                         "(diagnostic: every launch) or 'none'")
    ap.add_argument("--profile-stride", default="gemm=29,attn=11,layernorm=23,rope=17,conv=3",
                    help="bracket only every n-th launch of a class (an event pair costs the launch stream ~2.5 us per record: "
                         "bracketing all ~2.4 k GEMM launches of a block cost it ~12 ms, the ~940 row-kernel launches 9 ms - "
                         "profiles/r04_bench_bracket_overhead.txt); class times are the sampled times scaled by work.  r06: the r04-r05 "
                         "strides (13 / 5 / 11 / 7 / 3) still cost a block 2.4 ms; these cost 0.5 ms and read the same rates "
                         "(roofline.frac 0.5176 vs 0.5177, attention 1103-1106 vs 1101-1103 TF/s: profiles/r06_bracket_stride_ab.txt). "
                         "Strides are coprime with the launches per layer of their class (7 GEMMs, 2 attention, 3 LayerNorm), so the "
                         "sample cycles through every shape; the convolutions keep 3: few launches of very different sizes")
    ap.add_argument("--hipgraph", action="store_true",
                    help="replay every DiT forward from a captured hipGraph (SURVEY 8f-2).  Kernel launches inside a graph "
                         "cannot be bracketed with events, so this run carries no roofline block: a diagnostic of the "
                         "launch-gap cost, not the contract line")
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE config 5's weight path (reference enable_fp8: e4m3 weights + dynamic per-tensor e4m3 "
                         "activations in every nn.Linear).  NOT the headline precision: the line is flagged dtype fp8 and "
                         "carries no vs_baseline")
    ap.add_argument("--keep-first-frame", action="store_true",
                    help="GenerateParams.keep_first_frame=True: skip the per-block first-frame VAE re-encode (A/B runs)")
    ap.add_argument("--gemm-tile-cfg", type=int, default=0)
    ap.add_argument("--cp-exchange", default="auto", choices=["auto", "heads", "rows"],
                    help="context-parallel exchange around self-attention: heads = all-to-all pair (head-sharded KV cache), "
                         "rows = K/V all-gather (replicated cache); auto = heads when the head count divides")
    ap.add_argument("--cp-attn-splits", type=int, default=0,
                    help="context parallel: cut every rank's self-attention launch along the keys into this many ranges "
                         "(rtv_attn_fwd_split; 1 = one launch, bit-identical with the unsharded forward; 0 = the count that "
                         "fills the 256 CUs best for this rank count: 14B: 2 / 4 / 2 at 2 / 4 / 8 ranks)")
    ap.add_argument("--no-cp-overlap", action="store_true",
                    help="A/B: complete every context-parallel collective before the next kernel is issued (default: the "
                         "q all-to-all runs under the k|v projection / the K/V all-gather under the q projection)")
    ap.add_argument("--simulate-cp", type=int, default=0,
                    help="diagnostic (INVALID as a result): run all shards of an N-way context-parallel forward in lockstep "
                         "on this one GPU (no collectives, --no-vae implied); kernel_ms_per_block / N = one rank's compute")
    ap.add_argument("--cp-host-probe", action="store_true",
                    help="diagnostic (INVALID as a result): what the HOST side of the context-parallel forward costs per block - the "
                         "Python loop of 4 C calls + 3 collectives per layer (causal_model.py) - measured without queue back-pressure: "
                         "a ONE-rank process group (RCCL, collectives = self-exchanges through the same code path) on the 40-layer "
                         "narrow rig; prints config.cp_host_ms_per_block (VERDICT r04 item 4)")
    ap.add_argument("--parallel", default="cp", choices=["cp", "replicas"],
                    help="N>1: cp = context-parallel single stream (strong scaling, RCCL all-gather per layer); "
                         "replicas = one independent stream per GPU (weak scaling, no collective)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints its version banner on stdout when a
    # process group comes up or goes down): file descriptor 1 is pointed at stderr for the whole run and the line goes to the saved
    # descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.cp_host_probe:
        if args.gpus != 1:
            raise SystemExit("--cp-host-probe is a one-rank measurement")
        args.model, args.no_vae, args.no_cpu_baseline = "hostrig", True, True
        args.profile_classes = "none"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
    shared_gpu = os.environ.get("RTV_BENCH_SHARED_GPU") == "1"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU of this node
        if torch.cuda.device_count() < args.gpus and not shared_gpu:
            raise SystemExit(f"bench.py: --gpus {args.gpus} requested but only {torch.cuda.device_count()} GPU(s) visible")
        import socket
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        os.dup2(json_fd, 1)      # the launcher and its ranks inherit the real stdout
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a line that claims N GPUs must have run on N ranks: never fall back to fewer silently
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if world > 1 and torch.cuda.device_count() < world and not shared_gpu:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    # RTV_BENCH_SHARED_GPU=1 (test rigs with fewer GPUs than ranks): all ranks share cuda:0 and the collectives run over
    # gloo staged through the host -- exercises the multi-rank code path, the numbers mean nothing
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    elif args.cp_host_probe:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    from realtime_video_amd import ops
    from realtime_video_amd.causal_model import CausalWanModel
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    from realtime_video_amd.wan_wrapper import WanDiffusionWrapper

    if os.environ.get("RTV_ROPE_WAVE") is not None:      # A/B of the RoPE / cache kernel forms (include/rtv_hip_lab.h), diagnostic
        from realtime_video_amd import _lib
        _lib.load().rtv_rope_set_wave(int(os.environ["RTV_ROPE_WAVE"]))
    if os.environ.get("RTV_FRESH_TAP_SKIP") is not None:   # A/B of the fresh one-frame encode (last time tap only vs all 27 taps), diagnostic
        from realtime_video_amd import _lib
        import realtime_video_amd.vae_encoder  # noqa: F401  (registers the signature)
        _lib.call("rtv_vae_set_fresh_tap_skip", int(os.environ["RTV_FRESH_TAP_SKIP"]))
    if os.environ.get("RTV_DIRECT_V") is not None:       # A/B of the V cache write (GEMM epilogue vs copy, include/rtv_hip_lab.h), diagnostic
        from realtime_video_amd import _lib
        _lib.load().rtv_dit_set_direct_v(int(os.environ["RTV_DIRECT_V"]))
    mc = MODELS[args.model]
    model = CausalWanModel(dim=mc["dim"], ffn_dim=mc["ffn_dim"], num_heads=mc["num_heads"], num_layers=mc["num_layers"],
                           text_dim=4096, freq_dim=256, device=dev).init_random_weights(seed=0)
    model.gemm_tile_cfg = args.gemm_tile_cfg
    if args.hipgraph:
        model.use_hip_graphs = True      # (context parallel: the collectives are captured with the kernels, causal_model.py)
        args.profile_classes = "none"
    if args.fp8:
        model.enable_fp8()
    use_cp = (world > 1 and args.parallel == "cp") or args.cp_host_probe
    cp_world = world if use_cp else max(1, args.simulate_cp)
    # Context-parallel runs stay EAGER by default (r06).  VERDICT r05 asked for hipGraph replay as the --gpus N default; measured
    # first: with a one-rank RCCL group made to issue its collectives for real (r05's probe short-circuited them at world 1, so its
    # "captured with the RCCL collectives" never captured one), capturing the head-exchange forward segfaults inside
    # hipStreamEndCapture on ROCm 7.0.2 / RCCL 2.26.6 whenever an ASYNCHRONOUS all-to-all is part of the capture (a forked branch);
    # synchronous all-to-alls and asynchronous all-gathers capture and replay fine (profiles/r06_cp_graph_capture_bisect.txt).  A
    # segfault cannot be caught by a pre-flight, the first run on real xGMI must not die on it, and eager costs a rank 38-47 ms of
    # host time per block against ~117 ms of GPU work.  `--hipgraph` opts in (all-to-alls are then issued synchronously inside the
    # capture, parallel.py) after a small RCCL-in-graph pre-flight on every rank.  Either way a DIAGNOSTIC block behind the timed
    # region runs eagerly with every kernel class and every wait for a collective bracketed (config.cp_diagnostics).
    cp_graph_note = None
    if world > 1 and use_cp and args.hipgraph and not shared_gpu:
        ok, why = rccl_graph_preflight(dev, world, rank)
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        if float(flag.item()) < 1.0:
            model.use_hip_graphs = False
            cp_graph_note = f"eager: the RCCL-in-hipGraph pre-flight failed on some rank (this rank: {why or 'ok'})"
            print(f"[bench] rank {rank}: {cp_graph_note}", file=sys.stderr)
    if args.cp_attn_splits <= 0:   # the count that fills the 256 CUs best for this rank count (parallel.attn_kv_splits_for)
        from realtime_video_amd.parallel import attn_kv_splits_for
        args.cp_attn_splits = attn_kv_splits_for(cp_world, mc["num_heads"],
                                                 cus=torch.cuda.get_device_properties(dev).multi_processor_count)
    if use_cp:
        from realtime_video_amd.parallel import ContextParallel
        model.context_parallel = ContextParallel(exchange=args.cp_exchange, overlap=not args.no_cp_overlap,
                                                 attn_kv_splits=args.cp_attn_splits)
        model.context_parallel.force_single_rank = bool(args.cp_host_probe)
    if args.simulate_cp > 1:
        from realtime_video_amd.parallel import SimulatedContextParallel
        model.context_parallel = SimulatedContextParallel(args.simulate_cp, args.cp_exchange, attn_kv_splits=args.cp_attn_splits)
        args.no_vae = True
    wr = WanDiffusionWrapper(model, timestep_shift=5.0)

    class TimedTransformer:
        """hipEvents around every WanDiffusionWrapper.forward on the launch stream: BASELINE.json's "per-step DiT latency"
        (denoise steps) and the KV-recompute forward, collectives included under context parallelism."""

        def __init__(self, inner):
            self.inner, self.events, self.on = inner, [], False

        def __getattr__(self, name):
            return getattr(self.inner, name)

        def __call__(self, *a, **k):
            if not self.on:
                return self.inner(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.inner(*a, **k)
            e1.record()
            self.events.append((model.block_mask is not None, e0, e1))
            return out

    wr = TimedTransformer(wr)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3, denoising_step_list=[1000, 750, 500, 250]), dev,
                                   generator=wr)
    if args.no_vae:
        vae = None
    elif use_cp:   # row-sharded decode: every rank decodes one horizontal stripe, one all-gather of pixels per block
        from realtime_video_amd.parallel import ShardedVAEDecoder
        vae = ShardedVAEDecoder(model.context_parallel, dev).init_random_weights(seed=1)
    else:
        vae = VAEDecoderWrapper(dev).init_random_weights(seed=1)
    vae_enc = None if args.no_vae else VAEEncoderWrapper(device=dev).init_random_weights(seed=2)
    keep_first = bool(args.no_vae or args.keep_first_frame)   # reference default: False = re-encode the first context frame every block
    g = torch.Generator(device=dev).manual_seed(42)
    prompt = torch.zeros(1, 512, 4096, dtype=torch.bfloat16, device=dev)
    prompt[:, :64] = torch.randn(1, 64, 4096, generator=g, device=dev).to(torch.bfloat16)
    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(prompt), vae_decoder=vae,
                    vae_encoder=vae_enc)
    diag_blocks = 1 if use_cp else 0      # one eager, fully bracketed block behind the timed region (untimed; --cp-host-probe too)
    # (the session's noise tensor is drawn in one call whose values depend on its SIZE - the generator's grid follows numel - so every
    # run reserves the diagnostic block, context parallel or not: an N-rank run and the single-GPU run of the same command see the same
    # noise and their last timed block can be compared bit for bit, tests/test_context_parallel_gpu.py)
    # hipGraph replay: a forward geometry is captured on its SECOND sighting (block 2 for the recompute pass); with fewer than 3
    # warm-up blocks the capture would fall into the timed region, so the missing ones are run as extra untimed priming blocks
    priming = max(0, 3 - args.warmup) if model.use_hip_graphs else 0
    n_blocks = priming + args.warmup + args.steps + 1
    params = GenerateParams(prompt="synthetic", seed=42, kv_cache_num_frames=args.kv_cache_num_frames,
                            num_blocks=n_blocks, num_denoising_steps=args.denoising_steps, keep_first_frame=keep_first)
    # frame delivery (release_server.py:978-991): every block's pixels go to pinned host memory as rgb8 on the download
    # stream (rank 0 under context parallelism, every rank for replicas); the previous block's frames are fetched while
    # the next block runs, the last ones before the clock stops.
    downloader = None
    if vae is not None and (rank == 0 or not use_cp):
        from realtime_video_amd.frames import FrameDownloader
        downloader = FrameDownloader(dev, slots=2)
    tickets = []

    wait_s = [0.0]      # host time spent WAITING for the previous block's frames (a sync with the GPU, not launch issue)

    def deliver(pixels, frame_ids, event):
        if downloader is not None:
            if tickets:
                tw = time.perf_counter()
                downloader.fetch(tickets.pop())
                wait_s[0] += time.perf_counter() - tw
            tickets.append(downloader(pixels, frame_ids, event))

    sess = GenerationSession(params, models, frame_callback=deliver, device=dev)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(priming + args.warmup):
        sess.generate_block()
    barrier()
    ops.prof_reset()
    ops.dispatch_counts(reset=True)      # the kernel variants that run inside the timed region go into config.kernel_variants
    strides = {k: int(v) for k, v in (kv.split("=") for kv in args.profile_stride.split(",") if kv)}
    for cls in ("gemm", "attn", "layernorm", "rope", "conv", "misc"):
        ops.prof_set_stride(cls, 1 if args.profile_classes == "all" else strides.get(cls, 1))
    ops.prof_enable(args.profile_classes != "none",
                    None if args.profile_classes == "all" else [c for c in args.profile_classes.split(",") if c != "none"])
    wr.on = True
    wait_s[0] = 0.0
    t0 = time.perf_counter()
    frames = 0
    for _ in range(args.steps):
        out = sess.generate_block()
        frames += 12
    host_issue_s = time.perf_counter() - t0   # host side of the timed blocks (launch issue; the GPU may still be running)
    if tickets:
        downloader.fetch(tickets.pop())
    barrier()
    elapsed = time.perf_counter() - t0
    # the loop fetches block n-1's frames while block n is queued: that fetch blocks until the GPU has finished block n-1, so
    # the loop's wall time is mostly that wait; what is left is the host's own work (Python session loop + ctypes launches)
    host_ms = {"loop": 1e3 * host_issue_s / args.steps, "waiting_for_frames": 1e3 * wait_s[0] / args.steps,
               "launch_issue": 1e3 * (host_issue_s - wait_s[0]) / args.steps}
    print(f"[bench] rank {rank}: host loop {1e3 * host_issue_s / args.steps:.1f} ms per block of {1e3 * elapsed / args.steps:.1f} ms, "
          f"of which {1e3 * wait_s[0] / args.steps:.1f} ms waiting for the previous block's frames -> "
          f"launch issue {1e3 * (host_issue_s - wait_s[0]) / args.steps:.1f} ms per block", file=sys.stderr)
    ops.prof_enable(False)
    kernel_variants = ops.dispatch_counts()
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out.float()).all(), "non-finite output"
    # fingerprint of the LAST block's denoised latents ([1, 3, 16, 60, 104] bf16; seeds fixed): two builds / two boxes can be
    # compared - sha256 of the bytes (equal only if the kernels sum in the same order) and two order-insensitive moments
    import hashlib
    lat = sess.last_pred.detach().float().cpu()
    latents_checksum = {"sha256_bf16": hashlib.sha256(sess.last_pred.detach().cpu().contiguous().view(torch.int16).numpy().tobytes()).hexdigest()[:16],
                        "abs_sum": float(lat.double().abs().sum()), "mean": float(lat.double().mean()),
                        "shape": list(lat.shape)}

    step_ms = [e0.elapsed_time(e1) for rc, e0, e1 in wr.events if not rc]
    recompute_ms = [e0.elapsed_time(e1) for rc, e0, e1 in wr.events if rc]
    prof = {k: ops.prof_read(k) for k in ("gemm", "attn", "layernorm", "rope", "conv", "misc")}
    # what an event bracket reads with nothing inside (the two markers on either side of a dispatch): every bracketed launch's time
    # carries it, so the class times are corrected by it - without the correction the classes sum past the wall time of a block
    # (profiles/r04_bench_bracket_overhead.txt)
    bracket_ms = ops.prof_bracket_overhead(256) if args.profile_classes != "none" else 0.0
    for v in prof.values():
        v["ms_raw"], v["ms_class_raw"] = v["ms"], v["ms_class"]
        v["ms"] = max(v["ms"] - bracket_ms * v["launches"], 0.0)
        v["ms_class"] = v["ms"] * (v["seen_work"] / v["work"] if v["work"] > 0 else 1.0)
    peak_mem = torch.cuda.max_memory_allocated(dev)
    # ---- context parallel: the diagnostic block (eager, every kernel class bracketed, exposed-communication brackets on), then one
    # record per rank gathered to rank 0.  Outside the timed region; its brackets serialise neighbouring launches, so its own wall
    # time is not a throughput figure - the per-class kernel times and the waits for collectives are what it is for.
    cp_diag = None
    if diag_blocks:
        cpo = model.context_parallel
        graphs_were = model.use_hip_graphs
        model.use_hip_graphs = False
        wr.on = False
        barrier()
        ops.prof_reset()
        for cls in ("gemm", "attn", "layernorm", "rope", "conv", "misc"):
            ops.prof_set_stride(cls, 1)
        ops.prof_enable(True, None)
        cpo.start_timing()
        td = time.perf_counter()
        sess.generate_block()
        host_diag_ms = 1e3 * (time.perf_counter() - td)
        if tickets:
            downloader.fetch(tickets.pop())
        torch.cuda.synchronize()
        wall_diag_ms = 1e3 * (time.perf_counter() - td)
        ops.prof_enable(False)
        exposed = cpo.read_timing()
        model.use_hip_graphs = graphs_were
        bms = ops.prof_bracket_overhead(256)
        kinds = ["exchange_q", "exchange_kv", "exchange_o", "gather_kv", "head_rows", "vae_pixel_rows", "all_gather_rows", "all_to_all"]
        classes = ["gemm", "attn", "layernorm", "rope", "conv", "misc"]
        rec = []
        for c in classes:
            pr = ops.prof_read(c)
            rec.append(max(pr["ms"] - bms * pr["launches"], 0.0))
        for k in kinds:
            n, ms = exposed.get(k, (0, 0.0))
            rec += [float(n), max(ms - bms * n, 0.0)]
        rec += [host_diag_ms, wall_diag_ms, host_ms["launch_issue"], host_ms["loop"]]
        mine = torch.tensor(rec, device=dev, dtype=torch.float64)
        allr = torch.empty((world, mine.numel()), device=dev, dtype=torch.float64)
        if shared_gpu:      # gloo rig: gather through the host
            parts = [torch.empty(mine.numel(), dtype=torch.float64) for _ in range(world)]
            torch.distributed.all_gather(parts, mine.cpu())
            allr = torch.stack(parts)
        else:
            torch.distributed.all_gather_into_tensor(allr, mine)
        allr = allr.cpu().tolist()
        L, fwd = mc["num_layers"], args.denoising_steps + 1
        per_rank = []
        for r in range(world):
            row = allr[r]
            kern = dict(zip(classes, row[:len(classes)]))
            ex, off = {}, len(classes)
            for i, k in enumerate(kinds):
                n, ms = row[off + 2 * i], row[off + 2 * i + 1]
                if n:
                    ex[k] = {"waits": int(n), "ms_per_block": ms}
            tail = row[off + 2 * len(kinds):]
            per_layer = sum(v["ms_per_block"] for k, v in ex.items() if k in ("exchange_q", "exchange_kv", "exchange_o", "gather_kv"))
            per_rank.append({"rank": r, "kernel_ms_per_block": kern, "kernel_ms_per_block_sum": sum(kern.values()),
                             "exposed_collective_ms_per_block": ex, "exposed_collective_ms_per_block_sum": sum(v["ms_per_block"] for v in ex.values()),
                             "exposed_collective_ms_per_layer": per_layer / (L * fwd),
                             "host_ms_diagnostic_block_eager": tail[0], "wall_ms_diagnostic_block_eager": tail[1],
                             "host_launch_issue_ms_per_timed_block": tail[2], "host_loop_ms_per_timed_block": tail[3]})
        cp_diag = {"what": "ONE eager block behind the timed region on every rank: all kernel classes bracketed with hipEvents (bracket "
                           "overhead subtracted), and every point where the compute stream waits for a collective bracketed on that "
                           "stream - Pending.wait() of the asynchronous exchanges (what the overlap did not hide), the whole call of the "
                           "synchronous ones (exchange_o, head rows, VAE pixel rows).  per-layer = the four per-layer exchanges / "
                           f"({L} layers x {fwd} forwards)",
                   "timed_region": "hipGraph replay (kernels + RCCL collectives captured per forward geometry)" if graphs_were else "eager",
                   "per_rank": per_rank}
    if rank != 0:
        return
    total_frames = frames if use_cp or world == 1 else frames * world  # replicas: every rank generates its own stream
    gm = prof["gemm"]
    achieved = gm["work"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
    fwd_per_block = args.denoising_steps + 1
    # algorithmic TFLOP of one block of the contract configuration
    block_tflop = None
    if world == 1 and not args.no_vae and not args.simulate_cp and args.kv_cache_num_frames == 3 and args.denoising_steps == 4:
        block_tflop = {"14b": 4 * 149.8 + 131.8 + 40.65 + 2.72, "1.3b": 4 * 20.2 + 16.2 + 40.65 + 2.72}.get(args.model)
    traffic, traffic_src, traffic_n = measured_traffic(args.model) if world == 1 and not args.fp8 else (None, None, None)
    result = {
        "metric": "frames/sec at 832x480, 4-step 14B T2V (per-step DiT latency in config)",
        "value": total_frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong" if use_cp else "weak",
        "vs_baseline": ((total_frames / elapsed) / 11.0
                        if args.model == "14b" and world == 1 and not args.no_vae and not args.fp8 and not args.simulate_cp
                        and not args.cp_host_probe else None),
        "dtype": "fp8 e4m3 linears (per-tensor dynamic activations, fp32 accumulation), bf16 elsewhere" if args.fp8 else "bf16",
        "data": "synthetic (random-init weights of the named architecture, N(0,1) latents/noise, N(0,1) prompt embeddings)",
        "config": {
            "workload": f"{mc['name']}, 832x480 (latent 60x104, 1560 tokens/frame), {args.denoising_steps} denoising steps, "
                        f"kv_cache_num_frames={args.kv_cache_num_frames}, 3 latent frames (12 pixel frames) per block, "
                        f"KV-recompute forward every block, first-frame VAE re-encode every block, streaming VAE decode {'OFF (INVALID: diagnostic run)' if args.no_vae else 'on (fp16)'}",
            "model": args.model,
            "frame_delivery": None if args.no_vae else "rgb8 [T,H,W,3] in pinned host memory per block (GPU-side conversion + async "
                                                       "copy on a download stream), inside the timed region",
            "keep_first_frame": keep_first,
            "note": "reference default keep_first_frame=False: from block 2 on every block re-encodes the first context "
                    "frame through the streaming VAE encoder (release_server.py:572-575); warm-up >= 2 blocks puts the timed "
                    "blocks in that steady state",
            "parallelism": "single GPU" if world == 1 else (
                f"cp{world}: one stream, token axis sharded {world}-way, "
                + ("self-attention head-sharded through two all-to-alls per layer over RCCL (KV cache holds "
                   f"{mc['num_heads'] // world} of {mc['num_heads']} heads per rank)"
                   if model.context_parallel.head_exchange(mc["num_heads"]) else "K/V all-gather per layer over RCCL")
                + "; VAE decode "
                f"sharded by output rows ({world} stripes + conv halos, one pixel all-gather per block), first-frame "
                f"re-encode replicated" if use_cp else f"{world} independent replicas"),
            "cp_attn_kv_splits": args.cp_attn_splits if (use_cp or args.simulate_cp > 1) else None,
            "cp_hipgraph_replay": bool(model.use_hip_graphs) if use_cp else None,
            "cp_hipgraph_note": cp_graph_note,
            "hipgraph_priming_blocks": priming,
            "cp_diagnostics": cp_diag,
            "dit_ms_per_denoise_step": sum(step_ms) / max(1, len(step_ms)),        # BASELINE.json "per-step DiT latency"
            "dit_ms_per_recompute_forward": sum(recompute_ms) / max(1, len(recompute_ms)) if recompute_ms else None,
            # kernel-class times exist only for the classes bracketed with events (--profile-classes; 'all' = diagnostic)
            "dit_ms_per_forward": ((prof["gemm"]["ms_class"] + prof["attn"]["ms_class"] + prof["layernorm"]["ms_class"]
                                    + prof["rope"]["ms_class"] + prof["misc"]["ms_class"]) / (args.steps * fwd_per_block)
                                   if args.profile_classes == "all" else None),
            "kernel_ms_per_block": {k: v["ms_class"] / args.steps for k, v in prof.items() if v["launches"] > 0},
            # host side of a block: the session loop's wall time, the part of it spent blocked on the previous block's frames
            # (a wait on the GPU, not work) and the rest = Python + ctypes launch issue (6 ms under --hipgraph)
            "host_ms_per_block": host_ms,
            # --cp-host-probe: the host side of ONE rank's context-parallel block (5 forwards x 40 layers x (4 C calls + 3
            # collectives + event fences)), no queue back-pressure; to be held against the ~134 ms of GPU work a rank has at 8 ranks
            "cp_host_ms_per_block": host_ms["launch_issue"] if args.cp_host_probe else None,
            "cp_forwards_issued_from_python": getattr(model, "cp_forwards_issued", 0) if use_cp else None,   # (graph replays excluded)
            # ms an EMPTY event bracket reads on the launch stream, already subtracted per bracketed launch from every class time
            # and rate of this line (kernel_ms_per_block, roofline.*); `kernel_ms_per_block_uncorrected` = the raw brackets
            "event_bracket_overhead_us": 1e3 * bracket_ms,
            "kernel_ms_per_block_uncorrected": {k: v["ms_class_raw"] / args.steps for k, v in prof.items() if v["launches"] > 0},
            "max_memory_allocated_GB": peak_mem / 1e9,
            # launches per kernel VARIANT inside the timed region (rtv_dispatch_counts; under --hipgraph the replayed launches are not
            # counted: the dispatch ran at capture time): a shape falling back to an older kernel shows here
            "kernel_variants": kernel_variants,
            "last_block_latents_checksum": latents_checksum,
        },
        "roofline": {
            "kernel": "gemm8_kernel / gemm_kernel (bf16 MFMA projection GEMMs with fused epilogues: all DiT linears)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": MFMA_PEAK_TFLOPS * (2.0 if args.fp8 else 1.0),
            "unit": "TFLOP/s",
            "frac": achieved / (MFMA_PEAK_TFLOPS * (2.0 if args.fp8 else 1.0)),
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_launches_sampled": traffic_n,
            "algorithmic_bytes_per_launch": None if args.fp8 else gemm_algorithmic_bytes(mc),
            "frac_of_sustained_mfma": achieved / (MFMA_SUSTAINED_FP8_TFLOPS if args.fp8 else MFMA_SUSTAINED_TFLOPS),
            # the whole block against the same peak: algorithmic flops of a block (DiT GEMMs + attention of the 4 denoise
            # and the recompute forward, VAE decode, first-frame re-encode = 774.5 TF for the contract configuration) / wall time
            "whole_block_frac": (block_tflop / (elapsed / args.steps) / (MFMA_PEAK_TFLOPS * (2.0 if args.fp8 else 1.0))
                                 if block_tflop is not None else None),
            "whole_block_TFLOP": block_tflop,
            "whole_block_note": ("SURVEY 8d's algorithmic count, incl. 2.72 TF for the first-frame re-encode as the reference runs it; since "
                                 "r06 the native encoder skips the 1.8 TF of it that multiply the two zero cache slices of a fresh "
                                 "stream (0.23 % of the block's count)") if block_tflop is not None else None,
            "launches": gm["launches"],                  # bracketed with events (every `sample_stride`-th launch of the class)
            "launches_in_timed_region": gm["seen_launches"],
            "sample_stride": 1 if args.profile_classes == "all" else strides.get("gemm", 1),
            "avg_launch_ms": gm["ms"] / max(gm["launches"], 1),
            "attention_TFLOPs": prof["attn"]["work"] / (prof["attn"]["ms"] * 1e-3) / 1e12 if prof["attn"]["ms"] > 0 else None,
            "conv_TFLOPs": prof["conv"]["work"] / (prof["conv"]["ms"] * 1e-3) / 1e12 if prof["conv"]["ms"] > 0 else None,
            # the other kernel classes against THEIR roofline (same event brackets; only the classes named in --profile-classes):
            # attention / VAE conv against the dense MFMA peak, the row kernels against 8 TB/s of HBM (algorithmic bytes; a plain
            # copy of the same tensors reaches 0.59-0.67 of that peak on this chip, profiles/r04_row_kernels.txt)
            "other_kernels": {
                name: ({"bound": "mfma", "achieved_TFLOPs": v["work"] / (v["ms"] * 1e-3) / 1e12,
                        "frac": v["work"] / (v["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "ms_per_block": v["ms_class"] / args.steps,
                        "launches_bracketed": v["launches"], "launches": v["seen_launches"]}
                       if name in ("attn", "conv") else
                       {"bound": "hbm", "achieved_TBps": v["work"] / (v["ms"] * 1e-3) / 1e12,
                        "frac": v["work"] / (v["ms"] * 1e-3) / 1e12 / HBM_PEAK_TBPS, "ms_per_block": v["ms_class"] / args.steps,
                        "launches_bracketed": v["launches"], "launches": v["seen_launches"]})
                for name, v in prof.items() if name in ("attn", "conv", "layernorm", "rope") and v["ms"] > 0},
        },
    }
    if not args.no_cpu_baseline and world == 1:      # the CPU baseline belongs to the N = 1 line only (rank 0's host cores, ~2 min)
        result["cpu_baseline"] = cpu_baseline(mc, with_vae=not args.no_vae)
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(result) + "\n").encode())


if __name__ == "__main__":
    main()
