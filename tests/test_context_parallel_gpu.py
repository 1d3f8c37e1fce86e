"""Two real processes driving the context-parallel path (ContextParallel + phase API + row-sharded VAE decode + HIP
kernels) on the MI355X.  On a ONE-GPU box both ranks share cuda:0 and the collectives run over gloo (staged through the
host); the `nccl` variants run the identical session with one GPU per rank over RCCL / xGMI and skip themselves when fewer
than two GPUs are visible - a multi-GPU driver runs them unmodified (bench.py --gpus N is the same code path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_session(cp, dev="cuda:0"):
    from oracle import wan_oracle as wo
    from oracle.make_golden import TEXT_DIM, TINY
    from realtime_video_amd.causal_model import CausalWanModel
    from realtime_video_amd.pipeline import CausalInferencePipeline, make_args
    from realtime_video_amd.session import GenerateParams, GenerationSession, Models, StaticTextEncoder
    from realtime_video_amd.parallel import ShardedVAEDecoder
    from realtime_video_amd.vae_decoder import VAEDecoderWrapper
    from realtime_video_amd.vae_encoder import VAEEncoderWrapper
    from realtime_video_amd.wan_wrapper import WanDiffusionWrapper
    cfg = dict(TINY)
    w = wo.make_weights(cfg, seed=0, text_dim=TEXT_DIM)
    model = CausalWanModel(dim=cfg["dim"], ffn_dim=cfg["ffn_dim"], num_heads=cfg["num_heads"],
                           num_layers=cfg["num_layers"], text_dim=TEXT_DIM, device=dev)
    model.load_state_dict(w)
    model.context_parallel = cp
    wr = WanDiffusionWrapper(model, timestep_shift=5.0)
    pipe = CausalInferencePipeline(make_args(num_frame_per_block=3), dev, generator=wr)
    g = torch.Generator().manual_seed(5)
    prompt = torch.zeros(1, 512, TEXT_DIM, dtype=torch.bfloat16)
    prompt[0, :64] = torch.randn(64, TEXT_DIM, generator=g).to(torch.bfloat16)
    # VAE: row-sharded decode + one all-gather of the stripes under context parallelism, whole frames otherwise; the
    # first-frame re-encode (keep_first_frame=False, block 2) consumes the gathered pixels
    dec = (ShardedVAEDecoder(cp, dev) if cp is not None else VAEDecoderWrapper(dev)).init_random_weights(seed=1)
    enc = VAEEncoderWrapper(device=dev).init_random_weights(seed=2)
    if os.environ.get("RTV_DBG_ENC"):
        _enc, rec = enc, []

        def enc(frames, cache, stream=False):   # noqa: F811
            mu, c = _enc(frames, cache, stream=stream)
            rec.append((frames.float().cpu(), mu.float().cpu()))
            return mu, c
        _run_session.rec = rec
    models = Models(transformer=wr, pipeline=pipe, text_encoder=StaticTextEncoder(prompt.to(dev)), vae_decoder=dec,
                    vae_encoder=enc)
    sess = GenerationSession(GenerateParams(seed=3, num_blocks=3, num_denoising_steps=4, keep_first_frame=False),
                             models, device=dev)
    outs = [sess.generate_block().clone() for _ in range(3)]
    torch.cuda.synchronize()
    return [o.cpu() for o in outs] + [sess.all_latents.cpu()], pipe.kv_cache1[1]["k"].cpu()


def _worker(rank, world, port, exchange, backend, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL between processes)
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from realtime_video_amd.parallel import ContextParallel
        outs, k = _run_session(ContextParallel(exchange=exchange), dev)
        ret[rank] = (outs, k)
        if os.environ.get("RTV_DBG_ENC"):
            ret[f"enc{rank}"] = list(_run_session.rec)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("exchange", ["rows", "heads"])
def test_two_process_context_parallel_session_equals_single_process(exchange, backend):
    """exchange="rows": K/V all-gather into replicated caches; "heads": all-to-all pair, every rank holding only its own
    heads of the KV cache (allocated head-sharded by the pipeline's cache manager).  backend "nccl": one GPU per rank,
    RCCL collectives on the communication stream overlapped with the projections (needs >= 2 GPUs)."""
    world = 2
    if backend == "nccl" and torch.cuda.device_count() < world:
        pytest.skip(f"RCCL variant needs {world} GPUs, {torch.cuda.device_count()} visible")
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), exchange, backend, ret), nprocs=world, join=True)
    ref_outs, ref_k = _run_session(None)
    hn = ref_k.shape[2] // world
    for rank in range(world):
        outs, k = ret[rank]
        for a, b in zip(outs, ref_outs):
            assert torch.equal(a, b), rank           # same kernels, same data: bit-identical on every rank
        if exchange == "heads":
            assert k.shape[2] == hn
            assert torch.equal(k, ref_k[:, :, rank * hn:(rank + 1) * hn])
        else:
            assert torch.equal(k, ref_k)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("exchange", ["heads", "rows"])
def test_eight_rank_bench_launcher_path_equals_single_process(exchange):
    """Eight-rank readiness without an eight-GPU node: `bench.py --gpus 8` through ITS OWN launcher path
    (re-exec under torch.distributed.run, one process per rank, ContextParallel + row-sharded VAE decode + pixel all-gather +
    frame delivery), all ranks sharing cuda:0 with gloo collectives (RTV_BENCH_SHARED_GPU=1), on the test-rig model (8 heads: one
    head per rank under the head exchange; 585 token rows per rank).  The last block's latents must be bit-identical with the
    single-process run of the same command.  attn_kv_splits 1 and GEMM tile config 4 (no split-K): the default dispatch splits K
    on tail tiles by the launch's tile count, i.e. by the shard's row count, which re-associates fp32 sums (scripts/
    gemm_shard_identity.py: N = 2048, K = 1024 at M = 4680 vs its shards) - with it sharded and unsharded agree to one-ulp bf16
    flips only (tests/test_dit_gpu.py::test_full_width_layer_context_parallel_equals_unsharded states that tolerance)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RTV_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)

    def run(n):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--model", "tiny", "--steps", "1", "--warmup", "2",
               "--no-cpu-baseline", "--cp-exchange", exchange, "--cp-attn-splits", "1", "--profile-classes", "none",
               "--gemm-tile-cfg", "4"]
        res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=800, cwd=root)
        assert res.returncode == 0, res.stderr[-3000:]
        line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)

    one, eight = run(1), run(8)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong" and one["n_gpus"] == 1
    assert "cp8" in eight["config"]["parallelism"]
    a, b = one["config"]["last_block_latents_checksum"], eight["config"]["last_block_latents_checksum"]
    assert a["shape"] == b["shape"] == [1, 3, 16, 60, 104]
    assert a["sha256_bf16"] == b["sha256_bf16"], (a, b)


@pytest.mark.timeout(600)
def test_bench_context_parallel_line_carries_per_rank_diagnostics():
    """`bench.py --gpus N` is the first thing that will ever run on real xGMI, so its one JSON line has to be enough to diagnose that
    run: per rank the kernel time per class, the time the compute stream WAITED for each kind of collective
    (per block and per layer) and the host time.  Exercised here on what a one-GPU box offers: (a) two ranks sharing cuda:0 over
    gloo through the launcher path (eager: the gloo route cannot be captured), (b) a one-rank RCCL group with the forwards replayed
    from hipGraphs - the default of a real N-rank run - followed by the eager diagnostic block."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)

    def run(args, **extra):
        res = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args + ["--steps", "1", "--warmup", "2", "--no-cpu-baseline"],
                             capture_output=True, text=True, env=dict(env, **extra), timeout=500, cwd=root)
        assert res.returncode == 0, res.stderr[-3000:]
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])

    two = run(["--gpus", "2", "--model", "tiny"], RTV_BENCH_SHARED_GPU="1")
    d = two["config"]["cp_diagnostics"]
    assert two["config"]["cp_hipgraph_replay"] is False and d["timed_region"] == "eager" and len(d["per_rank"]) == 2
    for r, rec in enumerate(d["per_rank"]):
        assert rec["rank"] == r and rec["kernel_ms_per_block"]["gemm"] > 0 and rec["kernel_ms_per_block"]["attn"] > 0
        ex = rec["exposed_collective_ms_per_block"]
        assert {"exchange_o", "head_rows", "vae_pixel_rows"} <= set(ex)           # the synchronous ones are always bracketed
        assert ex["exchange_o"]["waits"] == 2 * 5 - 1                             # 2 layers x 5 forwards, minus the kv-only last layer
        assert rec["exposed_collective_ms_per_layer"] > 0 and rec["host_launch_issue_ms_per_timed_block"] > 0
    one = run(["--cp-host-probe", "--hipgraph"])
    d = one["config"]["cp_diagnostics"]
    assert one["config"]["cp_hipgraph_replay"] is True and d["timed_region"].startswith("hipGraph replay")
    rec = d["per_rank"][0]
    # (the q and k|v all-to-alls share ONE wait, for the later of the two - parallel.wait_in_order -, bracketed as exchange_kv)
    assert rec["kernel_ms_per_block"]["gemm"] > 0 and set(rec["exposed_collective_ms_per_block"]) >= {"exchange_kv", "exchange_o"}
    assert rec["exposed_collective_ms_per_block"]["exchange_kv"]["waits"] == 40 * 5
    assert rec["exposed_collective_ms_per_block"]["exchange_o"]["waits"] == 40 * 5 - 1
