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
