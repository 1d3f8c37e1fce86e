"""World-size-2 / 4 / 8 gloo tests (CPU) of the context-parallel decomposition used on the GPUs
(realtime_video_amd/parallel.py + the phase API of rtv_dit_*): the token axis is cut into contiguous
shards, per-token work runs on local rows, ONE in-place all-gather per layer moves the new K/V rows into
the replicated cache.  The arithmetic here is the CPU oracle; what is under test is the sharding math
(global frame / position lookup from local rows, cache row placement, collective placement) — the same
host code drives RCCL on the MI355Xs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import wan_oracle as wo
from realtime_video_amd.parallel import ContextParallel, shard_rows

GRID = (2, 4, 6)     # F, gh, gw  -> 48 tokens, 24 per frame
# two heads by default; the wider-world cases (one head per rank at 4 / 8 ranks) set the environment variable before they spawn
# their ranks, which import this module afresh
H = int(os.environ.get("RTV_CP_TEST_HEADS", "2"))
D, FFN = 128 * H, 256 * H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    cfg = dict(dim=D, ffn_dim=FFN, num_heads=H, num_layers=1)
    w = wo.make_weights(cfg, seed=4, text_dim=64, dtype=torch.float32)
    g = torch.Generator().manual_seed(8)
    M = GRID[0] * GRID[1] * GRID[2]
    x = torch.randn(1, M, D, generator=g)
    e = torch.randn(1, GRID[0], 6, D, generator=g) * 0.2
    ctx = torch.randn(1, 16, D, generator=g)
    prev_k = torch.randn(1, 20, H, 128, generator=g)   # 20 rows already in the cache (an earlier block)
    prev_v = torch.randn(1, 20, H, 128, generator=g)
    return cfg, w, x, e, ctx, prev_k, prev_v


def _attn(q, k, v):
    return wo.attention_sdpa(q, k, v, dtype=None)


def _cp_block(w, x_local, e, ctx, k_cache, v_cache, row0, cp, M):
    """attention_block (causal_model.py:440-492) on the local token shard; mirrors rtv_dit_layer_qkv /
    gather / rtv_dit_layer_rest."""
    pre = "blocks.0"
    r0, rc = cp.shard(M)
    fs = GRID[1] * GRID[2]
    frame = (torch.arange(r0, r0 + rc) // fs)
    em = (w[pre + ".modulation"].unsqueeze(1) + e)[0][frame]            # [rc, 6, D] per-row modulation
    freqs = wo.rope_table(128)

    def lin(t, name):
        return torch.nn.functional.linear(t, w[name + ".weight"], w[name + ".bias"])

    def rope_local(t):  # RoPE by global token position: embed the shard in a full-length tensor
        full = torch.zeros(1, M, H, 128)
        full[0, r0:r0 + rc] = t
        return wo.rope_apply(full, GRID, freqs, start_frame=3)[0, r0:r0 + rc]

    h = wo.layer_norm(x_local) * (1 + em[:, 1]) + em[:, 0]
    sa = pre + ".self_attn"
    q = wo.rms_norm(lin(h, sa + ".q"), w[sa + ".norm_q.weight"]).view(rc, H, 128)
    k = wo.rms_norm(lin(h, sa + ".k"), w[sa + ".norm_k.weight"]).view(rc, H, 128)
    v = lin(h, sa + ".v").view(rc, H, 128)
    if cp.head_exchange(H):
        # rtv_dit_layer_qkv_hp / exchange_qkv / layer_attn_hp / exchange_o / layer_rest_hp (include/rtv_hip.h): the caches
        # passed in hold this rank's heads only
        W, hn = cp.world, H // cp.world
        gc = hn * 128
        bufs = {"hn": hn, "q_send": rope_local(q).reshape(rc, W, gc).transpose(0, 1).contiguous(),
                "kv_send": torch.stack([rope_local(k).reshape(rc, W, gc), v.reshape(rc, W, gc)], 2).transpose(0, 1).contiguous(),
                "q_all": torch.empty(M, gc), "o_all": None, "o_recv": torch.empty(W, rc, gc)}
        # the overlapped form the GPU path uses: both exchanges in flight, waited for right before attention
        pend_q = cp.exchange_q([(cp.rank, bufs)], async_op=True)
        pend_kv = cp.exchange_kv([(cp.rank, bufs)], k_cache[0], v_cache[0], row0, M, async_op=True)
        pend_q.wait()
        pend_kv.wait()
        bufs["o_all"] = _attn(bufs["q_all"].view(1, M, hn, 128), k_cache[:, :row0 + M], v_cache[:, :row0 + M])[0].reshape(M, gc)
        cp.exchange_o([(cp.rank, bufs)])
        out = bufs["o_recv"].transpose(0, 1).reshape(rc, H, 128)
    else:
        k_cache[0, row0 + r0:row0 + r0 + rc] = rope_local(k)
        v_cache[0, row0 + r0:row0 + r0 + rc] = v
        pend = cp.gather_kv(k_cache[0], v_cache[0], row0, M, async_op=True)   # the one exchange of the layer, in flight ...
        rq = rope_local(q).unsqueeze(0)                                       # ... under the q projection's RoPE
        pend.wait()
        out = _attn(rq, k_cache[:, :row0 + M], v_cache[:, :row0 + M])[0]
    x = x_local + lin(out.flatten(1), sa + ".o") * em[:, 2]
    ca = pre + ".cross_attn"
    hq = wo.rms_norm(lin(wo.layer_norm(x, 1e-6, w[pre + ".norm3.weight"], w[pre + ".norm3.bias"]), ca + ".q"),
                     w[ca + ".norm_q.weight"]).view(1, rc, H, 128)
    ck = wo.rms_norm(lin(ctx, ca + ".k"), w[ca + ".norm_k.weight"]).view(1, -1, H, 128)
    cv = lin(ctx, ca + ".v").view(1, -1, H, 128)
    x = x + lin(_attn(hq, ck, cv)[0].flatten(1), ca + ".o")
    h = wo.layer_norm(x) * (1 + em[:, 4]) + em[:, 3]
    y = lin(torch.nn.functional.gelu(lin(h, pre + ".ffn.0"), approximate="tanh"), pre + ".ffn.2")
    return x + y * em[:, 5]


def _worker(rank, world, port, interleaved, exchange, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        cp = ContextParallel(exchange=exchange)
        cfg, w, x, e, ctx, prev_k, prev_v = _inputs()
        M = x.shape[1]
        kv_size = 20 + M
        hc = H // world if exchange == "heads" else H          # heads held by this rank's cache
        if exchange == "heads":
            prev_k, prev_v = (t[:, :, rank * hc:(rank + 1) * hc] for t in (prev_k, prev_v))
        if interleaved:
            arena = torch.zeros(1, kv_size, 2, hc, 128)
            kc, vc = arena[:, :, 0], arena[:, :, 1]
        else:
            kc, vc = torch.zeros(1, kv_size, hc, 128), torch.zeros(1, kv_size, hc, 128)
        kc[:, :20], vc[:, :20] = prev_k, prev_v
        r0, rc = cp.shard(M)
        out_local = _cp_block(w, x[0, r0:r0 + rc], e, ctx, kc, vc, 20, cp, M)
        full = torch.zeros(M, D)
        full[r0:r0 + rc] = out_local
        cp.all_gather_rows_(full)                                       # head-output style gather
        if exchange == "heads":                                          # reassemble the head-sharded caches for the check
            parts = [torch.empty_like(kc) for _ in range(world)], [torch.empty_like(vc) for _ in range(world)]
            dist.all_gather(parts[0], kc.contiguous())
            dist.all_gather(parts[1], vc.contiguous())
            kc, vc = torch.cat(parts[0], 2), torch.cat(parts[1], 2)
        if rank == 0:
            ret["out"], ret["k"], ret["v"] = full, kc.clone(), vc.clone()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,interleaved,exchange", [(2, True, "rows"), (2, False, "rows"), (2, True, "heads"), (2, False, "heads"),
                                                        (4, True, "heads"), (8, True, "heads"), (8, True, "rows")])
def test_context_parallel_block_equals_unsharded(world, interleaved, exchange, monkeypatch):
    """world 4 / 8: one head per rank under the head exchange (H = world), six token rows per rank at 8 ranks - the geometry of
    `bench.py --gpus 8` (585 rows, 5 heads per rank) in small."""
    import sys
    heads = max(2, world)
    monkeypatch.setenv("RTV_CP_TEST_HEADS", str(heads))          # the spawned ranks import this module with it
    me = sys.modules[__name__]
    monkeypatch.setattr(me, "H", heads)
    monkeypatch.setattr(me, "D", 128 * heads)
    monkeypatch.setattr(me, "FFN", 256 * heads)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), interleaved, exchange, ret), nprocs=world, join=True)
    cfg, w, x, e, ctx, prev_k, prev_v = _inputs()
    M = x.shape[1]
    kv = {"k": torch.zeros(1, 20 + M, H, 128), "v": torch.zeros(1, 20 + M, H, 128),
          "global_end_index": 3 * 24, "local_end_index": 20}
    kv["k"][:, :20], kv["v"][:, :20] = prev_k, prev_v
    ca = {"k": None, "v": None, "is_init": False}
    # unsharded oracle block; FRAME_SEQLEN is hard-coded to 1560 in the reference, so drive start_frame via monkeypatch
    old = wo.FRAME_SEQLEN
    wo.FRAME_SEQLEN = 24
    try:
        ref = wo.attention_block(w, "blocks.0", x, e, GRID, wo.rope_table(128), ctx, H, kv, ca, 3 * 24, False,
                                 attn_fn=_attn)
    finally:
        wo.FRAME_SEQLEN = old
    assert torch.allclose(ret["out"], ref[0], atol=2e-5, rtol=1e-5)
    # (fp32 host GEMMs over 6-row shards and over 48 rows block their sums differently: 1e-6 at 256 columns, a few 1e-6 at 1024)
    tol = 1e-6 if heads == 2 else 1e-5
    assert torch.allclose(ret["k"], kv["k"], atol=tol) and torch.allclose(ret["v"], kv["v"], atol=tol)


def test_shard_rows():
    assert shard_rows(4680, 8, 3) == (1755, 585)
    assert [shard_rows(4680, 2, r) for r in range(2)] == [(0, 2340), (2340, 2340)]
    with pytest.raises(ValueError):
        shard_rows(100, 8, 0)


def _stripe_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from realtime_video_amd.parallel import gather_row_stripes
        cp = ContextParallel()
        out = {}
        for H in (16, 15):   # even and uneven stripe heights
            full = torch.arange(2 * 3 * H * 4, dtype=torch.float32).view(2, 3, H, 4)
            r0, r1 = H * rank // world, H * (rank + 1) // world
            out[H] = gather_row_stripes(cp, full[:, :, r0:r1].contiguous(), H)
        ret[rank] = out
    finally:
        dist.destroy_process_group()


def test_gather_row_stripes_world2():
    """The pixel-stripe all-gather of the row-sharded VAE decode (parallel.ShardedVAEDecoder)."""
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_stripe_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for rank in range(world):
        for H in (16, 15):
            full = torch.arange(2 * 3 * H * 4, dtype=torch.float32).view(2, 3, H, 4)
            assert torch.equal(ret[rank][H], full), (rank, H)


def test_attn_kv_split_rule():
    """parallel.attn_kv_splits_for: the number of key ranges that fills 256 CUs best for a rank's attention launch (bench.py uses it
    for --cp-attn-splits 0): 14B -> 1 / 2 / 4 / 2 at 1 / 2 / 4 / 8 ranks, never more ranges than ranks (the partials share the
    DiT workspace: splits x local heads <= heads)."""
    from realtime_video_amd.parallel import attn_kv_splits_for
    assert [attn_kv_splits_for(w, 40, cus=256) for w in (1, 2, 4, 8)] == [1, 2, 4, 2]     # the MI355X: 256 CUs, stated here
    for heads in (12, 16, 40):
        for w in (1, 2, 4, 8):
            s = attn_kv_splits_for(w, heads, cus=256)
            assert s in (1, 2, 4) and s <= max(1, w)
