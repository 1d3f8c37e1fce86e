"""Context (sequence) parallelism for the causal DiT forward — one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference only has sequence parallelism for the NON-causal model (xfuser USP monkey-patch,
wan/distributed/xdit_context_parallel.py:131-142,:179-184); this is the new design for the causal path
(SURVEY.md §8e): the token axis of every forward is cut into `world` contiguous shards, all per-token
work (LayerNorm/modulation, every GEMM, RMSNorm, RoPE, cross-attention against the replicated text K/V,
FFN, head) runs on the local shard only, and the single exchange per layer is an in-place all-gather of
the new block's roped K and V rows into the (replicated) KV cache; queries then attend the full window
locally.  One more all-gather returns the head output rows.  xGMI is point-to-point, so the per-layer
message is deliberately one contiguous buffer per rank (K and V rows interleaved in the cache arena).

Two exchange patterns around self-attention (`ContextParallel(exchange=...)`):
  "rows"  - the all-gather above: KV cache replicated, 2*M*d*(w-1)/w elements received per rank and layer;
  "heads" - what xFuserLongContextAttention does for the reference (xdit_context_parallel.py:179-184): an all-to-all
            turns the token shard into a head shard (rank r attends ALL rows for heads [r*H/w, (r+1)*H/w) over a cache that
            holds only those heads) and a second one turns the output back: 4*(M/w)*d*(w-1)/w elements per rank and
            layer, i.e. w/2 times less xGMI traffic and 1/w of the cache memory.  Needs H % w == 0.
"auto" (default) picks "heads" whenever the head count divides.
"""
import torch
import torch.distributed as dist


def shard_rows(M, world, rank):
    """Contiguous token shard of rank `rank`: (row_begin, row_count).  M must divide evenly (4680 = 8*585)."""
    if M % world:
        raise ValueError(f"token count {M} is not divisible by the context-parallel degree {world}")
    n = M // world
    return rank * n, n


class _ExchangeChoice:
    exchange = "auto"
    # A ONE-rank group normally takes no collective at all.  True (bench.py --cp-host-probe): it runs the context-parallel code
    # path in full - head exchange, every collective issued on the one-rank group as a self-exchange through the same
    # torch.distributed / RCCL host path, fences and all - so that the host cost of a rank can be measured on a one-GPU box.
    force_single_rank = False

    def head_exchange(self, num_heads):
        """True when self-attention runs head-sharded (all-to-all exchange) for a model with `num_heads` heads."""
        if self.exchange not in ("auto", "heads", "rows"):
            raise ValueError(f"unknown exchange {self.exchange!r}")
        if (self.world == 1 and not self.force_single_rank) or self.exchange == "rows":
            return False
        if num_heads % self.world:
            if self.exchange == "heads":
                raise ValueError(f"head exchange needs num_heads ({num_heads}) divisible by the world size ({self.world})")
            return False
        return True


def attn_kv_splits_for(world, num_heads, q_tiles=19, cus=None):
    """How many key ranges a context-parallel rank's self-attention launch should be cut into (ContextParallel(attn_kv_splits=)).
    One rank's launch has num_heads x q_tiles / world workgroups (either exchange), one per CU at a time: rounds of `cus`.  S
    ranges make S x as many workgroups of 1/S the length, + ~6 % per extra range for the merge kernel and the per-workgroup
    prologue (profiles/r03_attn_kv_split_ab.log).  14B (40 heads): 2 / 4 / 2 ranges at 2 / 4 / 8 ranks; 1 rank: 1."""
    if world <= 1:
        return 1
    if cus is None:     # the device's real CU count (the same number the GEMM dispatch and plan_split_k use), 256 without a GPU
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count if torch.cuda.is_available() else 256
    g = max(1, num_heads * q_tiles // world)
    cost = {s: -(-g * s // cus) / s * (1 + 0.06 * (s - 1)) for s in (1, 2, 4) if s <= world}
    return min(cost, key=lambda s: (cost[s], s))


class Pending:
    """A collective in flight (issued with async_op=True on the process group's communication stream, RCCL running beside the
    compute stream) plus the work that has to follow it; `wait()` makes the CURRENT stream wait for it - the host does not
    block - and runs the follow-up.  `Pending()` is an already completed exchange.  `timer` = (ContextParallel, kind): the wait
    is bracketed with events on the compute stream when that object collects timings (`ContextParallel.start_timing`)."""

    def __init__(self, work=None, after=None, timer=None):
        self.work, self.after, self.timer = work, after, timer

    def wait(self):
        if self.work is not None:
            if self.timer is not None and self.timer[0].timing is not None:
                self.timer[0]._bracket(self.timer[1], self.work.wait)
            else:
                self.work.wait()
            self.work = None
        if self.after is not None:
            self.after()
            self.after = None

    def covered(self):
        """This exchange was issued BEFORE one on the same communication stream that has just been waited for: the stream runs its
        collectives in order, so it is complete as well - drop the handle without a second cross-stream wait (see wait_in_order)."""
        self.work = None
        if self.after is not None:
            self.after()
            self.after = None


def wait_in_order(*pendings):
    """Make the compute stream wait for several exchanges issued in this order on ONE communication stream: a single wait for the
    LAST one (in-order stream: the earlier ones are complete when it is).  One join instead of one per exchange - and the form a
    hipGraph capture survives: two event waits into the capture stream from different points of the communication stream segfault
    in hipStreamEndCapture on ROCm 7.0.2 / RCCL 2.26.6 (profiles/r06_cp_graph_capture_bisect.txt: q and k|v all-to-alls both in
    flight; one join, or one exchange in flight at a time, is fine)."""
    real = [p for p in pendings if p.work is not None]
    if real:
        real[-1].wait()
    for p in pendings:
        if p.work is not None:
            p.covered()
        else:
            p.wait()          # completed / simulated exchanges: run their follow-up


class ContextParallel(_ExchangeChoice):
    def __init__(self, group=None, exchange="auto", overlap=True, attn_kv_splits=1):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._gloo = dist.get_backend(group) == "gloo"
        self.exchange = exchange
        # overlap=True: the per-layer exchanges are issued asynchronously and waited for right before their consumer, so the
        # q all-to-all runs under the k|v projection (head exchange) and the K/V all-gather under the q projection (row
        # exchange); False = every collective completes before the next kernel is issued (A/B, debugging)
        self.overlap = overlap
        # > 1: a rank's dense self-attention launch (world x fewer workgroups than the unsharded one: 95 at 8 ranks) is cut along
        # the keys into this many ranges merged by a second kernel (rtv_attn_fwd_split).  Off by default: with one range the
        # sharded forward is bit-identical with the unsharded one (what the tests assert); with more it differs by fp32
        # summation order.
        self.attn_kv_splits = int(attn_kv_splits)
        self.timing = None        # start_timing(): list of (kind, start event, stop event)

    def shard(self, M):
        return shard_rows(M, self.world, self.rank)

    def local_ranks(self):
        return [self.rank]

    # ---- exposed-communication diagnostics (bench.py --gpus N)
    def start_timing(self):
        """From here on every point where the COMPUTE stream has to wait for a collective is bracketed with two events on that
        stream: the `Pending.wait()` of an asynchronous exchange (what the overlap did not hide) and the whole call of a
        synchronous one (`exchange_o`, the head-row gather: exposed by construction).  Eager forwards only - events inside a
        captured graph cannot be timed.  `read_timing()` -> {kind: (count, total ms)}."""
        self.timing = []

    def _bracket(self, kind, fn):
        if self.timing is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.timing.append((kind, e0, e1))
        return out

    def read_timing(self, stop=True):
        torch.cuda.synchronize()
        out = {}
        for kind, e0, e1 in self.timing or []:
            n, ms = out.get(kind, (0, 0.0))
            out[kind] = (n + 1, ms + e0.elapsed_time(e1))
        if stop:
            self.timing = None
        return out

    def all_gather_rows_(self, buf, async_op=False, kind="all_gather_rows"):
        """In-place all-gather along dim 0 of a contiguous buffer whose local shard is already filled.  async_op: returns a
        Pending (the collective runs on the communication stream; wait() orders the current stream behind it)."""
        if self.timing is not None and not async_op:
            return self._bracket(kind, lambda: self._all_gather_rows_(buf, False, kind))
        return self._all_gather_rows_(buf, async_op, kind)

    def _all_gather_rows_(self, buf, async_op, kind):
        done = Pending() if async_op else buf
        if self.world == 1 and not self.force_single_rank:
            return done
        if not buf.is_contiguous() or buf.shape[0] % self.world:
            raise ValueError("all_gather_rows_ needs a contiguous buffer with rows divisible by the world size")
        n = buf.shape[0] // self.world
        mine = buf[self.rank * n:(self.rank + 1) * n]
        if self._gloo and buf.is_cuda:
            # test-only route (several ranks sharing one GPU under gloo): stage through the host
            host = [torch.empty(mine.shape, dtype=buf.dtype) for _ in range(self.world)]
            dist.all_gather(host, mine.cpu(), group=self.group)
            for r in range(self.world):
                if r != self.rank:
                    buf[r * n:(r + 1) * n].copy_(host[r])
        elif self._gloo:
            work = dist.all_gather([buf[r * n:(r + 1) * n] for r in range(self.world)], mine.clone(), group=self.group,
                                   async_op=async_op and self.overlap)
            if async_op:
                return Pending(work if self.overlap else None, timer=(self, kind))
        else:
            work = dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=async_op and self.overlap)
            if async_op:
                return Pending(work if self.overlap else None, timer=(self, kind))
        return done

    def gather_kv(self, k, v, row0, M, async_op=False):
        """All-gather cache rows [row0, row0+M) of one layer's K and V ([kv_size, H, hd] views).  If K and V rows
        are interleaved in one arena ([kv_size, 2, H, hd]) this is ONE collective, otherwise two."""
        if self.world == 1 and not self.force_single_rank:
            return Pending() if async_op else None
        H, hd = k.shape[1], k.shape[2]
        row = H * hd
        inter = (k.stride(0) == 2 * row and v.stride(0) == 2 * row and k.stride(1) == hd and
                 v.data_ptr() == k.data_ptr() + row * k.element_size())
        if inter:
            both = torch.as_strided(k, (M, 2 * row), (2 * row, 1), k.storage_offset() + row0 * 2 * row)
            pend = self.all_gather_rows_(both, async_op, kind="gather_kv")
            return pend if async_op else None
        if k.stride(0) != row or v.stride(0) != row:
            raise ValueError("gather_kv needs dense or K/V-interleaved cache rows")
        p1 = self.all_gather_rows_(k[row0:row0 + M], async_op, kind="gather_kv")
        p2 = self.all_gather_rows_(v[row0:row0 + M], async_op, kind="gather_kv")
        return Pending(after=lambda: (p1.wait(), p2.wait())) if async_op else None

    # ---- head exchange
    def all_to_all_(self, out, inp, async_op=False, kind="all_to_all"):
        """out[g-th block] <- rank g's inp[rank-th block]; both contiguous with dim 0 divisible by the world size."""
        if self.timing is not None and not async_op:
            return self._bracket(kind, lambda: self._all_to_all_(out, inp, False, kind))
        return self._all_to_all_(out, inp, async_op, kind)

    def _all_to_all_(self, out, inp, async_op, kind):
        if not out.is_contiguous() or not inp.is_contiguous() or out.numel() != inp.numel() or out.numel() % self.world:
            raise ValueError("all_to_all_ needs contiguous, equally sized buffers divisible by the world size")
        work = None
        if self._gloo and inp.is_cuda:      # test-only route, as in all_gather_rows_
            host_out = torch.empty(inp.numel(), dtype=inp.dtype)
            dist.all_to_all_single(host_out, inp.reshape(-1).cpu(), group=self.group)
            out.view(-1).copy_(host_out)
        else:
            # Inside a hipGraph capture the all-to-all is issued SYNCHRONOUSLY: an asynchronous one (a forked branch of the capture
            # with the projection kernels beside it) segfaults in hipStreamEndCapture on ROCm 7.0.2 / RCCL 2.26.6, measured on a
            # one-rank group (profiles/r06_cp_graph_capture_bisect.txt); asynchronous all-gathers capture fine.
            go_async = async_op and self.overlap and not (inp.is_cuda and torch.cuda.is_current_stream_capturing())
            work = dist.all_to_all_single(out.view(-1), inp.view(-1), group=self.group, async_op=go_async)
            return Pending(work if go_async else None, timer=(self, kind)) if async_op else out
        return Pending(None) if async_op else out

    def exchange_q(self, parts, async_op=False):
        """q_send -> q_all: every rank receives ALL query rows of its own heads."""
        (rank, b), = parts
        return self.all_to_all_(b["q_all"], b["q_send"], async_op, kind="exchange_q")

    def exchange_kv(self, parts, k, v, row0, M, async_op=False):
        """kv_send -> rows [row0, row0+M) of this rank's heads of one layer's K/V cache (k, v: [kv_size, hn or H, hd] views)."""
        (rank, b), = parts
        hn, hd = b["hn"], k.shape[2]
        h0 = 0 if k.shape[1] == hn else rank * hn
        rows = _interleaved_rows(k, v, row0, M, h0, hn)
        if rows is not None:
            return self.all_to_all_(rows, b["kv_send"], async_op, kind="exchange_kv")            # straight into the cache arena
        tmp = torch.empty_like(b["kv_send"])
        pend = self.all_to_all_(tmp, b["kv_send"], async_op, kind="exchange_kv")

        def scatter():
            t = tmp.view(M, 2, hn, hd)
            k[row0:row0 + M, h0:h0 + hn] = t[:, 0]
            v[row0:row0 + M, h0:h0 + hn] = t[:, 1]
        if async_op:
            return Pending(after=lambda: (pend.wait(), scatter()))
        scatter()
        return None

    def exchange_qkv(self, parts, k, v, row0, M):
        """Both exchanges, completed (the non-overlapped form)."""
        self.exchange_q(parts)
        self.exchange_kv(parts, k, v, row0, M)

    def exchange_o(self, parts):
        (rank, b), = parts
        self.all_to_all_(b["o_recv"], b["o_all"], kind="exchange_o")


def _interleaved_rows(k, v, row0, M, h0, hn):
    """The [M, 2, hn*hd] view of cache rows [row0, row0+M) when K and V rows sit side by side in one arena that holds
    exactly this rank's heads (pipeline._initialize_kv_cache) - the all-to-all then receives in place - else None."""
    hd = k.shape[2]
    gc = hn * hd
    if (k.shape[1] != hn or h0 != 0 or k.stride(0) != 2 * gc or v.stride(0) != 2 * gc or k.stride(1) != hd or
            k.stride(2) != 1 or v.data_ptr() != k.data_ptr() + gc * k.element_size()):
        return None
    return torch.as_strided(k, (M, 2, gc), (2 * gc, gc, 1), k.storage_offset() + row0 * 2 * gc)


class SimulatedContextParallel(_ExchangeChoice):
    """Runs all `world` token shards inside ONE process in lockstep (test support on a single GPU): every shard's
    kernels write straight into the shared cache / head buffer, so the collectives are no-ops.  Exercises the
    sharded launch geometry (row offsets, per-frame lookups, cache row placement) of the phase API."""

    def __init__(self, world, exchange="auto", attn_kv_splits=1):
        self.world, self.rank = world, 0
        self.exchange = exchange
        self.attn_kv_splits = int(attn_kv_splits)
        self.timing = None

    def shard(self, M):
        return shard_rows(M, self.world, 0)

    def local_ranks(self):
        return list(range(self.world))

    def all_gather_rows_(self, buf, async_op=False, kind="all_gather_rows"):
        return Pending() if async_op else buf

    def gather_kv(self, k, v, row0, M, async_op=False):
        return Pending() if async_op else None

    def exchange_qkv(self, parts, k, v, row0, M):
        """All ranks live in this process and share one full-head cache: the all-to-alls become block copies."""
        rl = M // self.world
        for s, bs in parts:
            hn, hd = bs["hn"], k.shape[2]
            qs = bs["q_send"].view(self.world, rl, hn * hd)
            kvs = bs["kv_send"].view(self.world, rl, 2, hn, hd)
            for g, bg in parts:
                bg["q_all"].view(M, hn * hd)[s * rl:(s + 1) * rl] = qs[g]
                k[row0 + s * rl:row0 + (s + 1) * rl, g * hn:(g + 1) * hn] = kvs[g][:, 0]
                v[row0 + s * rl:row0 + (s + 1) * rl, g * hn:(g + 1) * hn] = kvs[g][:, 1]

    def exchange_q(self, parts, async_op=False):
        rl = parts[0][1]["q_all"].shape[0] // self.world
        for s, bs in parts:
            qs = bs["q_send"].view(self.world, rl, -1)
            for g, bg in parts:
                bg["q_all"][s * rl:(s + 1) * rl] = qs[g]
        return Pending() if async_op else None

    def exchange_kv(self, parts, k, v, row0, M, async_op=False):
        rl = M // self.world
        for s, bs in parts:
            hn, hd = bs["hn"], k.shape[2]
            kvs = bs["kv_send"].view(self.world, rl, 2, hn, hd)
            for g, bg in parts:
                k[row0 + s * rl:row0 + (s + 1) * rl, g * hn:(g + 1) * hn] = kvs[g][:, 0]
                v[row0 + s * rl:row0 + (s + 1) * rl, g * hn:(g + 1) * hn] = kvs[g][:, 1]
        return Pending() if async_op else None

    def exchange_o(self, parts):
        for s, bs in parts:
            M = bs["o_all"].shape[0]
            rl = M // self.world
            for g, bg in parts:
                bs["o_recv"].view(self.world, rl, -1)[g] = bg["o_all"][s * rl:(s + 1) * rl]


class ShardedVAEDecoder:
    """Spatially sharded streaming VAE decode for the context-parallel path ("VAE decode: shard by output
    rows with halo"): rank r decodes the r-th horizontal stripe of every frame (`rtv_vae_decode_rows`: stage 0 and the
    global mid-block attention replicated, stages 1-3 on row windows with conv halos) and ONE all-gather per block
    returns whole frames on every rank (they feed the first-frame re-encode and the frame callback).  Same call
    contract as VAEDecoderWrapper: `pixels, cache = decoder(latents, *cache)`; the stripes are bit-identical to the
    corresponding rows of the unsharded decode."""

    def __init__(self, cp, device="cuda"):
        from .vae_decoder import VAEDecoderWrapper
        self.cp = cp
        self.inner = VAEDecoderWrapper(device, row_shard=(cp.rank, cp.world))

    def load_state_dict(self, sd, strict=True):
        return self.inner.load_state_dict(sd, strict)

    def init_random_weights(self, seed=0):
        self.inner.init_random_weights(seed)
        return self

    def eval(self):
        return self

    def forward(self, z, *feat_cache):
        stripe, cache = self.inner(z, *feat_cache)          # [1, T', 3, rows, W]
        return gather_row_stripes(self.cp, stripe[0], 8 * z.shape[3]).unsqueeze(0), cache

    __call__ = forward


def gather_row_stripes(cp, stripe, H):
    """stripe: [T, C, rows_r, W] = rows [H*r//n, H*(r+1)//n) of a [T, C, H, W] image stack on rank r -> full stack."""
    n = cp.world
    T, C, _, W = stripe.shape
    bounds = [(H * r // n, H * (r + 1) // n) for r in range(n)]
    rows_max = max(b - a for a, b in bounds)
    buf = torch.zeros((n, T, C, rows_max, W), dtype=stripe.dtype, device=stripe.device)
    buf[cp.rank, :, :, :stripe.shape[2]] = stripe
    cp.all_gather_rows_(buf, kind="vae_pixel_rows")
    if all(b - a == rows_max for a, b in bounds):
        return buf.permute(1, 2, 0, 3, 4).reshape(T, C, H, W)
    return torch.cat([buf[r, :, :, :b - a] for r, (a, b) in enumerate(bounds)], dim=2)
