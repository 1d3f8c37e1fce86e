// DiT forward orchestrator: CausalWanModel._forward_inference
// (wan/modules/causal_model.py:825-954; block body :440-492; head :495-523; unpatchify :1126-1149).
// Pure launch sequencing over the kernels of this library on ONE stream: no allocation, no sync, so
// every entry is hipGraph-capturable.  13 launches per DiT layer.
//
// Token-axis (context-parallel) sharding: a rank owns token rows [row_begin, row_begin + row_count) of the
// M = F*gh*gw tokens.  Every per-token op runs on the local rows only; the single exchange per layer is the
// all-gather of the new K/V rows into the replicated KV cache, done by the host (RCCL via torch.distributed)
// between rtv_dit_layer_qkv and rtv_dit_layer_rest.  rtv_dit_forward = the unsharded composition.
//
// Head-parallel exchange (the *_hp phases; the reference's usp_attn_forward, xdit_context_parallel.py:149-190, does the
// same through xFuserLongContextAttention): tokens stay sharded for every per-token op, but around self-attention the
// shard flips to heads - each rank attends ALL M query rows for its H/world heads over a KV cache that holds only those
// heads.  Two all-to-alls per layer (q|k|v out, attention output back) move 4*(M/world)*d*(world-1)/world elements per
// rank instead of the all-gather's 2*M*d*(world-1)/world: world/2 times less xGMI traffic.
#include <atomic>

#include "rtv_common.h"
#include "rtv_internal.h"

namespace rtv {

__global__ void silu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = f32_to_bf16(silu(bf16_to_f32(x[i])));
}

struct Workspace {
  char* base;
  size_t off, cap;
  bool ok;
  void* take(size_t bytes) {
    size_t a = (off + 255) & ~(size_t)255;
    if (base && a + bytes > cap) ok = false;
    off = a + bytes;
    return base ? base + a : nullptr;
  }
};

struct DitBuffers {
  uint16_t *x, *xn, *qkv, *q, *ao, *h, *prow, *hrow;
  uint16_t *sinus, *te1, *e, *se, *e0, *emod, *ehead, *ctx1, *ctx, *ktmp;
  uint8_t* q8;      // fp8 path: the quantised activation of the linear being computed
  float* fscale;    // fp8 path: its per-tensor scale (device), followed by the 4-byte absmax scratch
  float* attn_part; // KV-split self-attention of a sharded call: fp32 partial outputs + (m, l) of splits x heads x rows <= H x M
  size_t attn_part_bytes;
};

// Buffers are sized for the full token count so that one workspace serves sharded and unsharded calls.
static size_t carve(const rtv_dit_config* c, int F, int gh, int gw, char* base, size_t cap, DitBuffers* b,
                    bool* ok) {
  Workspace ws{base, 0, cap, true};
  const size_t M = (size_t)F * gh * gw, d = c->dim, e = 2;
  DitBuffers t;
  t.x = (uint16_t*)ws.take(M * d * e);
  t.xn = (uint16_t*)ws.take(M * d * e);
  t.qkv = (uint16_t*)ws.take(M * 3 * d * e);
  t.q = (uint16_t*)ws.take(M * d * e);
  t.ao = (uint16_t*)ws.take(M * d * e);
  t.h = (uint16_t*)ws.take(M * (size_t)c->ffn_dim * e);
  t.prow = (uint16_t*)ws.take(M * (size_t)c->in_dim * 4 * e);
  t.hrow = (uint16_t*)ws.take(M * (size_t)c->out_dim * 4 * e);
  t.sinus = (uint16_t*)ws.take((size_t)F * c->freq_dim * e);
  t.te1 = (uint16_t*)ws.take((size_t)F * d * e);
  t.e = (uint16_t*)ws.take((size_t)F * d * e);
  t.se = (uint16_t*)ws.take((size_t)F * d * e);
  t.e0 = (uint16_t*)ws.take((size_t)F * 6 * d * e);
  t.emod = (uint16_t*)ws.take((size_t)c->num_layers * F * 6 * d * e);
  t.ehead = (uint16_t*)ws.take((size_t)F * 2 * d * e);
  t.ctx1 = (uint16_t*)ws.take((size_t)c->text_len * d * e);
  t.ctx = (uint16_t*)ws.take((size_t)c->text_len * d * e);
  t.ktmp = (uint16_t*)ws.take((size_t)c->text_len * d * e);
  {
    size_t rows = M > (size_t)c->text_len ? M : (size_t)c->text_len, k = d;
    if ((size_t)c->ffn_dim > k) k = c->ffn_dim;
    if ((size_t)c->text_dim > k) k = c->text_dim;
    if ((size_t)c->freq_dim > k) k = c->freq_dim;
    t.q8 = (uint8_t*)ws.take(c->use_fp8 ? rows * k : 0);
    t.fscale = (float*)ws.take(256);
  }
  // KV-split self-attention of sharded calls, only when the configuration asks for it (97 MB per workspace at 14B
  // that an unsharded forward never touches).  A sharded call attends rows x heads <= M x H / world; with kv_splits <= world the
  // partials of all splits fit M x H rows.
  t.attn_part_bytes = c->max_attn_kv_splits > 1 ? M * (size_t)c->num_heads * (128 + 2) * sizeof(float) : 0;
  t.attn_part = (float*)ws.take(t.attn_part_bytes);
  if (b) *b = t;
  if (ok) *ok = ws.ok;
  return ws.off;
}

struct Ctx {
  const rtv_dit_config* cfg;
  const rtv_dit_weights* w;
  const rtv_dit_step* st;
  DitBuffers b;
  rtv_stream_t stream;
  int d, H, L, ffn, hd, F, gh, gw, fs, M, r0, rc, tc;
};

static int make_ctx(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, void* workspace,
                    size_t workspace_bytes, rtv_stream_t stream, Ctx* c) {
  if (!cfg || !w || !st || !workspace) return set_error(-1, "dit: null argument");
  c->cfg = cfg;
  c->w = w;
  c->st = st;
  c->stream = stream;
  c->d = cfg->dim;
  c->H = cfg->num_heads;
  c->L = cfg->num_layers;
  c->ffn = cfg->ffn_dim;
  if (c->H <= 0 || c->d % c->H || c->d / c->H != 128) return set_error(-1, "dit: head_dim must be 128");
  c->hd = 128;
  c->F = st->F;
  c->gh = st->gh;
  c->gw = st->gw;
  c->fs = st->gh * st->gw;
  c->M = st->F * c->fs;
  if (c->M <= 0) return set_error(-1, "dit: empty token grid");
  c->r0 = st->row_begin;
  c->rc = st->row_count > 0 ? st->row_count : c->M - st->row_begin;
  if (c->r0 < 0 || c->rc <= 0 || c->r0 + c->rc > c->M) return set_error(-1, "dit: local row range outside the token grid");
  if (st->kv_lo < 0 || st->kv_hi <= st->kv_lo) return set_error(-1, "dit: empty attention window");
  if (st->cache_row0 < 0 || st->cache_row0 + c->M > st->kv_hi)
    return set_error(-1, "dit: the rows written by this call must lie inside the attention window");
  if (st->ring_size < 0 || st->ring_lo < 0 || st->ring_shift < 0 || (st->ring_size > 0 && st->ring_shift >= st->ring_size) ||
      (st->ring_size > 0 && st->kv_hi > st->ring_lo + st->ring_size))
    return set_error(-1, "dit: bad ring window (0 <= ring_shift < ring_size, kv_hi <= ring_lo + ring_size)");
  if (st->ring_size > 0 && st->ring_shift > 0 && st->causal_block > 0)
    return set_error(-1, "dit: the block-causal recompute pass writes an unrotated cache (ring_shift must be 0)");
  if (((uintptr_t)workspace) & 255) return set_error(-1, "dit: workspace must be 256-byte aligned");
  bool ok = true;
  carve(cfg, c->F, c->gh, c->gw, (char*)workspace, workspace_bytes, &c->b, &ok);
  if (!ok) return set_error(-1, "dit: workspace too small (see rtv_dit_workspace_bytes)");
  c->tc = st->gemm_tile_cfg;
  return 0;
}

}  // namespace rtv

using namespace rtv;

#define RTV_TRY(expr)        \
  do {                       \
    int _s = (expr);         \
    if (_s != 0) return _s;  \
  } while (0)

extern "C" int rtv_silu(const void* x, void* out, int64_t n, rtv_stream_t stream) {
  if (n <= 0) return 0;
  int blocks = (int)((n + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(silu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)out, n);
  return check_launch("silu");
}

extern "C" size_t rtv_dit_workspace_bytes(const rtv_dit_config* cfg, int F, int gh, int gw) {
  if (!cfg || F <= 0 || gh <= 0 || gw <= 0) return 0;
  return carve(cfg, F, gh, gw, nullptr, 0, nullptr, nullptr) + 256;
}

// nn.Linear on `M` rows with optional fused epilogue; `row_off` = global index of row 0 for the per-frame gate.
// fp8 mode (cfg->use_fp8, release_server.py:179-182): every nn.Linear quantises its input per tensor (dynamic) and
// multiplies e4m3 operands; `sidx` indexes the weight's per-tensor scale in rtv_dit_weights.fp8_scales (< 0: the op is
// not an nn.Linear - the Conv3d patch embedding - and stays bf16).
enum { S_TEXT0 = 0, S_TEXT2, S_TIME0, S_TIME2, S_TPROJ, S_HEAD, S_LAYER0 };
enum { S_QKV = 0, S_O, S_CQ, S_CK, S_CV, S_CO, S_FFN0, S_FFN2, S_PER_LAYER };
static int linear(Ctx& c, int sidx, const void* a, int K, const void* w, const void* bias, void* out, int M, int N, int act,
                  const void* gate, int gate_stride, int rpf, int row_off, const void* res, int cfg, rtv_stream_t s) {
  if (!c.cfg->use_fp8 || sidx < 0)
    return rtv_gemm(a, K, w, K, out, N, M, N, K, bias, act, gate, gate_stride, rpf, row_off, res, N, RTV_DTYPE_BF16, cfg, s);
  if (!c.w->fp8_scales) return set_error(-1, "dit: use_fp8 needs rtv_dit_weights.fp8_scales");
  // token-sharded calls quantise the rows this rank holds with their own scale - what a per-rank torchao linear does
  if (rtv_quantize_fp8(a, K, M, K, c.b.q8, K, c.b.fscale, c.b.fscale + 1, s)) return -1;
  return rtv_gemm_fp8(c.b.q8, K, w, K, c.b.fscale, c.w->fp8_scales[sidx], out, N, M, N, K, bias, act, gate, gate_stride, rpf,
                      row_off, res, N, s);
}

// ---- embeddings, modulation tables, cross-attention K/V (causal_model.py:874-902)
static int dit_begin(Ctx& c) {
  const rtv_dit_config* cfg = c.cfg;
  const rtv_dit_weights* w = c.w;
  const rtv_dit_step* st = c.st;
  DitBuffers& b = c.b;
  const int d = c.d, F = c.F, L = c.L, tc = c.tc;
  rtv_stream_t stream = c.stream;
  const int pk = cfg->in_dim * 4;
  RTV_TRY(rtv_patchify(st->x, b.prow, cfg->in_dim, F, c.gh, c.gw, stream));
  RTV_TRY(linear(c, -1, b.prow + (size_t)c.r0 * pk, pk, w->patch_w, w->patch_b, b.x, c.rc, d, 0, nullptr, 0, 0, 0, nullptr, tc, stream));
  RTV_TRY(rtv_sinusoidal_embedding(st->t, b.sinus, F, cfg->freq_dim, stream));
  RTV_TRY(linear(c, S_TIME0, b.sinus, cfg->freq_dim, w->time0_w, w->time0_b, b.te1, F, d, RTV_ACT_SILU, nullptr, 0, 0, 0, nullptr, tc, stream));
  RTV_TRY(linear(c, S_TIME2, b.te1, d, w->time2_w, w->time2_b, b.e, F, d, 0, nullptr, 0, 0, 0, nullptr, tc, stream));
  RTV_TRY(rtv_silu(b.e, b.se, (int64_t)F * d, stream));
  RTV_TRY(linear(c, S_TPROJ, b.se, d, w->tproj_w, w->tproj_b, b.e0, F, 6 * d, 0, nullptr, 0, 0, 0, nullptr, tc, stream));
  RTV_TRY(rtv_modulation_table(w->modulation, b.e0, b.emod, L, F, 6, 6, d, stream));
  RTV_TRY(rtv_modulation_table(w->head_modulation, b.e, b.ehead, 1, F, 2, 1, d, stream));
  // text context -> cross-attention K/V caches, only while they are not initialised (model.py:186-192;
  // the text MLP output is consumed nowhere else, causal_model.py:897-902).  Replicated on every rank.
  if (st->compute_cross_kv) {
    if (!st->context) return set_error(-1, "dit: context required to initialise the cross-attention cache");
    const int T = cfg->text_len;
    RTV_TRY(linear(c, S_TEXT0, st->context, cfg->text_dim, w->text0_w, w->text0_b, b.ctx1, T, d, RTV_ACT_GELU_TANH, nullptr, 0, 0, 0, nullptr, tc, stream));
    RTV_TRY(linear(c, S_TEXT2, b.ctx1, d, w->text2_w, w->text2_b, b.ctx, T, d, 0, nullptr, 0, 0, 0, nullptr, tc, stream));
    for (int l = 0; l < L; ++l) {
      const rtv_dit_layer_weights& lw = w->layers[l];
      RTV_TRY(linear(c, S_LAYER0 + S_PER_LAYER * l + S_CK, b.ctx, d, lw.ck_w, lw.ck_b, b.ktmp, T, d, 0, nullptr, 0, 0, 0, nullptr, tc, stream));
      RTV_TRY(rtv_rmsnorm(b.ktmp, d, st->ca_k[l], d, T, d, cfg->eps, lw.cnorm_k_w, stream));
      RTV_TRY(linear(c, S_LAYER0 + S_PER_LAYER * l + S_CV, b.ctx, d, lw.cv_w, lw.cv_b, st->ca_v[l], T, d, 0, nullptr, 0, 0, 0, nullptr, tc, stream));
    }
  }
  return 0;
}

static int hp_check(Ctx& c, int world);

// ---- layer, part 1: LN+modulate -> QKV -> RMSNorm(q,k)+RoPE -> local K/V rows into the cache (or the exchange buffers).
// `parts` selects what this call does, so that the host can put a collective between the two projections and let it run
// under the other one (RTV_PROJ_*): LN (the shared input of both), Q (columns [0, d) of the fused QKV weight), KV (columns
// [d, 3d)).  q_send / kv_send non-null = head-parallel exchange buffers (see below), null = local q + cache rows.
static std::atomic<int> g_direct_v{1};   // rtv_dit_set_direct_v(0): the V cache rows are copied by the RoPE / cache kernel (r04 form; A/B, tests)
static int dit_layer_proj(Ctx& c, int l, int parts, int world, void* q_send, void* kv_send) {
  const rtv_dit_layer_weights& lw = c.w->layers[l];
  const rtv_dit_step* st = c.st;
  DitBuffers& b = c.b;
  const int d = c.d;
  if (!(parts & (RTV_PROJ_LN | RTV_PROJ_Q | RTV_PROJ_KV))) return set_error(-1, "dit: empty projection phase");
  const uint16_t* em = b.emod + (size_t)l * c.F * 6 * d;  // [F][6][d]: shift_sa, scale_sa, gate_sa, shift_ffn, scale_ffn, gate_ffn
  if (parts & RTV_PROJ_LN)
    RTV_TRY(rtv_layernorm_modulate(b.x, b.xn, c.rc, d, c.cfg->eps, em + 0 * d, em + 1 * d, 6 * d, c.fs, c.r0, nullptr, nullptr, c.stream));
  const bool q = parts & RTV_PROJ_Q, kv = parts & RTV_PROJ_KV;
  if (!q && !kv) return 0;
  // the fused weight is [3d, d] row-major: a column range of the output is a row range of the weight
  const int n0 = q ? 0 : d, nn = (q ? d : 0) + (kv ? 2 * d : 0);
  const uint16_t* wq = (const uint16_t*)lw.qkv_w;
  const uint16_t* bq = (const uint16_t*)lw.qkv_b;
  const size_t welt = c.cfg->use_fp8 ? 1 : 2;   // fp8 weights are bytes
  bool v_in_place = false;
  {
    const void* wp = (const char*)lw.qkv_w + (size_t)n0 * d * welt;
    (void)wq;
    // r05: when this call's cache rows are ONE physical row range (always, unless a ring write wraps), the V third of the projection
    // is a GEMM of its own whose output matrix IS those rows of the V cache; the RoPE / cache kernel then moves a third less
    // (rope_parts | 4).  (One launch with a second output matrix was built first: three more kernel arguments, or one more epilogue
    // instantiation, cost the ping-pong kernel 3-4 % on the ffn-in shape - more than the copy.)
    void* v_dst = nullptr;
    if (kv && !kv_send && g_direct_v.load(std::memory_order_relaxed) && st->kv_v[l] && !(st->kv_row_stride & 7)) {
      int r0 = st->cache_row0 + c.r0;
      const int S = st->ring_lo, R = st->ring_size;
      bool one_range = true;
      if (R > 0 && r0 + c.rc > S) {
        if (r0 < S) one_range = false;                          // sink rows and ring rows in one call
        else {
          r0 = S + (r0 - S + st->ring_shift) % R;
          one_range = r0 + c.rc <= S + R;
        }
      }
      uint16_t* dst = (uint16_t*)st->kv_v[l] + (size_t)r0 * st->kv_row_stride;
      if (one_range && !((uintptr_t)dst & 15)) v_dst = dst;
    }
    const void* wv = (const char*)lw.qkv_w + (size_t)2 * d * d * welt;   // the V rows of the fused weight
    // Everything the RoPE / cache launch behind the GEMMs would refuse is refused HERE, before a GEMM has written V rows into the
    // cache (a failing call used to leave the V rows overwritten and the K rows untouched).
    {
      const int rp = (q ? 1 : 0) | (kv ? 2 : 0) | (v_dst ? 4 : 0);
      if (q_send || kv_send) {
        RTV_TRY(hp_check(c, world));
        const int gc = d / world;
        if ((q && !q_send) || (kv && !kv_send)) return set_error(-1, "dit: exchange buffers required");
        RTV_TRY(qk_norm_rope_check(2 * gc, 0, c.rc, d, c.H, c.F, c.gh, c.gw, st->start_frame, c.r0, gc, (int64_t)c.rc * gc,
                                   (int64_t)c.rc * 2 * gc, 0, 0, 0, rp));
      } else {
        RTV_TRY(qk_norm_rope_check(st->kv_row_stride, st->cache_row0, c.rc, d, c.H, c.F, c.gh, c.gw, st->start_frame, c.r0, 0, 0, 0,
                                   st->ring_lo, st->ring_size, st->ring_shift, rp));
      }
    }
    if (!c.cfg->use_fp8) {
      if (v_dst) {
        // two launches: columns [n0, 2d) into the projection buffer, the V third straight into the cache rows (Q | K at M = 4680 is 760
        // tiles = 2.97 rounds of the 256 CUs, V the o-projection's shape: +6 us on the fused launch's 563, against the 21 us the copy cost)
        if (2 * d > n0)
          RTV_TRY(rtv_gemm(b.xn, d, wp, d, b.qkv + n0, 3 * d, c.rc, 2 * d - n0, d, bq + n0, 0, nullptr, 0, 0, 0, nullptr, 0, RTV_DTYPE_BF16,
                           c.tc, c.stream));
        RTV_TRY(rtv_gemm(b.xn, d, wv, d, v_dst, (int)st->kv_row_stride, c.rc, d, d, bq + 2 * d, 0, nullptr, 0, 0, 0, nullptr, 0,
                         RTV_DTYPE_BF16, c.tc, c.stream));
        v_in_place = true;
      } else {
        RTV_TRY(rtv_gemm(b.xn, d, wp, d, b.qkv + n0, 3 * d, c.rc, nn, d, bq + n0, 0, nullptr, 0, 0, 0, nullptr, 0, RTV_DTYPE_BF16, c.tc,
                         c.stream));
      }
    } else {
      if (!c.w->fp8_scales) return set_error(-1, "dit: use_fp8 needs rtv_dit_weights.fp8_scales");
      if (rtv_quantize_fp8(b.xn, d, c.rc, d, b.q8, d, b.fscale, b.fscale + 1, c.stream)) return -1;
      const float ws = c.w->fp8_scales[S_LAYER0 + S_PER_LAYER * l + S_QKV];   // ONE scale for the fused weight, whichever rows a launch takes
      if (v_dst) {
        if (2 * d > n0)
          RTV_TRY(rtv_gemm_fp8(b.q8, d, wp, d, b.fscale, ws, b.qkv + n0, 3 * d, c.rc, 2 * d - n0, d, bq + n0, 0, nullptr, 0, 0, 0, nullptr, 0,
                               c.stream));
        RTV_TRY(rtv_gemm_fp8(b.q8, d, wv, d, b.fscale, ws, v_dst, (int)st->kv_row_stride, c.rc, d, d, bq + 2 * d, 0, nullptr, 0, 0, 0, nullptr,
                             0, c.stream));
        v_in_place = true;
      } else {
        RTV_TRY(rtv_gemm_fp8(b.q8, d, wp, d, b.fscale, ws, b.qkv + n0, 3 * d, c.rc, nn, d, bq + n0, 0, nullptr, 0, 0, 0, nullptr, 0, c.stream));
      }
    }
  }
  const int rope_parts = (q ? 1 : 0) | (kv ? 2 : 0) | (v_in_place ? 4 : 0);
  if (q_send || kv_send) {
    RTV_TRY(hp_check(c, world));
    const int gc = d / world;
    if ((q && !q_send) || (kv && !kv_send)) return set_error(-1, "dit: exchange buffers required");
    return qk_norm_rope_launch(b.qkv, q_send, kv_send, kv_send ? (void*)((uint16_t*)kv_send + gc) : nullptr, 2 * gc, 0, c.rc, d, c.H,
                               c.cfg->eps, lw.norm_q_w, lw.norm_k_w, c.w->rope_cs, c.F, c.gh, c.gw, st->start_frame, c.r0, gc,
                               (int64_t)c.rc * gc, (int64_t)c.rc * 2 * gc, 0, 0, 0, rope_parts, c.stream);
  }
  return qk_norm_rope_launch(b.qkv, b.q, st->kv_k[l], st->kv_v[l], st->kv_row_stride, st->cache_row0, c.rc, d, c.H, c.cfg->eps,
                             lw.norm_q_w, lw.norm_k_w, c.w->rope_cs, c.F, c.gh, c.gw, st->start_frame, c.r0, 0, 0, 0, st->ring_lo,
                             st->ring_size, st->ring_shift, rope_parts, c.stream);
}

static int dit_layer_qkv(Ctx& c, int l) { return dit_layer_proj(c, l, RTV_PROJ_LN | RTV_PROJ_Q | RTV_PROJ_KV, 0, nullptr, nullptr); }

static int dit_after_attn(Ctx& c, int l);

// Physical row ranges of the logical attention window [kv_lo, kv_hi) of a ring-indexed cache: at most two, because the
// sink rows [.., ring_lo) are adjacent to the ring's first physical row and the ring part wraps at most once.
struct KeyWindow {
  int row0, n0, row1, n1;
};
static KeyWindow key_window(const rtv_dit_step* st) {
  const int lo = st->kv_lo, hi = st->kv_hi, S = st->ring_lo, R = st->ring_size;
  if (R <= 0 || hi <= S) return KeyWindow{lo, hi - lo, 0, 0};
  const int rl = lo > S ? lo : S;                       // logical start of the ring part
  const int n = hi - rl;                                // ring rows in the window
  const int p0 = S + (rl - S + st->ring_shift) % R;     // its first physical row
  const int first = (p0 + n <= S + R) ? n : S + R - p0; // rows before the wrap
  const int sink = lo < S ? S - lo : 0;
  if (first == n) {                                     // no wrap: [lo, S) then [p0, p0 + n)
    if (sink == 0) return KeyWindow{p0, n, 0, 0};
    if (p0 == S) return KeyWindow{lo, sink + n, 0, 0};
    return KeyWindow{lo, sink, p0, n};
  }
  // wrapped: [p0, S + R) and [S, S + n - first); the sink rows are adjacent to the second piece
  return KeyWindow{sink ? lo : S, sink + n - first, p0, first};
}

// Self-attention of a sharded call: one launch, or the KV-split one when the step asks for it (an error when the workspace was
// not sized for it: never a silent unsplit run).
static int sharded_self_attn(Ctx& c, const void* q, const void* k, const void* v, void* o, int rows, int n0, int n1, int seg1_row,
                             int heads, int64_t q_rs, int64_t kv_rs, int64_t o_rs, int q_offset) {
  const rtv_dit_step* st = c.st;
  const float scale = 1.0f / sqrtf((float)c.hd);
  const int S = st->attn_kv_splits;
  const int qo = st->causal_block > 0 ? q_offset : 0;
  if (S > 1 && (S > c.cfg->max_attn_kv_splits || rtv_attn_split_workspace_bytes(1, rows, heads, S) > c.b.attn_part_bytes))
    return set_error(-1, "dit: attn_kv_splits exceeds rtv_dit_config.max_attn_kv_splits / the partials do not fit the workspace "
                         "(splits <= number of shards)");
  if (S > 1)
    return rtv_attn_fwd_split(q, k, v, o, 1, rows, n0, n1, seg1_row, heads, c.hd, 0, q_rs, 0, kv_rs, 0, kv_rs, 0, o_rs, scale,
                              st->causal_block, qo, S, c.b.attn_part, c.b.attn_part_bytes, RTV_DTYPE_BF16, c.stream);
  return rtv_attn_fwd_win(q, k, v, o, 1, rows, n0, n1, seg1_row, heads, c.hd, 0, q_rs, 0, kv_rs, 0, kv_rs, 0, o_rs, scale,
                          st->causal_block, qo, RTV_DTYPE_BF16, c.stream);
}

// ---- layer, part 2: attention over the cache window -> o-proj(+gate,+res) -> cross-attn -> FFN
static int dit_layer_rest(Ctx& c, int l) {
  const rtv_dit_step* st = c.st;
  DitBuffers& b = c.b;
  const int d = c.d, H = c.H, hd = c.hd, rc = c.rc, r0 = c.r0;
  rtv_stream_t stream = c.stream;
  const float scale = 1.0f / sqrtf((float)hd);
  const int64_t rs = st->kv_row_stride;
  const int Lkv = st->kv_hi - st->kv_lo;
  const int q_offset = st->cache_row0 - st->kv_lo + r0;  // position of local query row 0 inside the window
  uint16_t* kc = (uint16_t*)st->kv_k[l];
  uint16_t* vc = (uint16_t*)st->kv_v[l];
  // self attention (causal_model.py:386-390, :470-476)
  KeyWindow kw = key_window(st);
  if (rc < c.M) {   // token-sharded call: the query grid is rc / 256 tiles per head - optionally cut along the keys
    RTV_TRY(sharded_self_attn(c, b.q, kc + (size_t)kw.row0 * rs, vc + (size_t)kw.row0 * rs, b.ao, rc, kw.n0, kw.n1,
                              kw.row1 - kw.row0, H, d, rs, d, q_offset));
  } else {
    RTV_TRY(rtv_attn_fwd_win(b.q, kc + (size_t)kw.row0 * rs, vc + (size_t)kw.row0 * rs, b.ao, 1, rc, kw.n0, kw.n1,
                             kw.row1 - kw.row0, H, hd, 0, d, 0, rs, 0, rs, 0, d, scale, st->causal_block,
                             st->causal_block > 0 ? q_offset : 0, RTV_DTYPE_BF16, stream));
  }
  (void)Lkv;
  return dit_after_attn(c, l);
}

// ---- layer, part 3: b.ao (self-attention output of the local rows) -> o-proj(+gate,+res) -> cross-attn -> FFN
static int dit_after_attn(Ctx& c, int l) {
  const rtv_dit_layer_weights& lw = c.w->layers[l];
  const rtv_dit_step* st = c.st;
  DitBuffers& b = c.b;
  const int d = c.d, H = c.H, hd = c.hd, fs = c.fs, tc = c.tc, rc = c.rc, r0 = c.r0;
  rtv_stream_t stream = c.stream;
  const float eps = c.cfg->eps;
  const uint16_t* em = b.emod + (size_t)l * c.F * 6 * d;
  const float scale = 1.0f / sqrtf((float)hd);
  RTV_TRY(linear(c, S_LAYER0 + S_PER_LAYER * l + S_O, b.ao, d, lw.o_w, lw.o_b, b.x, rc, d, 0, em + 2 * d, 6 * d, fs, r0, b.x, tc, stream));
  // cross attention (causal_model.py:480, model.py:171-228)
  RTV_TRY(rtv_layernorm_modulate(b.x, b.xn, rc, d, eps, nullptr, nullptr, 0, 0, 0, lw.norm3_w, lw.norm3_b, stream));
  RTV_TRY(linear(c, S_LAYER0 + S_PER_LAYER * l + S_CQ, b.xn, d, lw.cq_w, lw.cq_b, b.qkv, rc, d, 0, nullptr, 0, 0, 0, nullptr, tc, stream));
  RTV_TRY(rtv_rmsnorm(b.qkv, d, b.q, d, rc, d, eps, lw.cnorm_q_w, stream));
  if (st->text_rows > 0 && st->text_rows + 1 < c.cfg->text_len)   // the padding rows share one K / V row: attend it once, weighted
    RTV_TRY(rtv_attn_fwd_dup(b.q, st->ca_k[l], st->ca_v[l], b.ao, 1, rc, st->text_rows + 1, H, hd, 0, d, 0, d, 0, d, 0, d, scale,
                             st->text_rows, c.cfg->text_len - st->text_rows, RTV_DTYPE_BF16, stream));
  else
    RTV_TRY(rtv_attn_fwd(b.q, st->ca_k[l], st->ca_v[l], b.ao, 1, rc, c.cfg->text_len, H, hd, 0, d, 0, d, 0, d, 0, d,
                         scale, 0, 0, RTV_DTYPE_BF16, stream));
  RTV_TRY(linear(c, S_LAYER0 + S_PER_LAYER * l + S_CO, b.ao, d, lw.co_w, lw.co_b, b.x, rc, d, 0, nullptr, 0, 0, 0, b.x, tc, stream));
  // FFN (causal_model.py:482-488)
  RTV_TRY(rtv_layernorm_modulate(b.x, b.xn, rc, d, eps, em + 3 * d, em + 4 * d, 6 * d, fs, r0, nullptr, nullptr, stream));
  RTV_TRY(linear(c, S_LAYER0 + S_PER_LAYER * l + S_FFN0, b.xn, d, lw.ffn0_w, lw.ffn0_b, b.h, rc, c.ffn, RTV_ACT_GELU_TANH, nullptr, 0, 0, 0, nullptr, tc, stream));
  RTV_TRY(linear(c, S_LAYER0 + S_PER_LAYER * l + S_FFN2, b.h, c.ffn, lw.ffn2_w, lw.ffn2_b, b.x, rc, d, 0, em + 5 * d, 6 * d, fs, r0, b.x, tc, stream));
  return 0;
}

// ---- head-parallel phases.  Exchange buffers are caller-owned (the host runs the all-to-alls on them):
//   q_send  [world][rc][gc]       gc = (H/world)*128 columns of one head group
//   kv_send [world][rc][2][gc]    K and V of a row adjacent, i.e. the row layout of a K/V-interleaved cache arena
//   q_all / o_all [M][gc]         all query rows / attention output for this rank's heads
//   o_recv  [world][rc][gc]       attention output of the local rows, one block per head group
// In these phases step->kv_k / kv_v point at THIS RANK'S heads ([kv_size][H/world][128], row stride kv_row_stride).
static int hp_check(Ctx& c, int world) {
  // (world == 1 is the degenerate one-rank exchange: bench.py --cp-host-probe runs the whole head-parallel path on a one-rank group)
  if (world < 1 || c.H % world) return set_error(-1, "dit: head-parallel exchange needs num_heads % world == 0");
  if (c.M % world || c.rc != c.M / world || c.r0 % c.rc)
    return set_error(-1, "dit: head-parallel exchange needs equal token shards (M % world == 0)");
  return 0;
}

static int dit_layer_qkv_hp(Ctx& c, int l, int world, void* q_send, void* kv_send) {
  return dit_layer_proj(c, l, RTV_PROJ_LN | RTV_PROJ_Q | RTV_PROJ_KV, world, q_send, kv_send);
}

static int dit_layer_attn_hp(Ctx& c, int l, int world, const void* q_all, void* o_all) {
  RTV_TRY(hp_check(c, world));
  const rtv_dit_step* st = c.st;
  const int gc = c.d / world, hn = c.H / world;
  const int64_t rs = st->kv_row_stride;
  const int Lkv = st->kv_hi - st->kv_lo;
  const int q_offset = st->cache_row0 - st->kv_lo;
  const uint16_t* kc = (const uint16_t*)st->kv_k[l];
  const uint16_t* vc = (const uint16_t*)st->kv_v[l];
  KeyWindow kw = key_window(st);
  (void)Lkv;
  return sharded_self_attn(c, q_all, kc + (size_t)kw.row0 * rs, vc + (size_t)kw.row0 * rs, o_all, c.M, kw.n0, kw.n1,
                           kw.row1 - kw.row0, hn, gc, rs, gc, q_offset);
}

static int dit_layer_rest_hp(Ctx& c, int l, int world, const void* o_recv) {
  RTV_TRY(hp_check(c, world));
  RTV_TRY(regroup_heads(o_recv, c.b.ao, c.rc, world, c.d / world, c.stream));
  return dit_after_attn(c, l);
}

// ---- head on local rows (causal_model.py:512-523) -> head_rows[row_begin : row_begin+row_count)
static int dit_head(Ctx& c, void* head_rows) {
  DitBuffers& b = c.b;
  const int d = c.d, n = c.cfg->out_dim * 4;
  RTV_TRY(rtv_layernorm_modulate(b.x, b.xn, c.rc, d, c.cfg->eps, b.ehead + 0 * d, b.ehead + 1 * d, 2 * d, c.fs, c.r0, nullptr, nullptr, c.stream));
  RTV_TRY(linear(c, S_HEAD, b.xn, d, c.w->head_w, c.w->head_b, (uint16_t*)head_rows + (size_t)c.r0 * n, c.rc, n, 0, nullptr, 0, 0, 0, nullptr, c.tc, c.stream));
  return 0;
}

extern "C" int rtv_dit_begin(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, void* ws,
                             size_t ws_bytes, rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, ws, ws_bytes, stream, &c));
  return dit_begin(c);
}

extern "C" int rtv_dit_layer_qkv(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, int layer,
                                 void* ws, size_t ws_bytes, rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, ws, ws_bytes, stream, &c));
  if (layer < 0 || layer >= c.L) return set_error(-1, "dit: layer out of range");
  return dit_layer_qkv(c, layer);
}

extern "C" int rtv_dit_layer_proj(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, int layer,
                                  int parts, int world, void* q_send, void* kv_send, void* ws, size_t ws_bytes,
                                  rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, ws, ws_bytes, stream, &c));
  if (layer < 0 || layer >= c.L) return set_error(-1, "dit: layer out of range");
  return dit_layer_proj(c, layer, parts, world, q_send, kv_send);
}

extern "C" int rtv_dit_layer_rest(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, int layer,
                                  void* ws, size_t ws_bytes, rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, ws, ws_bytes, stream, &c));
  if (layer < 0 || layer >= c.L) return set_error(-1, "dit: layer out of range");
  return dit_layer_rest(c, layer);
}

extern "C" int rtv_dit_layer_qkv_hp(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, int layer,
                                    int world, void* q_send, void* kv_send, void* ws, size_t ws_bytes, rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, ws, ws_bytes, stream, &c));
  if (layer < 0 || layer >= c.L) return set_error(-1, "dit: layer out of range");
  if (!q_send || !kv_send) return set_error(-1, "dit: exchange buffers required");
  return dit_layer_qkv_hp(c, layer, world, q_send, kv_send);
}

extern "C" int rtv_dit_layer_attn_hp(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, int layer,
                                     int world, const void* q_all, void* o_all, void* ws, size_t ws_bytes,
                                     rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, ws, ws_bytes, stream, &c));
  if (layer < 0 || layer >= c.L) return set_error(-1, "dit: layer out of range");
  if (!q_all || !o_all) return set_error(-1, "dit: exchange buffers required");
  return dit_layer_attn_hp(c, layer, world, q_all, o_all);
}

extern "C" int rtv_dit_layer_rest_hp(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, int layer,
                                     int world, const void* o_recv, void* ws, size_t ws_bytes, rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, ws, ws_bytes, stream, &c));
  if (layer < 0 || layer >= c.L) return set_error(-1, "dit: layer out of range");
  if (!o_recv) return set_error(-1, "dit: exchange buffers required");
  return dit_layer_rest_hp(c, layer, world, o_recv);
}

extern "C" int rtv_dit_head(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st, void* head_rows,
                            void* ws, size_t ws_bytes, rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, ws, ws_bytes, stream, &c));
  if (!head_rows) return set_error(-1, "dit: head_rows required");
  return dit_head(c, head_rows);
}

extern "C" int rtv_dit_finish(const rtv_dit_config* cfg, const rtv_dit_step* st, const void* head_rows,
                              rtv_stream_t stream) {
  if (!cfg || !st || !head_rows || !st->out) return set_error(-1, "dit: null argument");
  return rtv_unpatchify(head_rows, st->out, cfg->out_dim, st->F, st->gh, st->gw, stream);
}

extern "C" int rtv_dit_forward(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* st,
                               void* workspace, size_t workspace_bytes, rtv_stream_t stream) {
  Ctx c;
  RTV_TRY(make_ctx(cfg, w, st, workspace, workspace_bytes, stream, &c));
  if (c.r0 != 0 || c.rc != c.M)
    return set_error(-1, "dit_forward: sharded row ranges need the phase API (rtv_dit_begin/layer_qkv/layer_rest/head/finish)");
  RTV_TRY(dit_begin(c));
  for (int l = 0; l < c.L; ++l) {
    if (st->kv_only && l == c.L - 1)   // nothing downstream of the last layer's K / V rows is consumed
      return dit_layer_proj(c, l, RTV_PROJ_LN | RTV_PROJ_KV, 0, nullptr, nullptr);
    RTV_TRY(dit_layer_qkv(c, l));
    RTV_TRY(dit_layer_rest(c, l));
  }
  RTV_TRY(dit_head(c, c.b.hrow));
  return rtv_unpatchify(c.b.hrow, st->out, cfg->out_dim, c.F, c.gh, c.gw, stream);
}

extern "C" int rtv_dit_set_direct_v(int on) {
  g_direct_v = on ? 1 : 0;
  return 0;
}
