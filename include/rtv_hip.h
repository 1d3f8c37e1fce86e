/*
 * rtv_hip.h — C ABI of librtv_hip.so: the MI355X (gfx950) kernels behind the real-time
 * autoregressive video-diffusion hot path of krea-ai/realtime-video.
 *
 * Every entry point takes plain device pointers, sizes and a HIP stream (as void*); nothing here
 * depends on PyTorch.  All functions return 0 on success, a positive hipError_t or a negative
 * argument-error code otherwise; rtv_last_error() gives the message (the Python binding raises
 * RuntimeError with it, mirroring the Python-exception error contract of the reference's
 * attention()/pipeline API).  Kernels are enqueued on `stream` and never synchronise.
 *
 * Each group cites the reference interface (file:line under the upstream repo) it replaces.
 * 16-bit tensors are raw bf16 (RTV_DTYPE_BF16) unless a dtype argument says otherwise.
 */
#ifndef RTV_HIP_H
#define RTV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTV_DTYPE_BF16 0
#define RTV_DTYPE_F16 1

#define RTV_ACT_NONE 0
#define RTV_ACT_GELU_TANH 1
#define RTV_ACT_SILU 2

typedef void* rtv_stream_t; /* hipStream_t */

/* ---- library ---------------------------------------------------------------------------- */
/* ABI revision of this header: bumped whenever a struct layout or a signature below changes (101: rtv_dit_config gained the
 * trailing max_attn_kv_splits; 102: rtv_attn_set_waves value 3, r05; 103: rtv_dispatch_* added, r06).  A binding compares rtv_version() with the
 * RTV_ABI_VERSION it was written against and refuses a mismatch (realtime_video_amd/_lib.py does). */
#define RTV_ABI_VERSION 103
int rtv_version(void);
const char* rtv_last_error(void);

/* Per-kernel-class hipEvent timing (used by bench.py's roofline block).
 * class ids: 0 gemm, 1 attention, 2 layernorm/modulate, 3 rmsnorm+rope+cache, 4 conv, 5 misc.
 * rtv_prof_enable(mask): bit c set = bracket every launch of class c with two events (0 = off, 0x3f = all).  The event
 * pairs serialise neighbouring launches, so a timed run enables only the class it reports. */
int rtv_prof_enable(int class_mask);
int rtv_prof_read(int cls, double* total_ms, int64_t* launches, double* total_work);
int rtv_prof_reset(void);
/* Sampled bracketing: an event pair costs the launch stream a few microseconds, which matters for 20 us kernels and for thousands
 * of launches per block.  rtv_prof_set_stride(cls, n): bracket every n-th launch of class cls (default 1 = every launch);
 * rtv_prof_read_seen: ALL launches of the (enabled) class and their summed work since the last reset - the sampled time of
 * rtv_prof_read scales to the class by seen_work / sampled_work. */
int rtv_prof_set_stride(int cls, int stride);
int rtv_prof_read_seen(int cls, int64_t* launches, double* work);
/* What a bracket reads WITHOUT a kernel inside: n back-to-back (start, stop) event pairs on `stream`, their average elapsed time
 * in ms.  A bracketed launch reads kernel time + this (the two markers are processed by the queue on either side of the dispatch);
 * bench.py subtracts it per bracketed launch from the class times.  Synchronises the stream. */
int rtv_prof_bracket_overhead(int n, rtv_stream_t stream, double* avg_ms);

/* Which kernel VARIANT the dispatch rules chose, counted per launch since load / the last reset (one relaxed atomic per launch):
 * rtv_dispatch_counts fills counts[0 .. min(n, K)) and returns K, the number of variants; rtv_dispatch_name(i) is variant i's
 * name ("gemm8_kernel<256x256 ping-pong>", "attn_fwd_w4_kernel<one wave per SIMD>", ...).  bench.py records the variants that ran
 * in its JSON line, so that a shape silently falling back to an older kernel shows in the driver's numbers. */
int rtv_dispatch_counts(int64_t* counts, int n);
const char* rtv_dispatch_name(int id);
int rtv_dispatch_reset(void);

/* ---- K1/K2/K3: attention backend --------------------------------------------------------
 * Replaces wan/modules/attention.py:150-212 `attention(q,k,v,...)` (and the sage custom op
 * wan/modules/sage.py:12-19, flex_attention call causal_model.py:339-348, cross-attention
 * model.py:201-223).  Layout BLHD, head_dim 128.  q:[B,Lq,H,128] k,v:[B,Lkv,H,128] (k/v may be
 * strided views of the KV cache: strides in elements), o:[B,Lq,H,128].
 * softmax(scale * q k^T) v, no dropout.  Masking: if causal_block > 0, query row i attends keys
 * j < ((q_offset + i) / causal_block + 1) * causal_block (block-causal prefix rule of
 * causal_model.py:134-136; the diagonal term is implied); causal_block == 0 -> dense. */
int rtv_attn_fwd(const void* q, const void* k, const void* v, void* o,
                 int B, int Lq, int Lkv, int H, int D,
                 int64_t q_batch_stride, int64_t q_row_stride,
                 int64_t k_batch_stride, int64_t k_row_stride,
                 int64_t v_batch_stride, int64_t v_row_stride,
                 int64_t o_batch_stride, int64_t o_row_stride,
                 float scale, int causal_block, int q_offset, int dtype, rtv_stream_t stream);

/* The same over a key window made of TWO row ranges of the cache: keys [0, Lkv0) are rows 0.. of k / v, keys
 * [Lkv0, Lkv0 + Lkv1) are rows seg1_row.. (relative to k / v, may be negative).  This is how the rolling KV cache of
 * causal_model.py:363-379 is attended once it is kept as a ring instead of being shifted: the window [sink | ring] is at most
 * two physical ranges, and softmax attention does not depend on the key order.  Lkv1 == 0 is rtv_attn_fwd; the block-causal
 * mask needs Lkv1 == 0. */
int rtv_attn_fwd_win(const void* q, const void* k, const void* v, void* o,
                     int B, int Lq, int Lkv0, int Lkv1, int seg1_row, int H, int D,
                     int64_t q_batch_stride, int64_t q_row_stride,
                     int64_t k_batch_stride, int64_t k_row_stride,
                     int64_t v_batch_stride, int64_t v_row_stride,
                     int64_t o_batch_stride, int64_t o_row_stride,
                     float scale, int causal_block, int q_offset, int dtype, rtv_stream_t stream);
/* Dense attention in which key `dup_key` stands for `dup_count` IDENTICAL keys: softmax attention over a window that holds n
 * copies of one (k, v) row equals attention over one copy whose score gets + log(n).  This is the text cross-attention of
 * model.py:171-228 without its redundant work: the prompt embedding is zero-padded to 512 rows (utils/wan_wrapper.py:52-53), the
 * text MLP, the k / v projections and the k-norm map every padding row to the same K and V row, so the window is the real rows
 * plus ONE padding row counted 512 - n_real times (mathematically identical; fp32 summation order differs). */
int rtv_attn_fwd_dup(const void* q, const void* k, const void* v, void* o,
                     int B, int Lq, int Lkv, int H, int D,
                     int64_t q_batch_stride, int64_t q_row_stride,
                     int64_t k_batch_stride, int64_t k_row_stride,
                     int64_t v_batch_stride, int64_t v_row_stride,
                     int64_t o_batch_stride, int64_t o_row_stride,
                     float scale, int dup_key, int dup_count, int dtype, rtv_stream_t stream);
/* KV split of rtv_attn_fwd_win for launches whose query grid cannot fill 256 CUs (the head-parallel phase of a context-parallel
 * rank: 5 heads x 19 query tiles = 95 workgroups; wan/distributed/xdit_context_parallel.py:179 runs that attention through
 * xfuser's USP): the key window is cut into kv_splits ranges of 64-key tiles, one workgroup per (head, query tile, range); each
 * leaves its unnormalised fp32 output and its (reference point, row sum) in `workspace`, a second kernel merges them
 * (m = max m_s, O = sum 2^(m_s - m) O_s / sum 2^(m_s - m) l_s) and writes `o`.  Arguments as rtv_attn_fwd_win (a block-causal
 * launch splits every workgroup's own key-tile count).
 * Same mathematics as the unsplit launch; the fp32 summation order differs (stated tolerance in the tests: the split result is
 * within 2 bf16 ulps of the unsplit one).  workspace: rtv_attn_split_workspace_bytes(B, Lq, H, kv_splits), 16-byte aligned. */
size_t rtv_attn_split_workspace_bytes(int B, int Lq, int H, int kv_splits);
int rtv_attn_fwd_split(const void* q, const void* k, const void* v, void* o,
                       int B, int Lq, int Lkv0, int Lkv1, int seg1_row, int H, int D,
                       int64_t q_batch_stride, int64_t q_row_stride,
                       int64_t k_batch_stride, int64_t k_row_stride,
                       int64_t v_batch_stride, int64_t v_row_stride,
                       int64_t o_batch_stride, int64_t o_row_stride,
                       float scale, int causal_block, int q_offset,
                       int kv_splits, void* workspace, size_t workspace_bytes, int dtype, rtv_stream_t stream);

/* ---- K4: projection GEMM with fused epilogue ---------------------------------------------
 * C[M,N] = epi(A[M,K] @ W[N,K]^T): replaces nn.Linear (causal_model.py:196-199,:246,:433-435,
 * :614-623; model.py:184-198) plus the eager chains bias -> GELU(tanh) (:434), y*e[2]+x (:476),
 * y*e[5]+x (:487-488), x + cross_attn (:480).
 *   bias[N] (nullable); act RTV_ACT_*; gate (nullable): gate[((row_offset + m) / rows_per_frame)*gate_stride + n]
 *   (row_offset = global index of local row 0 when the token axis is sharded across GPUs);
 *   residual[M,ldr] (nullable, may alias C).  tile_cfg 0 = default. */
int rtv_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc,
             int M, int N, int K,
             const void* bias, int act,
             const void* gate, int gate_stride, int rows_per_frame, int row_offset,
             const void* residual, int ldr,
             int dtype, int tile_cfg, rtv_stream_t stream);

/* tile_cfg: 0 = default (gemm8 + split-K when the problem fills the chip with 256x256 tiles, else 128x128);
 * 1 = 128x128 tiles (2 workgroups per CU); 2/3 = 256x128 / 256x256 simple double buffer;
 * 4 = 256x256 ping-pong pipeline (gemm8.hip); 5 = 4 + split-K of the last partial round of tiles (needs a
 * workspace, below); 6 / 7 = the 128x256 ping-pong variant for few-row problems without / with that split-K.
 * The split-K partial sums live in a caller-owned fp32 workspace.  A workspace must not be used by two launches that can
 * overlap in time, so it is attached per (device, stream): rtv_gemm_set_stream_workspace(stream, ...) for launches on that
 * stream of the CURRENT device; rtv_gemm_set_workspace(...) attaches a device-wide default used by the streams that have none
 * of their own (correct while one stream per device issues split-K GEMMs).  Without a workspace the GEMM runs unsplit. */
size_t rtv_gemm_workspace_bytes(void);
int rtv_gemm_set_workspace(void* ptr, size_t bytes);   /* ptr == NULL detaches it.  Zeroes the arrival counters with a
                                                          synchronous hipMemset (call it at set-up time, not under graph
                                                          capture); launches leave the counters at zero. */
int rtv_gemm_set_stream_workspace(rtv_stream_t stream, void* ptr, size_t bytes);

/* ---- K5: fused norm / modulation / RoPE / KV-cache write ----------------------------------
 * rtv_layernorm_modulate: out = LN(x; eps, no affine) * (1 + scale[f]) + shift[f], f = (row_offset + m) / rows_per_frame
 *   (causal_model.py:471, :483-484, :522; WanLayerNorm model.py:88-98).  shift/scale nullable ->
 *   plain LN; weight/bias (nullable) -> affine LN (norm3, causal_model.py:424-426,:480).
 *   shift and scale point at frame 0's [d] vectors; frame_stride elements between frames. */
int rtv_layernorm_modulate(const void* x, void* out, int M, int d, float eps,
                           const void* shift, const void* scale, int frame_stride, int rows_per_frame,
                           int row_offset, const void* weight, const void* bias, rtv_stream_t stream);

/* rtv_rmsnorm: out = bf16(x * rsqrt(mean(x^2)+eps)) * weight over the full channel dim
 *   (WanRMSNorm model.py:69-85; used for cross-attention q/k, model.py:184,:189). */
int rtv_rmsnorm(const void* x, int ldx, void* out, int ldo, int M, int d, float eps,
                const void* weight, rtv_stream_t stream);

/* rtv_qk_norm_rope_cache: one pass over the fused QKV projection output qkv[M,3d]:
 *   q = rope(rmsnorm(q)*wq) -> q_out[M,d];  k = rope(rmsnorm(k)*wk) -> k_cache rows [cache_row0, cache_row0+M);
 *   v -> v_cache rows likewise.  Replaces causal_model.py:243-256 (norm), :143-171 / model.py:39-66
 *   (3-axis RoPE; rope_cs is float2 [1024][hd/2] = (cos,sin) of rope_params model.py:28-35 laid out as
 *   causal_model.py:639-645), :380-385 / :309-311 (cache write).  Token m -> (f,h,w) on grid (F,gh,gw),
 *   temporal position start_frame + f.  Cache row stride in elements.  With a sharded token axis the call
 *   covers local rows only: M rows starting at global token row_offset (cache rows cache_row0 + row_offset + m). */
int rtv_qk_norm_rope_cache(const void* qkv, void* q_out, void* k_cache, void* v_cache,
                           int64_t cache_row_stride, int cache_row0,
                           int M, int d, int num_heads, float eps,
                           const void* wq, const void* wk, const void* rope_cs,
                           int F, int gh, int gw, int start_frame, int row_offset, rtv_stream_t stream);
/* ... with the cache kept as a ring (replaces the eviction shift copy of causal_model.py:363-379): logical cache
 * row r >= ring_lo is stored at ring_lo + (r - ring_lo + ring_shift) % ring_size; rows below ring_lo (attention sink) do not
 * move.  ring_size == 0 is the plain call. */
int rtv_qk_norm_rope_cache_ring(const void* qkv, void* q_out, void* k_cache, void* v_cache,
                                int64_t cache_row_stride, int cache_row0,
                                int M, int d, int num_heads, float eps,
                                const void* wq, const void* wk, const void* rope_cs,
                                int F, int gh, int gw, int start_frame, int row_offset,
                                int ring_lo, int ring_size, int ring_shift, rtv_stream_t stream);

/* rtv_modulation_table: emod[l][f][j][:] = bf16(modulation[l][j][:] + e0[f][j][:]) for l<L, j<J
 *   (causal_model.py:466, :521).  modulation:[L][J][d], e0:[F][J0][d] with J0 = J or 1 (broadcast). */
int rtv_modulation_table(const void* modulation, const void* e0, void* emod,
                         int L, int F, int J, int J0, int d, rtv_stream_t stream);

/* rtv_sinusoidal_embedding: out[F][dim] = bf16([cos(t*w_i), sin(t*w_i)]) computed in float64
 *   (model.py:15-24), t: float32 [F]. */
int rtv_sinusoidal_embedding(const void* t, void* out, int F, int dim, rtv_stream_t stream);

/* rtv_patchify / rtv_unpatchify: latent [C,F,2gh,2gw] <-> token rows [F*gh*gw][C*4]
 *   (Conv3d k=s=(1,2,2) im2col order (c,p,q), causal_model.py:614-615,:874-877; unpatchify
 *   einsum 'fhwpqrc->cfphqwr', causal_model.py:1126-1149, column order (q,r,c)). */
int rtv_patchify(const void* x, void* rows, int C, int F, int gh, int gw, rtv_stream_t stream);
int rtv_unpatchify(const void* rows, void* x, int C, int F, int gh, int gw, rtv_stream_t stream);

/* rtv_scheduler_step: the flow-matching arithmetic of one denoising step in one launch, bit-exact with the reference's
 *   eager chains on bf16 latents [F][C][hw]:
 *     x0    = bf16(float(double(xt) - double(sigma[i]) * double(flow))),  i = argmin |double(timesteps) - t[f]|
 *             (WanDiffusionWrapper._convert_flow_pred_to_x0, utils/wan_wrapper.py:181-205);
 *     noisy = bf16((1 - sigma[j]) * float(x0) + sigma[j] * float(noise)),  j = argmin |timesteps - t_next[f]| in
 *             float32 (FlowMatchScheduler.add_noise, utils/scheduler.py:159-176).
 *   flow / xt are read through (frame, channel) element strides with contiguous hw (the model's [C][F][hw] output needs
 *   no permute copy); x0 / noise / noisy are contiguous.  flow == NULL: x0 is an input (add_noise alone).  t_next == NULL:
 *   no re-noising (last step).  t / t_next: [F] of t_kind 0 float32, 1 float64, 2 int64.  timesteps / sigmas: float32
 *   [n_table] on the device. */
int rtv_scheduler_step(const void* flow, int64_t flow_frame_stride, int64_t flow_channel_stride,
                       const void* xt, int64_t xt_frame_stride, int64_t xt_channel_stride,
                       const void* t, const void* t_next, int t_kind,
                       const void* timesteps, const void* sigmas, int n_table,
                       void* x0, const void* noise, void* noisy,
                       int F, int C, int hw, rtv_stream_t stream);

/* rtv_silu: out = bf16(silu(x)) elementwise (time_projection's leading nn.SiLU, causal_model.py:622-623). */
int rtv_silu(const void* x, void* out, int64_t n, rtv_stream_t stream);

/* ---- whole-forward orchestrator: one call = CausalWanModel._forward_inference ------------------
 * (wan/modules/causal_model.py:825-954 with the block body :440-492, head :495-523, unpatchify
 * :1126-1149).  The host keeps the reference's python-int cache bookkeeping (global_end_index /
 * local_end_index, causal_model.py:358-392) and passes the resulting row window; the library owns
 * no memory: weights, caches and the workspace are caller allocations (torch tensors). */
typedef struct rtv_dit_config {
  int dim, ffn_dim, num_heads, num_layers;
  int freq_dim, text_dim, text_len;
  int in_dim, out_dim;          /* latent channels (16, 16); patch size is (1,2,2) */
  float eps;
  int use_fp8;                  /* 1: every nn.Linear runs the e4m3 path (weights below are e4m3, fp8_scales set) */
  int max_attn_kv_splits;       /* largest rtv_dit_step.attn_kv_splits this workspace must serve (0 / 1: none - the workspace then
                                   holds no split-attention partials: 97 MB at 14B, M = 4680); a step asking for more fails */
} rtv_dit_config;

typedef struct rtv_dit_layer_weights { /* bf16 device pointers, reference state_dict names in comments */
  const void *qkv_w, *qkv_b;          /* self_attn.{q,k,v} fused [3d,d] (fuse_projections, causal_model.py:203-216) */
  const void *norm_q_w, *norm_k_w;    /* self_attn.norm_q / norm_k .weight [d] */
  const void *o_w, *o_b;              /* self_attn.o */
  const void *norm3_w, *norm3_b;      /* norm3 (affine LayerNorm) */
  const void *cq_w, *cq_b, *ck_w, *ck_b, *cv_w, *cv_b, *co_w, *co_b; /* cross_attn.{q,k,v,o} */
  const void *cnorm_q_w, *cnorm_k_w;  /* cross_attn.norm_q / norm_k */
  const void *ffn0_w, *ffn0_b, *ffn2_w, *ffn2_b; /* ffn.0 / ffn.2 */
} rtv_dit_layer_weights;

typedef struct rtv_dit_weights {
  const void *patch_w, *patch_b;      /* patch_embedding [d, in_dim*4] */
  const void *text0_w, *text0_b, *text2_w, *text2_b;
  const void *time0_w, *time0_b, *time2_w, *time2_b;
  const void *tproj_w, *tproj_b;      /* time_projection.1 [6d, d] */
  const void *head_w, *head_b;        /* head.head [out_dim*4, d] */
  const void *modulation;             /* blocks.*.modulation packed [L][6][d] */
  const void *head_modulation;        /* head.modulation [2][d] */
  const void *rope_cs;                /* float2 [1024][head_dim/2] */
  const rtv_dit_layer_weights* layers; /* host array [num_layers] */
  const float* fp8_scales;            /* fp8 mode: host array of per-tensor weight scales: text0, text2, time0, time2, tproj,
                                         head, then per layer qkv, o, cq, ck, cv, co, ffn0, ffn2 (6 + 8 L entries); the *_w
                                         pointers of those linears then address e4m3 [N][K] data.  NULL otherwise */
} rtv_dit_weights;

typedef struct rtv_dit_step {
  const void* x;            /* latent [in_dim, F, 2gh, 2gw] bf16 */
  const void* t;            /* float32 [F] timesteps */
  const void* context;      /* [text_len, text_dim] bf16 zero-padded prompt embedding; used iff compute_cross_kv */
  void* out;                /* flow prediction [out_dim, F, 2gh, 2gw] bf16 */
  int F, gh, gw;            /* token grid: M = F*gh*gw */
  void* const* kv_k;        /* host arrays [num_layers] of device pointers: KV cache [kv_size][H][hd] */
  void* const* kv_v;
  int64_t kv_row_stride;    /* elements between cache rows (H*hd when contiguous) */
  void* const* ca_k;        /* cross-attention caches [text_len][H][hd] */
  void* const* ca_v;
  int compute_cross_kv;     /* crossattn_cache["is_init"] == False (model.py:186-192) */
  int cache_row0;           /* local_start_index: first cache row written by this call */
  int kv_lo, kv_hi;         /* attention window over cache rows [kv_lo, kv_hi) (causal_model.py:386-390) */
  int start_frame;          /* RoPE temporal offset (current_start // 1560, :351-356; 0 for the recompute pass) */
  int causal_block;         /* 0 = dense; >0 = block-causal recompute pass (tokens per block, :305-348) */
  int gemm_tile_cfg;        /* 0 = default */
  int row_begin, row_count; /* token rows owned by this rank (context parallel); 0,0 = all M rows */
  int ring_lo, ring_size, ring_shift; /* rolling cache kept as a ring: cache_row0 / kv_lo / kv_hi are LOGICAL rows; logical row
                               r >= ring_lo lives at ring_lo + (r - ring_lo + ring_shift) % ring_size (ring_size 0: no ring) */
  int text_rows;            /* > 0: rows [text_rows, text_len) of the prompt embedding behind the cross-attention caches are all
                               zero padding, i.e. their cached K / V rows are identical: the cross-attention attends rows
                               [0, text_rows] with the last one counted text_len - text_rows times (rtv_attn_fwd_dup).
                               0: attend all text_len rows */
  int kv_only;              /* 1: the caller only wants the KV cache filled and discards the output (the session's KV-recompute
                               pass, release_server.py:611-632): everything behind the LAST layer's cache write - its q projection,
                               attention, o-projection, cross-attention, FFN, the head - is skipped, `out` is left untouched */
  int attn_kv_splits;       /* > 1: the self-attention launches of a token- or head-sharded call (rtv_dit_layer_attn_hp,
                               rtv_dit_layer_rest with row_count < M) cut their key window into this many ranges
                               (rtv_attn_fwd_split); 0 / 1: one launch.  Not bit-identical with the unsplit forward.  Must not
                               exceed rtv_dit_config.max_attn_kv_splits (the workspace holds the partials): an error otherwise. */
} rtv_dit_step;

size_t rtv_dit_workspace_bytes(const rtv_dit_config* cfg, int F, int gh, int gw);
int rtv_dit_forward(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step,
                    void* workspace, size_t workspace_bytes, rtv_stream_t stream);

/* Phase API for token-axis (context-parallel) sharding — the new design for wan/distributed
 * (xdit_context_parallel.py:131-142 chunk + all_gather pattern, applied to the causal model):
 *   begin; for each layer { layer_qkv; <host: all-gather cache rows [cache_row0, cache_row0+M) of K and V
 *   over RCCL>; layer_rest }; head -> head_rows[row_begin..]; <host: all-gather head_rows>; finish.
 * Every call works on the local rows [row_begin, row_begin+row_count) only; head_rows is a caller-owned
 * [M][out_dim*4] bf16 buffer. */
int rtv_dit_begin(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step,
                  void* workspace, size_t workspace_bytes, rtv_stream_t stream);
int rtv_dit_layer_qkv(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step, int layer,
                      void* workspace, size_t workspace_bytes, rtv_stream_t stream);
int rtv_dit_layer_rest(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step, int layer,
                       void* workspace, size_t workspace_bytes, rtv_stream_t stream);
/* rtv_dit_layer_qkv in pieces, so that the host can start the exchange of one projection and let it run UNDER the other
 * projection (north_star: collectives overlapped with compute; design hint xdit_context_parallel.py:131-142):
 *   parts = RTV_PROJ_LN (LayerNorm + modulation of the layer input, needed once) | RTV_PROJ_Q (q columns of the fused QKV
 *   weight -> RMSNorm + RoPE -> local q, or q_send) | RTV_PROJ_KV (k, v columns -> RMSNorm(k) + RoPE(k) -> cache rows, or
 *   kv_send).  q_send / kv_send: the head-parallel exchange buffers of rtv_dit_layer_qkv_hp (world > 1), or NULL for the
 *   K/V-all-gather exchange.  Typical: [LN|Q] -> all-to-all(q) async -> [KV] -> all-to-all(k|v) -> wait;  or, with the
 *   all-gather exchange, [LN|KV] -> all-gather(K/V rows) async -> [Q] -> wait. */
enum { RTV_PROJ_LN = 1, RTV_PROJ_Q = 2, RTV_PROJ_KV = 4 };
int rtv_dit_layer_proj(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step, int layer,
                       int parts, int world, void* q_send, void* kv_send,
                       void* workspace, size_t workspace_bytes, rtv_stream_t stream);
int rtv_dit_head(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step, void* head_rows,
                 void* workspace, size_t workspace_bytes, rtv_stream_t stream);
int rtv_dit_finish(const rtv_dit_config* cfg, const rtv_dit_step* step, const void* head_rows, rtv_stream_t stream);

/* Head-parallel exchange around self-attention (what the reference's usp_attn_forward does through
 * xFuserLongContextAttention, xdit_context_parallel.py:149-190): per layer
 *   layer_qkv_hp -> <host: all-to-all q_send -> q_all, kv_send -> this rank's cache rows [cache_row0, cache_row0+M)>
 *   -> layer_attn_hp -> <host: all-to-all o_all -> o_recv> -> layer_rest_hp
 * replaces layer_qkv / all-gather / layer_rest.  Needs num_heads % world == 0 and M % world == 0 (equal shards, row_begin =
 * rank * M/world).  With gc = (num_heads/world) * 128:
 *   q_send [world][M/world][gc], kv_send [world][M/world][2][gc] (K then V of a row), q_all / o_all [M][gc],
 *   o_recv [world][M/world][gc]; block g of a send buffer goes to rank g, block g of o_recv came from rank g.
 * In layer_attn_hp step->kv_k / kv_v address the heads THIS rank owns: [kv_size][num_heads/world][128], row stride
 * kv_row_stride.  The other step fields and the remaining phases (begin, head, finish) are unchanged. */
int rtv_dit_layer_qkv_hp(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step, int layer, int world,
                         void* q_send, void* kv_send, void* workspace, size_t workspace_bytes, rtv_stream_t stream);
int rtv_dit_layer_attn_hp(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step, int layer, int world,
                          const void* q_all, void* o_all, void* workspace, size_t workspace_bytes, rtv_stream_t stream);
int rtv_dit_layer_rest_hp(const rtv_dit_config* cfg, const rtv_dit_weights* w, const rtv_dit_step* step, int layer, int world,
                          const void* o_recv, void* workspace, size_t workspace_bytes, rtv_stream_t stream);

/* ---- optional fp8 weight path (BASELINE config 5; reference: release_server.py:179-182 = torchao
 * Float8DynamicActivationFloat8WeightConfig(PerTensor) on every nn.Linear): OCP e4m3 operands, fp32 accumulation,
 *   y = bf16( (q(x) . q(W)^T) * s_x * s_w + bias ), s = max|t| / 448 over the whole tensor, q(t) = e4m3(clamp(t / s, +-448)).
 * rtv_quantize_fp8: x [M, d] bf16 (row stride ld) -> q [M, d] e4m3 (row stride ldq bytes) and *scale_out = s_x (device float);
 *   amax_scratch = 4 device bytes.  d % 16 == 0.
 * rtv_gemm_fp8: C[M,N] bf16 = epilogue((A . W^T) * a_scale[0] * w_scale): A [M,K], W [N,K] e4m3 (lda / ldw in bytes, K % 128 == 0);
 *   the epilogue arguments are rtv_gemm's.  Uses the split-K workspace of rtv_gemm_set_workspace when attached. */
int rtv_quantize_fp8(const void* x, int64_t ld, int M, int d, void* q, int64_t ldq, float* scale_out, void* amax_scratch,
                     rtv_stream_t stream);
int rtv_gemm_fp8(const void* A, int lda, const void* W, int ldw, const float* a_scale, float w_scale, void* C, int ldc,
                 int M, int N, int K, const void* bias, int act, const void* gate, int gate_stride, int rows_per_frame,
                 int row_offset, const void* residual, int ldr, rtv_stream_t stream);

/* ---- K6/K7: streaming VAE decoder / encoder -----------------------------------------------------
 * fp16, channels-last activations [T][H][W][C].
 * rtv_conv_cl: implicit-GEMM convolution replacing CausalConv3d / Conv2d / time_conv of the VAE
 *   (wan/modules/vae.py:17-36,:84-96; demo_utils/vae_block3.py:19-29,:61-72).  `in` is the concat buffer
 *   [cached slices | new slices] for temporal kernels (output frame t reads slices t..t+kt-1); spatial
 *   zero padding kh/2; w is [Cout][kt*kh*kw][Cin]; T,H,W are the OUTPUT grid.  `resample`:
 *     RTV_CONV_NONE         plain conv;
 *     RTV_CONV_UPSAMPLE2X   input [T][H/2][W/2] read through a nearest 2x upsampling (decoder Resample);
 *     RTV_CONV_DOWN2X       1x3x3, stride 2, zero pad on the high side only = ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2)
 *                           of the encoder's Resample (vae.py:84-92), input [T][2H][2W];
 *     RTV_CONV_TIME_DOWN2X  3x1x1, time stride 2, no padding (encoder time_conv, vae.py:96,:151-156): output frame t
 *                           reads input slices 2t..2t+2, input [2T+1][H][W].
 *   n_split>0 scatters output channel halves to frames 2t, 2t+1 (decoder time_conv).
 *   Cin % 32 == 0, Cout % 8 == 0; `zeros` = >=16 zero bytes. */
enum { RTV_CONV_NONE = 0, RTV_CONV_UPSAMPLE2X = 1, RTV_CONV_DOWN2X = 2, RTV_CONV_TIME_DOWN2X = 3 };
/* OR-ed into `resample`: run a 3x3x3 stride-1 convolution on the implicit-GEMM gather kernel instead of the halo-tile kernel
 * (the VAE orchestrators set it for the latent-resolution layers: a 60 x 104 image is 16 halo tiles, too few for 256 CUs).
 * A property of the LAYER, never of the launch size: row-sharded and unsharded decodes must pick the same kernel per layer. */
enum { RTV_CONV_GATHER = 16 };
int rtv_conv_cl(const void* in, const void* w, const void* bias, const void* residual, int res_ld,
                void* out, int out_ld, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                int resample, int n_split, const void* zeros, rtv_stream_t stream);
/* Row-window form of RTV_CONV_UPSAMPLE2X for the spatially sharded decode: the output buffer holds image rows
 * [y_out0, y_out0+H) (output resolution), the input buffer in_rows rows starting at image row y_in0 (input resolution);
 * img_rows = image height at output resolution (taps outside the IMAGE read zeros). */
int rtv_conv_cl_win(const void* in, const void* w, const void* bias, const void* residual, int res_ld,
                    void* out, int out_ld, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                    int resample, int n_split, const void* zeros, int y_out0, int y_in0, int in_rows, int img_rows,
                    rtv_stream_t stream);
/* The first conv of a ResidualBlock with the RMS_norm * gamma + SiLU behind it (wan/modules/vae.py:186-192) in its epilogue:
 * out = SiLU(RMS_norm(conv3x3x3(in) + bias) * gamma), `in` = the causal concat buffer as for rtv_conv_cl.  Returns 1 - nothing
 * launched, no error - when the layer is not one the halo-tile conv kernel takes with all channels of a pixel in one workgroup
 * (Cout == 96, Cin % 32 == 0, not RTV_CONV_GATHER): run rtv_conv_cl + rtv_rmsnorm_silu_cl then.  (The two forms differ by
 * the fp32 summation order of the 96 squares; include/rtv_hip_lab.h has the A/B switch.) */
int rtv_conv3_norm_silu_cl(const void* in, const void* w, const void* bias, const void* gamma, void* out, int out_ld,
                           int T, int H, int W, int Cin, int Cout, int flags, const void* zeros, rtv_stream_t stream);
/* RMS_norm over channels (+SiLU) on channels-last pixels (wan/modules/vae.py:39-54): C in {96,192,384}. */
int rtv_rmsnorm_silu_cl(const void* x, void* out, const void* gamma, int C, int64_t npix, int apply_silu,
                        rtv_stream_t stream);
/* row softmax of the single-head mid-block attention (vae.py:241-245): p[r][:n] = softmax(s[r][:n]), zero pad to ldp */
int rtv_softmax_rows(const void* s, int lds, void* p, int ldp, int rows, int n, rtv_stream_t stream);

typedef struct rtv_vae_conv { const void* w; const void* b; } rtv_vae_conv;       /* fp16 [Cout][taps][Cin], [Cout] */
typedef struct rtv_vae_res {                                                        /* ResidualBlock, vae.py:175-209 */
  const void* gamma0; rtv_vae_conv conv_a; const void* gamma3; rtv_vae_conv conv_b; rtv_vae_conv shortcut; /* shortcut.w nullable */
} rtv_vae_res;
typedef struct rtv_vae_attn {                                                       /* AttentionBlock, vae.py:212-251 */
  const void* gamma; const void *wq, *bq, *wk, *bk, *wv, *bv, *wproj, *bproj;      /* wq/bq pre-scaled by 1/sqrt(384) */
} rtv_vae_attn;
typedef struct rtv_vae_weights {
  const void *conv2_w, *conv2_b, *mean, *std;   /* float32: [16][16], [16], [16], [16] (vae_block3.py:181-193) */
  rtv_vae_conv conv1;                            /* Cin padded 16 -> 32 */
  rtv_vae_res mid0, mid2, up[12];
  rtv_vae_attn attn;
  rtv_vae_conv time_conv[2], resample[3];
  const void* head_gamma;
  rtv_vae_conv head;                             /* Cout padded 3 -> 8 */
} rtv_vae_weights;

/* Caller-owned arena: 32 feature caches (the first two slices of each conv's concat buffer) + scratch.
 * Zero it before the first call of a stream (`first` = 1: feat_cache slots are None, the first latent
 * frame skips the temporal upsampling, vae_block3.py:51-53). */
size_t rtv_vae_arena_bytes(int h, int w);
int rtv_vae_cache_slot(int h, int w, int slot, size_t* offset, int* C, int* H, int* W);
/* z: fp16 [T][16][h][w] latents; pixels: float32 [T'][3][8h][8w] in [-1,1], T' = 4T (4T-3 when first). */
int rtv_vae_decode(const rtv_vae_weights* w, const void* z, int T, int h, int wd, int first,
                   void* arena, size_t arena_bytes, void* pixels, rtv_stream_t stream);
/* VAEDecoderWrapperSingle.forward (demo_utils/vae.py:150-195 over the VAEDecoder3d / Resample twins :50-123, :198-314): ONE latent
 * frame per call, `is_first_frame` given by the caller, the feature caches always present (zeros before the first frame:
 * demo_utils/constant.py:6-39).  z: fp16 [1][16][h][w]; pixels: float32 [4][3][8h][8w] in [-1,1] - also on the first frame, where
 * this form interleaves a zero frame instead of skipping the temporal doubling (:112-116) and leaves the time_conv caches
 * untouched (:86-90); after a later frame a time_conv cache is [zeros, x] (:106-111).  Same arena layout as rtv_vae_decode. */
int rtv_vae_decode_single(const rtv_vae_weights* w, const void* z, int h, int wd, int is_first_frame,
                          void* arena, size_t arena_bytes, void* pixels, rtv_stream_t stream);
/* Output side of the decoder (the frame path of release_server.py:978-991 + :972): pixels float32 [T][3][H][W] in
 * [-1,1] -> rgb8 [T][H][W][3] = u8(trunc(clamp((x + 1) * 0.5, 0, 1) * 255)) - the bytes the reference hands to the JPEG encoder
 * (host-side add_(1).mul_(0.5).clamp_(0,1), then torchvision to_pil_image's mul(255).byte()) - computed on the GPU so that
 * the device-to-host copy moves 1 byte per sample instead of 4.  H*W % 4 == 0. */
int rtv_pixels_to_rgb8(const void* pixels, void* rgb8, int T, int H, int W, rtv_stream_t stream);
/* Spatially sharded decode (multi-GPU, SURVEY 8e "VAE decode: shard by output rows with halo"): produce only pixel rows
 * [row0, row1) of every frame -> pixels float32 [T'][3][row1-row0][8w], bit-identical to those rows of rtv_vae_decode.
 * Stage 0 (latent resolution, global mid-block attention) runs on the whole image; stages 1-3 run on row windows with
 * halo rows (1 per 3x3 conv between the window and the wanted rows).  Arena / cache-slot geometry depend on the rows. */
size_t rtv_vae_arena_bytes_rows(int h, int w, int row0, int row1);
int rtv_vae_cache_slot_rows(int h, int w, int row0, int row1, int slot, size_t* offset, int* C, int* H, int* W,
                            int* first_row);
int rtv_vae_decode_rows(const rtv_vae_weights* w, const void* z, int T, int h, int wd, int first, int row0, int row1,
                        void* arena, size_t arena_bytes, void* pixels, rtv_stream_t stream);

/* Streaming VAE encoder: VAEEncoderWrapper.forward (demo_utils/vae_block3.py:116-175) over Encoder3d
 * (wan/modules/vae.py:254-345; _video_vae config :591-598: dim 96, z_dim 16, temperal_downsample F,T,T).
 * On the T2V path it re-encodes the first context frame once per block (release_server.py:572-575); in v2v /
 * webcam mode it encodes every input frame (release_server.py:489-527; v2v.py:138-158). */
typedef struct rtv_vae_enc_weights {
  rtv_vae_conv conv1;                 /* encoder.conv1, Cin padded 3 -> 32 */
  rtv_vae_res down[8];                /* encoder.downsamples.{0,1,3,4,6,7,9,10} (ResidualBlocks) */
  rtv_vae_conv resample[3];           /* encoder.downsamples.{2,5,8}.resample.1 (Conv2d 3x3 stride 2): [C][9][C] */
  rtv_vae_conv time_conv[2];          /* encoder.downsamples.{5,8}.time_conv (3,1,1)/stride 2: [C][3][C] */
  rtv_vae_res mid0, mid2;             /* encoder.middle.{0,2} */
  rtv_vae_attn attn;                  /* encoder.middle.1 */
  const void* head_gamma;             /* encoder.head.0 */
  rtv_vae_conv head;                  /* encoder.head.2: 384 -> 32 */
  const void *conv1x1_w, *conv1x1_b;  /* float32 [32][32], [32]: the wrapper's conv1 (vae_block3.py:120,:167) */
  const void *mean, *std;             /* float32 [16] latent statistics (vae_block3.py:121-130) */
} rtv_vae_enc_weights;

/* Caller-owned arena: 24 feature caches + scratch; zero it before the first chunk of a stream. */
size_t rtv_vae_enc_arena_bytes(int H, int W);
int rtv_vae_enc_cache_slot(int H, int W, int slot, size_t* offset, int* C, int* h, int* w, int* nslices);
/* One time chunk: frames f16 [3][Ttot][H][W] in [-1,1], chunk = frames t0..t0+tn (tn = 1 with first = 1 on fresh caches,
 * tn = 4 afterwards) -> one latent frame written to mu f16 [16][Tout_tot][H/8][W/8] at index tout. */
int rtv_vae_encode(const rtv_vae_enc_weights* w, const void* frames, int Ttot, int t0, int tn, int H, int W, int first,
                   void* arena, size_t arena_bytes, void* mu, int Tout_tot, int tout, rtv_stream_t stream);

/* ---- hardware-layout probes (test support; see csrc/probe.hip) --------------------------------- */
int rtv_probe_mfma(const void* A /*[32][16] bf16*/, const void* B /*[16][32] bf16*/, void* D /*[32][32] f32*/,
                   rtv_stream_t stream);
int rtv_probe_tr(const void* V /*[64][128] bf16*/, void* out /*[4][64][8] bf16*/, int kbk, int s,
                 rtv_stream_t stream);

/* ---- text encoder: the UMT5-XXL encoder of wan/modules/t5.py:267-313 as WanTextEncoder runs it
 * (utils/wan_wrapper.py:20-56), from token ids to prompt embeddings; head_dim 64, no attention scaling, per-layer relative
 * position bias (shared_pos=False), gated-GELU feed-forward, T5LayerNorm.  All linears are bias-free; weights bf16 (the
 * released checkpoint is bf16; the reference up-casts it to float32), residual stream / norms / softmax float32. */
typedef struct {
  int vocab, dim, dim_attn, dim_ffn, num_heads, num_layers;
  float eps;                      /* 1e-6 */
} rtv_t5_config;
typedef struct {
  const void* norm1_w;            /* bf16 [dim] */
  const void* qk_w;               /* bf16 [2*dim_attn][dim]: attn.q.weight rows, then attn.k.weight rows */
  const void* v_w;                /* bf16 [dim_attn][dim] */
  const void* o_w;                /* bf16 [dim][dim_attn] */
  const void* norm2_w;            /* bf16 [dim] */
  const void* gate_fc1_w;         /* bf16 [2*dim_ffn][dim]: ffn.gate.0.weight rows, then ffn.fc1.weight rows */
  const void* fc2_w;              /* bf16 [dim][dim_ffn] */
  const void* pos_bias;           /* float32 [num_heads][2*max_len-1]: pos_embedding.embedding.weight[bucket(r)][h] at index
                                     r + max_len - 1, r = key - query (T5RelativeEmbedding, t5.py:225-265) */
} rtv_t5_layer_weights;
typedef struct {
  const void* token_embedding;    /* bf16 [vocab][dim] */
  const void* final_norm_w;       /* bf16 [dim] */
  const rtv_t5_layer_weights* layers;   /* host array [num_layers] */
  int max_len;                    /* length the pos_bias tables were built for (512) */
} rtv_t5_weights;
size_t rtv_t5_workspace_bytes(const rtv_t5_config* cfg, int seq_len);
/* ids: device int32 [seq_len] (the prompt's tokens incl. </s>); out: float32 [out_rows][dim] - rows < seq_len hold the encoder
 * output, rows >= seq_len are zero (wan_wrapper.py:52-53).  One prompt per call. */
int rtv_t5_encode(const rtv_t5_config* cfg, const rtv_t5_weights* w, const int* ids, int seq_len, int out_rows,
                  void* workspace, size_t workspace_bytes, void* out, rtv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RTV_HIP_H */
