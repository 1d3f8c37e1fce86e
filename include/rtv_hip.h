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
int rtv_version(void);
const char* rtv_last_error(void);

/* Per-kernel-class hipEvent timing (used by bench.py's roofline block).
 * class ids: 0 gemm, 1 attention, 2 layernorm/modulate, 3 rmsnorm+rope+cache, 4 conv, 5 misc */
int rtv_prof_enable(int on);
int rtv_prof_read(int cls, double* total_ms, int64_t* launches, double* total_work);
int rtv_prof_reset(void);

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

/* ---- K4: projection GEMM with fused epilogue ---------------------------------------------
 * C[M,N] = epi(A[M,K] @ W[N,K]^T): replaces nn.Linear (causal_model.py:196-199,:246,:433-435,
 * :614-623; model.py:184-198) plus the eager chains bias -> GELU(tanh) (:434), y*e[2]+x (:476),
 * y*e[5]+x (:487-488), x + cross_attn (:480).
 *   bias[N] (nullable); act RTV_ACT_*; gate (nullable): gate[(m / rows_per_frame)*gate_stride + n];
 *   residual[M,ldr] (nullable, may alias C).  tile_cfg 0 = default. */
int rtv_gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc,
             int M, int N, int K,
             const void* bias, int act,
             const void* gate, int gate_stride, int rows_per_frame,
             const void* residual, int ldr,
             int dtype, int tile_cfg, rtv_stream_t stream);

/* ---- K5: fused norm / modulation / RoPE / KV-cache write ----------------------------------
 * rtv_layernorm_modulate: out = LN(x; eps, no affine) * (1 + scale[f]) + shift[f], f = m / rows_per_frame
 *   (causal_model.py:471, :483-484, :522; WanLayerNorm model.py:88-98).  shift/scale nullable ->
 *   plain LN; weight/bias (nullable) -> affine LN (norm3, causal_model.py:424-426,:480).
 *   shift and scale point at frame 0's [d] vectors; frame_stride elements between frames. */
int rtv_layernorm_modulate(const void* x, void* out, int M, int d, float eps,
                           const void* shift, const void* scale, int frame_stride, int rows_per_frame,
                           const void* weight, const void* bias, rtv_stream_t stream);

/* rtv_rmsnorm: out = bf16(x * rsqrt(mean(x^2)+eps)) * weight over the full channel dim
 *   (WanRMSNorm model.py:69-85; used for cross-attention q/k, model.py:184,:189). */
int rtv_rmsnorm(const void* x, int ldx, void* out, int ldo, int M, int d, float eps,
                const void* weight, rtv_stream_t stream);

/* rtv_qk_norm_rope_cache: one pass over the fused QKV projection output qkv[M,3d]:
 *   q = rope(rmsnorm(q)*wq) -> q_out[M,d];  k = rope(rmsnorm(k)*wk) -> k_cache rows [cache_row0, cache_row0+M);
 *   v -> v_cache rows likewise.  Replaces causal_model.py:243-256 (norm), :143-171 / model.py:39-66
 *   (3-axis RoPE; rope_cs is float2 [1024][hd/2] = (cos,sin) of rope_params model.py:28-35 laid out as
 *   causal_model.py:639-645), :380-385 / :309-311 (cache write).  Token m -> (f,h,w) on grid (F,gh,gw),
 *   temporal position start_frame + f.  Cache row stride in elements. */
int rtv_qk_norm_rope_cache(const void* qkv, void* q_out, void* k_cache, void* v_cache,
                           int64_t cache_row_stride, int cache_row0,
                           int M, int d, int num_heads, float eps,
                           const void* wq, const void* wk, const void* rope_cs,
                           int F, int gh, int gw, int start_frame, rtv_stream_t stream);

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

#ifdef __cplusplus
}
#endif
#endif /* RTV_HIP_H */
