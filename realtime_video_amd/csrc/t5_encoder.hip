// Text encoder: the UMT5-XXL *encoder* of wan/modules/t5.py:267-313 as WanTextEncoder drives it
// (utils/wan_wrapper.py:20-56) - once per prompt (and per prompt transition), 24 layers of
//   x += O( softmax(Q K^T + relative-position bias) V )          (T5Attention, t5.py:96-135: no 1/sqrt(d) scaling)
//   x += fc2( fc1(n) * gelu_tanh(gate(n)) )                       (T5FeedForward, t5.py:151-157), n = T5LayerNorm(x) (:62-67)
// over the prompt's tokens.  Only rows < seq_len are computed: padding keys are masked for every query (t5.py:121-125) and
// padding rows are zeroed by the caller (wan_wrapper.py:52-53), so they cannot influence the result.
//
// Precision: the reference holds this model in float32 with bf16-representable weights (bf16 checkpoint, wan_wrapper.py:24-33).
// Here the residual stream and every normalisation / softmax / gating stay float32; the linears run on the bf16 MFMA GEMM of
// gemm.hip (fp32 accumulation) with their inputs rounded to bf16 - weights are exact, activations see one rounding per linear.
//
// Attention (head_dim 64, <= 512 tokens, 64 heads): one wave per (head, 32 queries), no LDS.  S^T = K . Q^T with K rows read
// straight from the q|k buffer as the MFMA A operand; O^T += V^T . P^T with P^T = the lane's own S^T registers and V^T read
// from a TRANSPOSED value buffer (produced by running the value projection as W_v . n^T), so that the 8 keys a lane feeds to
// one MFMA are two contiguous 8-byte loads.  The bias of (key - query) comes from a per-layer table [H][2*max_len - 1].
#include "gemm_core.h"
#include "rtv_common.h"
#include "rtv_internal.h"

namespace rtv {

constexpr int T5_THREADS = 256;

__global__ void t5_embed_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ table, float* __restrict__ x,
                                int rows, int valid, int dim, int vocab) {
  const int row = blockIdx.x;
  int id = row < valid ? ids[row] : 0;      // rows in [seq_len, rows) are <pad> tokens (id 0), like the reference's padding
  id = min(max(id, 0), vocab - 1);
  const bf16_t* src = table + (size_t)id * dim;
  float* dst = x + (size_t)row * dim;
  for (int c = threadIdx.x * 8; c < dim; c += blockDim.x * 8) {
    float f[8];
    unpack_bf16x8(*(const u32x4*)(src + c), f);
    *(float4*)(dst + c) = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)(dst + c + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
}

__device__ __forceinline__ float t5_block_sum(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < T5_THREADS / 64; ++w) s += red[w];   // fixed order
  __syncthreads();
  return s;
}

// T5LayerNorm: out = weight * (x * rsqrt(mean(x^2) + eps)), float32 math; OUT_F32 selects the final (float32) output.
template <bool OUT_F32>
__global__ __launch_bounds__(T5_THREADS) void t5_rmsnorm_kernel(const float* __restrict__ x, const bf16_t* __restrict__ w,
                                                                void* __restrict__ out, int dim, float eps) {
  __shared__ float red[T5_THREADS / 64];
  const float* xr = x + (size_t)blockIdx.x * dim;
  float s = 0.f;
  for (int c = threadIdx.x * 4; c < dim; c += T5_THREADS * 4) {
    float4 v = *(const float4*)(xr + c);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  const float r = rsqrtf(t5_block_sum(s, red) / (float)dim + eps);
  for (int c = threadIdx.x * 4; c < dim; c += T5_THREADS * 4) {
    float4 v = *(const float4*)(xr + c);
    const float o0 = bf16_to_f32(w[c]) * (v.x * r), o1 = bf16_to_f32(w[c + 1]) * (v.y * r);
    const float o2 = bf16_to_f32(w[c + 2]) * (v.z * r), o3 = bf16_to_f32(w[c + 3]) * (v.w * r);
    if (OUT_F32) {
      *(float4*)((float*)out + (size_t)blockIdx.x * dim + c) = make_float4(o0, o1, o2, o3);
    } else {
      u32x2 p = {pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)};
      *(u32x2*)((bf16_t*)out + (size_t)blockIdx.x * dim + c) = p;
    }
  }
}

// x (float32) += y (bf16)
__global__ void t5_add_kernel(float* __restrict__ x, const bf16_t* __restrict__ y, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float f[8];
    unpack_bf16x8(*(const u32x4*)(y + i * 8), f);
    float4 a = *(float4*)(x + i * 8), b = *(float4*)(x + i * 8 + 4);
    a.x += f[0]; a.y += f[1]; a.z += f[2]; a.w += f[3];
    b.x += f[4]; b.y += f[5]; b.z += f[6]; b.w += f[7];
    *(float4*)(x + i * 8) = a;
    *(float4*)(x + i * 8 + 4) = b;
  }
}

// h = fc1 * gelu_tanh(gate) from gf = [rows][gate (dff) | fc1 (dff)]  (t5.py:48-52, :152)
__global__ void t5_geglu_kernel(const bf16_t* __restrict__ gf, bf16_t* __restrict__ h, int rows, int dff) {
  const size_t per_row = dff / 8, total = (size_t)rows * per_row;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / per_row, c = (i - r * per_row) * 8;
    float g[8], f[8], o[8];
    unpack_bf16x8(*(const u32x4*)(gf + r * 2 * dff + c), g);
    unpack_bf16x8(*(const u32x4*)(gf + r * 2 * dff + dff + c), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = g[j];
      const float ge = 0.5f * t * (1.0f + tanhf(0.7978845608028654f * (t + 0.044715f * t * t * t)));
      o[j] = f[j] * ge;
    }
    *(u32x4*)(h + r * dff + c) = pack_bf16x8(o);
  }
}

struct T5AttnParams {
  const bf16_t* qk;   // [rows][2*da]: q columns then k columns; head h at column h*64
  const bf16_t* vt;   // [da][rows]: V transposed
  bf16_t* o;          // [rows][da]
  const float* bias;  // [H][2*max_len - 1], entry (key - query) + max_len - 1
  int rows, valid, da, H, max_len;
};

// one wave = 32 queries of one head against all `valid` keys
__global__ __launch_bounds__(64) void t5_attn_kernel(T5AttnParams p) {
  const int lane = threadIdx.x;
  const int l31 = lane & 31, g = lane >> 5;
  const int h = blockIdx.y;
  const int q0 = blockIdx.x * 32;
  const int q_row = q0 + l31;                    // < rows (rows % 32 == 0)
  const int ld = 2 * p.da;
  const bf16_t* qp = p.qk + (size_t)q_row * ld + h * 64 + g * 8;
  u32x4 qf[4];
#pragma unroll
  for (int dc = 0; dc < 4; ++dc) qf[dc] = *(const u32x4*)(qp + dc * 16);
  const bf16_t* kb = p.qk + p.da + h * 64 + g * 8;                       // + key * ld + dc * 16
  const bf16_t* vb = p.vt + (size_t)(h * 64 + l31) * p.rows + 4 * g;     // + db * 32 * rows + key0 + 16 s (+ 8)
  const float* bias = p.bias + (size_t)h * (2 * p.max_len - 1) + (p.max_len - 1) - q_row;

  f32x16 oacc[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  constexpr float LOG2E = 1.4426950408889634f;

  for (int key0 = 0; key0 < p.valid; key0 += 32) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const bf16_t* kp = kb + (size_t)(key0 + l31) * ld;                   // key0 + l31 < rows
#pragma unroll
    for (int dc = 0; dc < 4; ++dc) {
      u32x4 kf = *(const u32x4*)(kp + dc * 16);
      s = Mfma32<false>::run(kf, qf[dc], s);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = key0 + (r & 3) + 8 * (r >> 2) + 4 * g;
      const float v = key < p.valid ? (s[r] + bias[key]) * LOG2E : -INFINITY;
      s[r] = v;
      mx = fmaxf(mx, v);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);        // finite: every block holds at least one valid key (key0 < valid)
    const float alpha = exp2f(m_run - m_new);    // first block: exp2(-inf) = 0
    m_run = m_new;
    l_run *= alpha;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    u32x4 pf[2];
    float ps = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float e0 = exp2f(s[2 * t] - m_run), e1 = exp2f(s[2 * t + 1] - m_run);
      ps += e0 + e1;
      pf[t >> 2][t & 3] = pack_bf16x2(e0, e1);
    }
    l_run += ps;
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const bf16_t* vp = vb + (size_t)db * 32 * p.rows + key0 + 16 * st;
        u32x2 lo = *(const u32x2*)vp;          // keys +4g .. +4g+3
        u32x2 hi = *(const u32x2*)(vp + 8);    // keys +8+4g ..
        u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
        oacc[db] = Mfma32<false>::run(vf, pf[st], oacc[db]);
      }
  }
  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32, 64));
  bf16_t* op = p.o + (size_t)q_row * p.da + h * 64 + 4 * g;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x2 w;
      w[0] = pack_bf16x2(oacc[db][4 * i + 0] * inv, oacc[db][4 * i + 1] * inv);
      w[1] = pack_bf16x2(oacc[db][4 * i + 2] * inv, oacc[db][4 * i + 3] * inv);
      *(u32x2*)(op + db * 32 + i * 8) = w;
    }
}

struct T5Buffers {
  float* x;
  uint16_t *xn, *qk, *vt, *ao, *tmp, *gf, *h;
};

static size_t t5_carve(const rtv_t5_config* c, int rows, char* base, size_t cap, T5Buffers* b, bool* ok) {
  size_t off = 0;
  bool fit = true;
  auto take = [&](size_t bytes) -> void* {
    size_t a = (off + 255) & ~(size_t)255;
    if (base && a + bytes > cap) fit = false;
    off = a + bytes;
    return base ? base + a : nullptr;
  };
  const size_t R = rows, d = c->dim, da = c->dim_attn, dff = c->dim_ffn;
  T5Buffers t;
  t.x = (float*)take(R * d * 4);
  t.xn = (uint16_t*)take(R * d * 2);
  t.qk = (uint16_t*)take(R * 2 * da * 2);
  t.vt = (uint16_t*)take(da * R * 2);
  t.ao = (uint16_t*)take(R * da * 2);
  t.tmp = (uint16_t*)take(R * d * 2);
  t.gf = (uint16_t*)take(R * 2 * dff * 2);
  t.h = (uint16_t*)take(R * dff * 2);
  if (b) *b = t;
  if (ok) *ok = fit;
  return off;
}

static int t5_rows(int seq_len) { return (seq_len + 31) / 32 * 32; }

}  // namespace rtv

using namespace rtv;

#define T5_TRY(expr)         \
  do {                       \
    int _s = (expr);         \
    if (_s != 0) return _s;  \
  } while (0)

extern "C" size_t rtv_t5_workspace_bytes(const rtv_t5_config* cfg, int seq_len) {
  if (!cfg || seq_len <= 0) return 0;
  return t5_carve(cfg, t5_rows(seq_len), nullptr, 0, nullptr, nullptr) + 256;
}

static int t5_gemm(const void* a, int lda, const void* w, int ldw, void* c, int ldc, int M, int N, int K, rtv_stream_t s) {
  return rtv_gemm(a, lda, w, ldw, c, ldc, M, N, K, nullptr, 0, nullptr, 0, 0, 0, nullptr, 0, RTV_DTYPE_BF16, 0, s);
}

extern "C" int rtv_t5_encode(const rtv_t5_config* cfg, const rtv_t5_weights* w, const int* ids, int seq_len, int out_rows,
                             void* workspace, size_t workspace_bytes, void* out, rtv_stream_t stream) {
  if (!cfg || !w || !ids || !workspace || !out) return set_error(-1, "t5_encode: null argument");
  const int d = cfg->dim, da = cfg->dim_attn, dff = cfg->dim_ffn, H = cfg->num_heads;
  if (H <= 0 || da != H * 64) return set_error(-1, "t5_encode: head_dim must be 64 (dim_attn = 64 * num_heads)");
  if (d % 64 || dff % 64 || da % 64) return set_error(-1, "t5_encode: dim, dim_attn, dim_ffn must be multiples of 64");
  if (seq_len <= 0 || seq_len > w->max_len || out_rows < seq_len)
    return set_error(-1, "t5_encode: need 0 < seq_len <= max_len (the bias tables' length) and out_rows >= seq_len");
  if (((uintptr_t)workspace) & 255) return set_error(-1, "t5_encode: workspace must be 256-byte aligned");
  const int rows = t5_rows(seq_len);
  T5Buffers b;
  bool ok = true;
  t5_carve(cfg, rows, (char*)workspace, workspace_bytes, &b, &ok);
  if (!ok) return set_error(-1, "t5_encode: workspace too small (see rtv_t5_workspace_bytes)");
  hipStream_t st = (hipStream_t)stream;
  const size_t n8 = (size_t)rows * d / 8;
  const int ew_blocks = (int)((n8 + 255) / 256 > 4096 ? 4096 : (n8 + 255) / 256);

  hipLaunchKernelGGL(t5_embed_kernel, dim3(rows), dim3(128), 0, st, ids, (const bf16_t*)w->token_embedding, b.x, rows, seq_len,
                     d, cfg->vocab);
  T5_TRY(check_launch("t5_embed"));
  for (int l = 0; l < cfg->num_layers; ++l) {
    const rtv_t5_layer_weights& lw = w->layers[l];
    hipLaunchKernelGGL(t5_rmsnorm_kernel<false>, dim3(rows), dim3(T5_THREADS), 0, st, b.x, (const bf16_t*)lw.norm1_w, b.xn, d,
                       cfg->eps);
    T5_TRY(check_launch("t5_rmsnorm"));
    T5_TRY(t5_gemm(b.xn, d, lw.qk_w, d, b.qk, 2 * da, rows, 2 * da, d, stream));        // q | k
    T5_TRY(t5_gemm(lw.v_w, d, b.xn, d, b.vt, rows, da, rows, d, stream));                // V^T = W_v . n^T
    T5AttnParams ap{b.qk, b.vt, b.ao, (const float*)lw.pos_bias, rows, seq_len, da, H, w->max_len};
    {
      ProfScope prof(PROF_ATTN, st, 4.0 * rows * (double)seq_len * da);
      hipLaunchKernelGGL(t5_attn_kernel, dim3(rows / 32, H), dim3(64), 0, st, ap);
    }
    T5_TRY(check_launch("t5_attn"));
    T5_TRY(t5_gemm(b.ao, da, lw.o_w, da, b.tmp, d, rows, d, da, stream));
    hipLaunchKernelGGL(t5_add_kernel, dim3(ew_blocks), dim3(256), 0, st, b.x, b.tmp, n8);
    T5_TRY(check_launch("t5_add"));
    hipLaunchKernelGGL(t5_rmsnorm_kernel<false>, dim3(rows), dim3(T5_THREADS), 0, st, b.x, (const bf16_t*)lw.norm2_w, b.xn, d,
                       cfg->eps);
    T5_TRY(check_launch("t5_rmsnorm"));
    T5_TRY(t5_gemm(b.xn, d, lw.gate_fc1_w, d, b.gf, 2 * dff, rows, 2 * dff, d, stream));  // gate | fc1
    {
      const size_t tot = (size_t)rows * dff / 8;
      const int blocks = (int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256);
      hipLaunchKernelGGL(t5_geglu_kernel, dim3(blocks), dim3(256), 0, st, b.gf, b.h, rows, dff);
      T5_TRY(check_launch("t5_geglu"));
    }
    T5_TRY(t5_gemm(b.h, dff, lw.fc2_w, dff, b.tmp, d, rows, d, dff, stream));
    hipLaunchKernelGGL(t5_add_kernel, dim3(ew_blocks), dim3(256), 0, st, b.x, b.tmp, n8);
    T5_TRY(check_launch("t5_add"));
  }
  // final norm on the prompt's rows; padding rows of the output are zero (wan_wrapper.py:52-53)
  hipLaunchKernelGGL(t5_rmsnorm_kernel<true>, dim3(seq_len), dim3(T5_THREADS), 0, st, b.x, (const bf16_t*)w->final_norm_w, out, d,
                     cfg->eps);
  T5_TRY(check_launch("t5_rmsnorm"));
  if (out_rows > seq_len) {
    hipError_t e = hipMemsetAsync((float*)out + (size_t)seq_len * d, 0, (size_t)(out_rows - seq_len) * d * 4, st);
    if (e != hipSuccess) return set_error(e, "t5_encode: hipMemsetAsync");
  }
  return 0;
}
