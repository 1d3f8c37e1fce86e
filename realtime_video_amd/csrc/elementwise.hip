// HBM-bound fused elementwise kernels of the DiT block: LayerNorm+AdaLN modulation, RMSNorm,
// RMSNorm(q,k)+3-axis RoPE+KV-cache write, modulation tables, sinusoidal embedding, (un)patchify.
// LayerNorm / RMSNorm: one 256-thread workgroup per token row; RoPE / cache write: two waves per row.  Every access is a 16-byte
// (8 x bf16) vector.
// bf16 rounding points follow the reference's eager chains (cited per kernel in include/rtv_hip.h).
#include "rtv_common.h"
#include "rtv_internal.h"

namespace rtv {

// RoPE / cache kernel: -1 = by row count (wave form from ROPE_WAVE_MIN_ROWS rows on), 0 = workgroup per row, 1 = two waves per row
static std::atomic<int> g_rope_wave{-1};
constexpr int ROPE_WAVE_MIN_ROWS = 2048;

constexpr int EW_THREADS = 256;
constexpr int EW_MAXC = 4;  // rows of up to EW_THREADS * 8 * EW_MAXC = 8192 elements (16 chunks per lane of the row's wave)

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // protect `red` from the previous use
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
// ------------------------------------------------------------------ LayerNorm (+ modulation), one 256-thread workgroup per row
__global__ __launch_bounds__(EW_THREADS) void layernorm_modulate_kernel(
    const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int d, float eps,
    const bf16_t* __restrict__ shift, const bf16_t* __restrict__ scale, int frame_stride,
    int rows_per_frame, int row_offset, const bf16_t* __restrict__ weight, const bf16_t* __restrict__ bias) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const int nchunks = d >> 3;
  const bf16_t* xr = x + (size_t)row * d;
  float v[EW_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < EW_MAXC; ++i) {
    int c = threadIdx.x + i * EW_THREADS;
    if (c < nchunks) {
      u32x4 raw = *(const u32x4*)(xr + c * 8);
      unpack_bf16x8(raw, v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    }
  }
  const float mean = block_sum(s, red) / (float)d;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < EW_MAXC; ++i) {
    int c = threadIdx.x + i * EW_THREADS;
    if (c < nchunks) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = v[i][j] - mean;
        s2 += t * t;
      }
    }
  }
  const float var = block_sum(s2, red) / (float)d;
  const float rstd = rsqrtf(var + eps);
  const int f = rows_per_frame > 0 ? (row_offset + row) / rows_per_frame : 0;
  const bf16_t* sh = shift ? shift + (size_t)f * frame_stride : nullptr;
  const bf16_t* sc = scale ? scale + (size_t)f * frame_stride : nullptr;
  bf16_t* orow = out + (size_t)row * d;
#pragma unroll
  for (int i = 0; i < EW_MAXC; ++i) {
    int c = threadIdx.x + i * EW_THREADS;
    if (c < nchunks) {
      float o[8];
      if (weight) {  // affine LayerNorm: one rounding at the output (torch kernel computes in f32)
        float w8[8], b8[8];
        unpack_bf16x8(*(const u32x4*)(weight + c * 8), w8);
        unpack_bf16x8(*(const u32x4*)(bias + c * 8), b8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * w8[j] + b8[j];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd;
        if (sc) {  // bf16(bf16(bf16(ln) * bf16(1 + scale)) + shift)
          float sc8[8], sh8[8];
          unpack_bf16x8(*(const u32x4*)(sc + c * 8), sc8);
          unpack_bf16x8(*(const u32x4*)(sh + c * 8), sh8);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            float n = round_bf16(o[j]);
            float a = round_bf16(1.0f + sc8[j]);
            float p = round_bf16(n * a);
            o[j] = p + sh8[j];
          }
        }
      }
      *(u32x4*)(orow + c * 8) = pack_bf16x8(o);
    }
  }
}

// ------------------------------------------------------------------ RMSNorm over the full channel dim, one workgroup per row
__global__ __launch_bounds__(EW_THREADS) void rmsnorm_kernel(const bf16_t* __restrict__ x, int ldx,
                                                             bf16_t* __restrict__ out, int ldo, int d,
                                                             float eps,
                                                             const bf16_t* __restrict__ weight) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const int nchunks = d >> 3;
  const bf16_t* xr = x + (size_t)row * ldx;
  float v[EW_MAXC][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < EW_MAXC; ++i) {
    int c = threadIdx.x + i * EW_THREADS;
    if (c < nchunks) {
      unpack_bf16x8(*(const u32x4*)(xr + c * 8), v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j] * v[i][j];
    }
  }
  const float r = rsqrtf(block_sum(s, red) / (float)d + eps);
  bf16_t* orow = out + (size_t)row * ldo;
#pragma unroll
  for (int i = 0; i < EW_MAXC; ++i) {
    int c = threadIdx.x + i * EW_THREADS;
    if (c < nchunks) {
      float w8[8], o[8];
      unpack_bf16x8(*(const u32x4*)(weight + c * 8), w8);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = round_bf16(v[i][j] * r) * w8[j];
      *(u32x4*)(orow + c * 8) = pack_bf16x8(o);
    }
  }
}

// ------------------------------------------------------------------ wave-per-row building blocks (r04: the RoPE / cache kernel)
// Measured in round 4 (profiles/r04_row_kernels.txt): a plain copy of a [4680, 5120] bf16 tensor runs at 4.7 TB/s on this chip
// (5.3 TB/s for 16x the rows) - that, not 8 TB/s, is what a read-once / write-once kernel of this size can reach.  LayerNorm and
// RMSNorm were rebuilt with one WAVE per row (all of a lane's chunks in registers, DPP sums, no LDS, no barrier) and came out 5-10 %
// SLOWER than the one-workgroup-per-row kernels above, stand-alone and inside the forward (20.8 vs 18.7 ms per block): those stay.
// The RoPE / cache kernel (3x the bytes per row, two independent norms) does gain from the wave form: 81 -> 64 us.
// Make the compiler forget what it knows about a register-resident chunk: the row kernels unpack their raw bf16 chunks once per
// pass; without this the unpacked floats of the first pass are kept (CSE) for the later ones - 8 registers per chunk instead of 4 -
// and the kernels spill or lose occupancy.
__device__ __forceinline__ void launder(u32x4& v) {
  uint32_t a = v[0], b = v[1], c = v[2], e = v[3];
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(e));
  v[0] = a;
  v[1] = b;
  v[2] = c;
  v[3] = e;
}

constexpr int ROWS_PER_WG = 4;

// ------------------------------------------------------------------ RMSNorm(q,k) + RoPE + KV-cache write
struct RopeArgs {
  const bf16_t* qkv;
  bf16_t* q_out;
  bf16_t* k_cache;
  bf16_t* v_cache;
  int64_t cache_row_stride;
  int cache_row0;
  int M, d, hd;
  float eps;
  const bf16_t* wq;
  const bf16_t* wk;
  const float2* rope_cs;  // [1024][hd/2]
  int gh, gw, start_frame, row_offset;
  // head-group scatter (head-parallel exchange, dit_forward.hip): group_cols > 0 sends columns [g*group_cols, (g+1)*group_cols)
  // of local row r to q_out + g*q_group_stride + r*group_cols and k/v + g*kv_group_stride + r*cache_row_stride.
  int group_cols;
  int64_t q_group_stride, kv_group_stride;
  // rolling cache as a ring: logical cache row r >= ring_lo lives at ring_lo + (r - ring_lo + ring_shift) % ring_size
  // (ring_size == 0: no ring, logical == physical); rows below ring_lo are the attention-sink rows and never move.
  int ring_lo, ring_size, ring_shift;
  int parts;   // bit 0: process q, bit 1: process k and v (the split projection of the context-parallel overlap; 3 = all);
               // bit 2 (with bit 1): V is in the cache already (written there by the projection GEMM, r05) - no copy
};

// rotate the 4 complex pairs of one 8-element chunk by the (cos, sin) values in w[0..4)
__device__ __forceinline__ void rope8(float* x, const float2* w) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float a = x[2 * t], b = x[2 * t + 1];
    x[2 * t] = a * w[t].x - b * w[t].y;
    x[2 * t + 1] = a * w[t].y + b * w[t].x;
  }
}

// LANE_CS: head_dim divides 512 (always on the DiT path: 128), so the column of a lane's chunks inside a head - ((lane + 64 i) * 8)
// % hd - does not depend on i: the four (cos, sin) pairs a lane rotates with are the SAME for all of its chunks of q and of k; they
// are fetched once per row, before the row data, and sit in 8 registers.  Otherwise they are fetched per chunk.
//
// TWO waves per row when K / V are processed: the q norm and the k norm are independent reductions, so one wave takes q (+ the first
// half of the V copy), the other k (+ the second half): 40 raw registers per wave instead of 80 at d = 5120, no exchange between
// the two.  (One wave per row held q and k: 148 registers = 3 waves per SIMD = 3072 rows in flight of 4680.)
template <int CPL, bool LANE_CS>
__global__ __launch_bounds__(EW_THREADS) __attribute__((amdgpu_waves_per_eu(CPL <= 10 ? 5 : 3, 8))) void qk_norm_rope_cache_kernel(RopeArgs a) {
  const int lane = threadIdx.x & 63;
  const bool do_q = a.parts & 1, do_kv = a.parts & 2;   // kernel-uniform
  const int nroles = do_kv ? 2 : 1;
  const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * ROWS_PER_WG + (threadIdx.x >> 6));
  const int row = wid / nroles, role = wid - row * nroles;   // role 0: q + V chunks [0, CPL/2); role 1: k + V chunks [CPL/2, CPL)
  if (row >= a.M) return;
  const int d = a.d;
  const int nchunks = d >> 3;
  const bf16_t* qr = a.qkv + (size_t)row * 3 * d;
  const bf16_t* vr = qr + 2 * d;
  const bool norm = role == 1 || do_q;                // this wave normalises + rotates a vector (q or k)
  const bf16_t* xr = role == 1 ? qr + d : qr;         // ... read from here
  const bf16_t* wx = role == 1 ? a.wk : a.wq;

  // positions of this token (wave-uniform) and the (cos, sin) pairs of this lane's columns
  const int half = a.hd >> 1;
  const int c1 = half / 3, c0 = half - 2 * c1;
  const int per_frame = a.gh * a.gw;
  const int grow = a.row_offset + row;  // global token index
  const int f = grow / per_frame;
  const int rem = grow - f * per_frame;
  const int pos_h = rem / a.gw, pos_w = rem - pos_h * a.gw;
  const int pos_f = a.start_frame + f;
  auto load_cs = [&](int col, float2* w) {
    const int pj0 = (col % a.hd) >> 1;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int pj = pj0 + t;
      const int pos = pj < c0 ? pos_f : (pj < c0 + c1 ? pos_h : pos_w);
      w[t] = a.rope_cs[pos * half + pj];
    }
  };
  float2 cs[4];
  if (LANE_CS && norm) load_cs(lane * 8, cs);

  const int gc = a.group_cols;
  size_t kv_row = gc ? (size_t)row : (size_t)(a.cache_row0 + grow);
  if (!gc && a.ring_size > 0 && (int)kv_row >= a.ring_lo)
    kv_row = (size_t)(a.ring_lo + ((int)kv_row - a.ring_lo + a.ring_shift) % a.ring_size);
  bf16_t* xo = role == 1 ? a.k_cache + kv_row * a.cache_row_stride : a.q_out + (size_t)row * (gc ? gc : d);
  const int64_t x_group_stride = role == 1 ? a.kv_group_stride : a.q_group_stride;
  bf16_t* vo = a.v_cache + kv_row * a.cache_row_stride;

  u32x4 raw[CPL];
  if (norm) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
      const int c = lane + i * 64;
      if (c < nchunks) raw[i] = *(const u32x4*)(xr + c * 8);
    }
  }
  // V is a plain copy into the cache; this wave moves its half of the row's chunks while its own vector is in flight
  if (do_kv && !(a.parts & 4)) {
    constexpr int VH = (CPL + 1) / 2;
    const int i0 = role * VH;
    u32x4 vraw[VH];
#pragma unroll
    for (int i = 0; i < VH; ++i) {
      const int c = lane + (i0 + i) * 64;
      if (i0 + i < CPL && c < nchunks) vraw[i] = *(const u32x4*)(vr + c * 8);
    }
#pragma unroll
    for (int i = 0; i < VH; ++i) {
      const int c = lane + (i0 + i) * 64;
      if (i0 + i < CPL && c < nchunks) {
        const int g = gc ? (c * 8) / gc : 0;
        *(u32x4*)(vo + g * a.kv_group_stride + (c * 8 - g * gc)) = vraw[i];
      }
    }
  }
  if (!norm) return;
  // The sum of squares in the CANONICAL order both forms of this kernel share (so that the form - chosen by the launch's row count -
  // never shows in the bits: a token shard of 585 rows and the unsharded 4680 rows give the same K / V): lane column l keeps four
  // accumulators, acc[w] += squares of chunk l + 64 (w + 4 i) for i = 0, 1, ... (elements in order, fused multiply-add) - exactly
  // what thread (wave w, lane l) of the workgroup-per-row form accumulates -, then ((acc0 + acc1) + acc2) + acc3, then the 64-lane
  // butterfly.
  float acc4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    if (lane + i * 64 < nchunks) {
      float t[8];
      unpack_bf16x8(raw[i], t);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc4[i & 3] = __builtin_fmaf(t[j], t[j], acc4[i & 3]);
    }
  }
  const float sx = ((acc4[0] + acc4[1]) + acc4[2]) + acc4[3];
  const float rx = rsqrtf(wave_sum(sx) / (float)d + a.eps);
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < CPL; ++i) launder(raw[i]);
#pragma unroll
  for (int i = 0; i < CPL; ++i) {
    const int c = lane + i * 64;
    if (c < nchunks) {
      const int g = gc ? (c * 8) / gc : 0;
      const int col = c * 8 - g * gc;
      float2 wc[4];
      if (!LANE_CS) load_cs(c * 8, wc);
      float w8[8], x[8];
      unpack_bf16x8(*(const u32x4*)(wx + c * 8), w8);
      unpack_bf16x8(raw[i], x);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = round_bf16(round_bf16(x[j] * rx) * w8[j]);
      rope8(x, LANE_CS ? cs : wc);
      *(u32x4*)(xo + g * x_group_stride + col) = pack_bf16x8(x);
    }
  }
}

// ---- the same pass with ONE 256-thread workgroup per row (the round 1-3 kernel): for launches of few rows - the token shards of
// context parallelism (585 rows at 8 ranks) - where a launch lasts as long as its slowest row and four waves finish a row sooner
// than one (r04: with the wave kernel on every launch the simulated 8-rank forward's row-kernel class went 96 -> 111 ms summed
// over the shards, profiles/r04_row_kernels.txt).
// Row sums in the canonical order of the wave form (see qk_norm_rope_cache_kernel): thread (wave w, lane l) holds acc_w(l); every
// wave combines the four accumulators of its lane column in the order ((0 + 1) + 2) + 3 and runs the same 64-lane butterfly.
__device__ __forceinline__ void block_sum2(float& a, float& b, float* red /* [512] */) {
  const int l = threadIdx.x & 63;
  red[threadIdx.x] = a;
  red[256 + threadIdx.x] = b;
  __syncthreads();
  a = wave_sum(((red[l] + red[64 + l]) + red[128 + l]) + red[192 + l]);
  b = wave_sum(((red[256 + l] + red[320 + l]) + red[384 + l]) + red[448 + l]);
}

__device__ __forceinline__ void rope8_cols(float* x, int col, int hd, int c0, int c1, int pos_f, int pos_h,
                                      int pos_w, const float2* __restrict__ cs) {
  const int half = hd >> 1;
  const int pj0 = (col % hd) >> 1;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    int pj = pj0 + t;
    int pos = pj < c0 ? pos_f : (pj < c0 + c1 ? pos_h : pos_w);
    float2 w = cs[pos * half + pj];
    float a = x[2 * t], b = x[2 * t + 1];
    x[2 * t] = a * w.x - b * w.y;
    x[2 * t + 1] = a * w.y + b * w.x;
  }
}

__global__ __launch_bounds__(EW_THREADS) void qk_norm_rope_cache_block_kernel(RopeArgs a) {
  __shared__ float red[512];
  const int row = blockIdx.x;
  const int d = a.d;
  const int nchunks = d >> 3;
  const bf16_t* qr = a.qkv + (size_t)row * 3 * d;
  const bf16_t* kr = qr + d;
  const bf16_t* vr = kr + d;
  float q[EW_MAXC][8], k[EW_MAXC][8];
  u32x4 vraw[EW_MAXC];
  float sq = 0.f, sk = 0.f;
  const bool do_q = a.parts & 1, do_kv = a.parts & 2;   // kernel-uniform
#pragma unroll
  for (int i = 0; i < EW_MAXC; ++i) {
    int c = threadIdx.x + i * EW_THREADS;
    if (c < nchunks) {
      if (do_q) {
        unpack_bf16x8(*(const u32x4*)(qr + c * 8), q[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) sq = __builtin_fmaf(q[i][j], q[i][j], sq);
      }
      if (do_kv) {
        unpack_bf16x8(*(const u32x4*)(kr + c * 8), k[i]);
        if (!(a.parts & 4)) vraw[i] = *(const u32x4*)(vr + c * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) sk = __builtin_fmaf(k[i][j], k[i][j], sk);
      }
    }
  }
  block_sum2(sq, sk, red);
  const float rq = rsqrtf(sq / (float)d + a.eps);
  const float rk = rsqrtf(sk / (float)d + a.eps);

  const int half = a.hd >> 1;
  const int c1 = half / 3, c0 = half - 2 * c1;
  const int per_frame = a.gh * a.gw;
  const int grow = a.row_offset + row;  // global token index
  const int f = grow / per_frame;
  const int rem = grow - f * per_frame;
  const int pos_h = rem / a.gw, pos_w = rem - pos_h * a.gw;
  const int pos_f = a.start_frame + f;

  const int gc = a.group_cols;
  size_t kv_row = gc ? (size_t)row : (size_t)(a.cache_row0 + grow);
  if (!gc && a.ring_size > 0 && (int)kv_row >= a.ring_lo)
    kv_row = (size_t)(a.ring_lo + ((int)kv_row - a.ring_lo + a.ring_shift) % a.ring_size);
  bf16_t* qo = a.q_out + (size_t)row * (gc ? gc : d);
  bf16_t* ko = a.k_cache + kv_row * a.cache_row_stride;
  bf16_t* vo = a.v_cache + kv_row * a.cache_row_stride;
#pragma unroll
  for (int i = 0; i < EW_MAXC; ++i) {
    int c = threadIdx.x + i * EW_THREADS;
    if (c < nchunks) {
      float wq8[8], wk8[8];
      unpack_bf16x8(*(const u32x4*)(a.wq + c * 8), wq8);
      unpack_bf16x8(*(const u32x4*)(a.wk + c * 8), wk8);
      const int g = gc ? (c * 8) / gc : 0;
      const int col = c * 8 - g * gc;
      if (do_q) {
#pragma unroll
        for (int j = 0; j < 8; ++j) q[i][j] = round_bf16(round_bf16(q[i][j] * rq) * wq8[j]);
        rope8_cols(q[i], c * 8, a.hd, c0, c1, pos_f, pos_h, pos_w, a.rope_cs);
        *(u32x4*)(qo + g * a.q_group_stride + col) = pack_bf16x8(q[i]);
      }
      if (do_kv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) k[i][j] = round_bf16(round_bf16(k[i][j] * rk) * wk8[j]);
        rope8_cols(k[i], c * 8, a.hd, c0, c1, pos_f, pos_h, pos_w, a.rope_cs);
        *(u32x4*)(ko + g * a.kv_group_stride + col) = pack_bf16x8(k[i]);
        if (!(a.parts & 4)) *(u32x4*)(vo + g * a.kv_group_stride + col) = vraw[i];
      }
    }
  }
}

// chunks per lane for a row of d elements: the smallest compiled CPL with d <= CPL * 512
#define RTV_ROW_DISPATCH(d, CALL)      \
  do {                                  \
    if ((d) <= 1024) { CALL(2); }       \
    else if ((d) <= 2048) { CALL(4); }  \
    else if ((d) <= 5120) { CALL(10); } \
    else { CALL(16); }                  \
  } while (0)

// ------------------------------------------------------------------ small per-forward tables
__global__ void modulation_table_kernel(const bf16_t* __restrict__ mod, const bf16_t* __restrict__ e0,
                                        bf16_t* __restrict__ emod, int L, int F, int J, int J0, int d) {
  size_t total = (size_t)L * F * J * d;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int c = (int)(i % d);
    size_t t = i / d;
    int j = (int)(t % J);
    t /= J;
    int f = (int)(t % F);
    int l = (int)(t / F);
    float m = bf16_to_f32(mod[((size_t)l * J + j) * d + c]);
    float e = bf16_to_f32(e0[((size_t)f * J0 + (J0 == 1 ? 0 : j)) * d + c]);
    emod[i] = f32_to_bf16(m + e);
  }
}

__global__ void sinusoidal_kernel(const float* __restrict__ t, bf16_t* __restrict__ out, int F, int dim) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int half = dim / 2;
  if (i >= F * half) return;
  int f = i / half, c = i % half;
  double pos = (double)t[f];
  double w = pow(10000.0, -((double)c / (double)half));
  double ang = pos * w;
  out[(size_t)f * dim + c] = f32_to_bf16((float)cos(ang));
  out[(size_t)f * dim + half + c] = f32_to_bf16((float)sin(ang));
}

__global__ void patchify_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ rows, int C, int F,
                                int gh, int gw) {
  // rows[m][c*4 + p*2 + q] = x[c][f][2h+p][2w+q]
  size_t total = (size_t)F * gh * gw * C * 4;
  const int Hh = 2 * gh, Ww = 2 * gw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int kq = (int)(i % (C * 4));
    size_t m = i / (C * 4);
    int c = kq >> 2, p = (kq >> 1) & 1, q = kq & 1;
    int w = (int)(m % gw);
    size_t t = m / gw;
    int h = (int)(t % gh);
    int f = (int)(t / gh);
    rows[i] = x[(((size_t)c * F + f) * Hh + 2 * h + p) * Ww + 2 * w + q];
  }
}

__global__ void unpatchify_kernel(const bf16_t* __restrict__ rows, bf16_t* __restrict__ x, int C, int F,
                                  int gh, int gw) {
  // x[c][f][2h+q][2w+r] = rows[m][(q*2+r)*C + c]
  const int Hh = 2 * gh, Ww = 2 * gw;
  size_t total = (size_t)C * F * Hh * Ww;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int xw = (int)(i % Ww);
    size_t t = i / Ww;
    int xh = (int)(t % Hh);
    t /= Hh;
    int f = (int)(t % F);
    int c = (int)(t / F);
    int h = xh >> 1, q = xh & 1, w = xw >> 1, r = xw & 1;
    size_t m = ((size_t)f * gh + h) * gw + w;
    x[i] = rows[m * (C * 4) + (q * 2 + r) * C + c];
  }
}

}  // namespace rtv

namespace rtv {
// Shared launcher; group_cols == 0 is the plain cache write of the C entry below.
// Argument validation of the launcher below, callable by itself: the DiT sequencer checks a call's ring / row range / exchange
// geometry BEFORE it launches the projection GEMMs whose output the RoPE / cache kernel consumes (the V third of the projection
// writes cache rows directly: a call that is going to be refused must not have touched the cache).
int qk_norm_rope_check(int64_t cache_row_stride, int cache_row0, int M, int d, int num_heads, int F, int gh, int gw, int start_frame,
                       int row_offset, int group_cols, int64_t q_group_stride, int64_t kv_group_stride, int ring_lo, int ring_size,
                       int ring_shift, int parts) {
  if (M <= 0) return 0;
  if (parts < 1 || parts > 7 || parts == 4 || parts == 5) return set_error(-1, "qk_norm_rope_cache: parts must be 1 (q), 2 (k, v) or 3 (+ 4: V in place)");
  if (ring_size < 0 || ring_lo < 0 || ring_shift < 0 || (ring_size > 0 && ring_shift >= ring_size))
    return set_error(-1, "qk_norm_rope_cache: bad ring (need ring_lo >= 0, 0 <= ring_shift < ring_size)");
  if (ring_size > 0 && cache_row0 + row_offset + M > ring_lo + ring_size)
    return set_error(-1, "qk_norm_rope_cache: rows beyond the end of the ring");
  if (num_heads <= 0 || d % num_heads) return set_error(-1, "qk_norm_rope_cache: d % num_heads != 0");
  const int hd = d / num_heads;
  if (d % 8 || d > EW_THREADS * 8 * EW_MAXC || hd % 8 || cache_row_stride % 8)
    return set_error(-1, "qk_norm_rope_cache: alignment (d, head_dim, cache stride multiples of 8; d <= 8192)");
  if (row_offset < 0 || row_offset + M > F * gh * gw) return set_error(-1, "qk_norm_rope_cache: rows outside the F*gh*gw token grid");
  if (start_frame < 0 || start_frame + F > 1024 || gh > 1024 || gw > 1024)
    return set_error(-1, "qk_norm_rope_cache: position exceeds the 1024-entry RoPE table");
  if (cache_row0 < 0) return set_error(-1, "qk_norm_rope_cache: negative cache row");
  if (group_cols && (group_cols % hd || d % group_cols || q_group_stride % 8 || kv_group_stride % 8))
    return set_error(-1, "qk_norm_rope_cache: head groups must be whole heads dividing d");
  return 0;
}

int qk_norm_rope_launch(const void* qkv, void* q_out, void* k_cache, void* v_cache, int64_t cache_row_stride,
                        int cache_row0, int M, int d, int num_heads, float eps, const void* wq, const void* wk,
                        const void* rope_cs, int F, int gh, int gw, int start_frame, int row_offset, int group_cols,
                        int64_t q_group_stride, int64_t kv_group_stride, int ring_lo, int ring_size, int ring_shift,
                        int parts, rtv_stream_t stream) {
  if (M <= 0) return 0;
  if (qk_norm_rope_check(cache_row_stride, cache_row0, M, d, num_heads, F, gh, gw, start_frame, row_offset, group_cols, q_group_stride,
                         kv_group_stride, ring_lo, ring_size, ring_shift, parts))
    return -1;
  const int hd = d / num_heads;
  RopeArgs a;
  a.qkv = (const bf16_t*)qkv;
  a.q_out = (bf16_t*)q_out;
  a.k_cache = (bf16_t*)k_cache;
  a.v_cache = (bf16_t*)v_cache;
  a.cache_row_stride = cache_row_stride;
  a.cache_row0 = cache_row0;
  a.M = M;
  a.d = d;
  a.hd = hd;
  a.eps = eps;
  a.wq = (const bf16_t*)wq;
  a.wk = (const bf16_t*)wk;
  a.rope_cs = (const float2*)rope_cs;
  a.gh = gh;
  a.gw = gw;
  a.start_frame = start_frame;
  a.row_offset = row_offset;
  a.group_cols = group_cols;
  a.q_group_stride = q_group_stride;
  a.kv_group_stride = kv_group_stride;
  a.ring_lo = ring_lo;
  a.ring_size = ring_size;
  a.ring_shift = ring_shift;
  a.parts = parts;
  ProfScope prof(PROF_ROPE, (hipStream_t)stream, ((parts & 1 ? 2.0 : 0.0) + (parts & 2 ? (parts & 4 ? 2.0 : 4.0) : 0.0)) * M * d * 2);
  const int g_sel = g_rope_wave.load(std::memory_order_relaxed);
  if (g_sel == 0 || (g_sel < 0 && M < ROPE_WAVE_MIN_ROWS)) {
    hipLaunchKernelGGL(qk_norm_rope_cache_block_kernel, dim3(M), dim3(EW_THREADS), 0, (hipStream_t)stream, a);
    return check_launch("qk_norm_rope_cache");
  }
  const int waves = M * ((parts & 2) ? 2 : 1);   // two waves per row when k / v are processed (q and k normalise independently)
  const dim3 grid((waves + ROWS_PER_WG - 1) / ROWS_PER_WG);
#define RTV_ROPE_CALL(CPL)                                                                                                  \
  if (512 % hd == 0) hipLaunchKernelGGL((qk_norm_rope_cache_kernel<CPL, true>), grid, dim3(EW_THREADS), 0, (hipStream_t)stream, a); \
  else hipLaunchKernelGGL((qk_norm_rope_cache_kernel<CPL, false>), grid, dim3(EW_THREADS), 0, (hipStream_t)stream, a)
  RTV_ROW_DISPATCH(d, RTV_ROPE_CALL);
#undef RTV_ROPE_CALL
  return check_launch("qk_norm_rope_cache");
}

// in [G][rows][gc]  ->  out [rows][G*gc]   (the inverse of the head-group scatter, for the attention output)
__global__ void regroup_heads_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, int rows, int G, int gc8) {
  const size_t total = (size_t)rows * G * gc8;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % gc8);
    const size_t t = i / gc8;
    const int g = (int)(t % G);
    const size_t r = t / G;
    out[i] = in[((size_t)g * rows + r) * gc8 + c];
  }
}

// pixels fp32 [T][3][H][W] in [-1,1]  ->  rgb8 [T][H][W][3]: u8(trunc(clamp((x + 1) * 0.5, 0, 1) * 255)), the arithmetic of the
// reference's frame path (release_server.py:984 `add_(1.0).mul_(0.5).clamp_(0.0, 1.0)` on the host copy, then
// torchvision to_pil_image = `mul(255).byte()`, :972).  One thread = 4 pixels: 3 float4 plane reads, 12 contiguous bytes out.
__global__ void pixels_to_rgb8_kernel(const float* __restrict__ px, uint8_t* __restrict__ out, int T, size_t hw) {
  const size_t quads = hw >> 2;
  const size_t total = (size_t)T * quads;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t t = i / quads, q = i - t * quads;
    const float* base = px + t * 3 * hw + q * 4;
    float4 c[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) c[ch] = *(const float4*)(base + ch * hw);
    uint8_t b[12];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float v[4] = {c[ch].x, c[ch].y, c[ch].z, c[ch].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float y = __fmul_rn(__fadd_rn(v[k], 1.0f), 0.5f);
        y = fminf(fmaxf(y, 0.0f), 1.0f);
        b[k * 3 + ch] = (uint8_t)(int)__fmul_rn(y, 255.0f);   // y is NaN-free after the clamp only if the input is; NaN -> 0
      }
    }
    uint32_t* o = (uint32_t*)(out + (t * hw + q * 4) * 3);
#pragma unroll
    for (int w = 0; w < 3; ++w)
      o[w] = (uint32_t)b[4 * w] | ((uint32_t)b[4 * w + 1] << 8) | ((uint32_t)b[4 * w + 2] << 16) | ((uint32_t)b[4 * w + 3] << 24);
  }
}

int regroup_heads(const void* in, void* out, int rows, int G, int group_cols, rtv_stream_t stream) {
  if (rows <= 0) return 0;
  if (group_cols % 8) return set_error(-1, "regroup_heads: group_cols % 8 != 0");
  const size_t total = (size_t)rows * G * (group_cols / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  ProfScope prof(PROF_MISC, (hipStream_t)stream, 2.0 * rows * G * group_cols * 2);
  hipLaunchKernelGGL(regroup_heads_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)in, (u32x4*)out,
                     rows, G, group_cols / 8);
  return check_launch("regroup_heads");
}
}  // namespace rtv

using namespace rtv;

extern "C" {

int rtv_rope_set_wave(int mode) {   // include/rtv_hip_lab.h
  g_rope_wave.store(mode < 0 ? -1 : (mode ? 1 : 0), std::memory_order_relaxed);
  return 0;
}


int rtv_layernorm_modulate(const void* x, void* out, int M, int d, float eps, const void* shift,
                           const void* scale, int frame_stride, int rows_per_frame, int row_offset,
                           const void* weight, const void* bias, rtv_stream_t stream) {
  if (M <= 0) return 0;
  if (d % 8 || d > EW_THREADS * 8 * EW_MAXC) return set_error(-1, "layernorm_modulate: d must be a multiple of 8 and <= 8192");
  if ((shift == nullptr) != (scale == nullptr)) return set_error(-1, "layernorm_modulate: shift and scale go together");
  if ((weight == nullptr) != (bias == nullptr)) return set_error(-1, "layernorm_modulate: weight and bias go together");
  if (weight && scale) return set_error(-1, "layernorm_modulate: affine and modulation are exclusive");
  if (scale && (rows_per_frame <= 0 || frame_stride % 8)) return set_error(-1, "layernorm_modulate: bad frame geometry");
  ProfScope prof(PROF_LN, (hipStream_t)stream, 2.0 * M * d * 2);
  hipLaunchKernelGGL(layernorm_modulate_kernel, dim3(M), dim3(EW_THREADS), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)out, d, eps, (const bf16_t*)shift, (const bf16_t*)scale,
                     frame_stride, rows_per_frame, row_offset, (const bf16_t*)weight, (const bf16_t*)bias);
  return check_launch("layernorm_modulate");
}

int rtv_rmsnorm(const void* x, int ldx, void* out, int ldo, int M, int d, float eps, const void* weight,
                rtv_stream_t stream) {
  if (M <= 0) return 0;
  if (d % 8 || d > EW_THREADS * 8 * EW_MAXC || ldx % 8 || ldo % 8)
    return set_error(-1, "rmsnorm: d/ld must be multiples of 8, d <= 8192");
  if (!weight) return set_error(-1, "rmsnorm: weight required");
  ProfScope prof(PROF_LN, (hipStream_t)stream, 2.0 * M * d * 2);
  hipLaunchKernelGGL(rmsnorm_kernel, dim3(M), dim3(EW_THREADS), 0, (hipStream_t)stream, (const bf16_t*)x,
                     ldx, (bf16_t*)out, ldo, d, eps, (const bf16_t*)weight);
  return check_launch("rmsnorm");
}

int rtv_pixels_to_rgb8(const void* pixels, void* rgb8, int T, int H, int W, rtv_stream_t stream) {
  if (T <= 0 || H <= 0 || W <= 0) return 0;
  if (!pixels || !rgb8) return set_error(-1, "pixels_to_rgb8: null argument");
  const size_t hw = (size_t)H * W;
  if (hw % 4 || ((uintptr_t)pixels & 15) || ((uintptr_t)rgb8 & 3))
    return set_error(-1, "pixels_to_rgb8: H*W must be a multiple of 4, pixels 16-byte and rgb8 4-byte aligned");
  const size_t total = (size_t)T * (hw / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  ProfScope prof(PROF_MISC, (hipStream_t)stream, (double)T * hw * 15.0);
  hipLaunchKernelGGL(pixels_to_rgb8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)pixels,
                     (uint8_t*)rgb8, T, hw);
  return check_launch("pixels_to_rgb8");
}

int rtv_qk_norm_rope_cache(const void* qkv, void* q_out, void* k_cache, void* v_cache,
                           int64_t cache_row_stride, int cache_row0, int M, int d, int num_heads, float eps,
                           const void* wq, const void* wk, const void* rope_cs, int F, int gh, int gw,
                           int start_frame, int row_offset, rtv_stream_t stream) {
  return rtv::qk_norm_rope_launch(qkv, q_out, k_cache, v_cache, cache_row_stride, cache_row0, M, d, num_heads, eps, wq, wk,
                                  rope_cs, F, gh, gw, start_frame, row_offset, 0, 0, 0, 0, 0, 0, 3, stream);
}

int rtv_qk_norm_rope_cache_ring(const void* qkv, void* q_out, void* k_cache, void* v_cache,
                                int64_t cache_row_stride, int cache_row0, int M, int d, int num_heads, float eps,
                                const void* wq, const void* wk, const void* rope_cs, int F, int gh, int gw,
                                int start_frame, int row_offset, int ring_lo, int ring_size, int ring_shift,
                                rtv_stream_t stream) {
  return rtv::qk_norm_rope_launch(qkv, q_out, k_cache, v_cache, cache_row_stride, cache_row0, M, d, num_heads, eps, wq, wk,
                                  rope_cs, F, gh, gw, start_frame, row_offset, 0, 0, 0, ring_lo, ring_size, ring_shift, 3, stream);
}

int rtv_modulation_table(const void* modulation, const void* e0, void* emod, int L, int F, int J, int J0,
                         int d, rtv_stream_t stream) {
  if (J0 != J && J0 != 1) return set_error(-1, "modulation_table: J0 must be J or 1");
  size_t total = (size_t)L * F * J * d;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  ProfScope prof(PROF_MISC, (hipStream_t)stream, (double)total * 6);
  hipLaunchKernelGGL(modulation_table_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)modulation, (const bf16_t*)e0, (bf16_t*)emod, L, F, J, J0, d);
  return check_launch("modulation_table");
}

int rtv_sinusoidal_embedding(const void* t, void* out, int F, int dim, rtv_stream_t stream) {
  if (dim % 2) return set_error(-1, "sinusoidal_embedding: dim must be even");
  int n = F * (dim / 2);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sinusoidal_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                     (const float*)t, (bf16_t*)out, F, dim);
  return check_launch("sinusoidal_embedding");
}

int rtv_patchify(const void* x, void* rows, int C, int F, int gh, int gw, rtv_stream_t stream) {
  size_t total = (size_t)F * gh * gw * C * 4;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(patchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)rows, C, F, gh, gw);
  return check_launch("patchify");
}

int rtv_unpatchify(const void* rows, void* x, int C, int F, int gh, int gw, rtv_stream_t stream) {
  size_t total = (size_t)F * gh * gw * C * 4;
  if (total == 0) return 0;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(unpatchify_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)rows, (bf16_t*)x, C, F, gh, gw);
  return check_launch("unpatchify");
}

}  // extern "C"
