// LDS-tiled MFMA GEMM core for gfx950:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// Both operands are K-contiguous (nn.Linear weight layout), 16-bit (bf16 or f16), f32 accumulate.
// Replaces the cuBLAS nn.Linear calls of the reference DiT block
//   (wan/modules/causal_model.py:196-199,:246,:433-435; wan/modules/model.py:184-198).
//
// Structure (per workgroup = WM x WN waves):
//   * A/W tiles are streamed HBM/L2 -> LDS with `global_load_lds_dwordx4` (no VGPR round trip),
//     double-buffered; one barrier per K-step, the next tile's DMA is in flight during the MFMAs.
//   * The DMA writes LDS lane-linearly, so the bank swizzle is applied to the per-lane *source*
//     chunk and again on the ds_read_b128 (same involution on both sides).
//   * MFMA 32x32x16 with swapped operands (W rows as the MFMA "A", activations as "B") so that each
//     lane owns one output row m and 4 consecutive n per register quad -> 8-byte epilogue accesses.
#pragma once
#include "rtv_common.h"

namespace rtv {

struct GemmParams {
  const uint16_t* A;  // [M][lda]
  const uint16_t* W;  // [N][ldw]
  uint16_t* C;        // [M][ldc]
  int lda, ldw, ldc;
  int M, N, K;
  const uint16_t* bias;      // [N] or null
  int act;                   // 0 none, 1 gelu(tanh), 2 silu
  const uint16_t* gate;      // per-frame gate rows or null: gate[(m / rows_per_frame) * gate_stride + n]
  int gate_stride;
  int rows_per_frame;
  int row_offset;            // global index of row 0 (token-axis sharding)
  const uint16_t* residual;  // [M][ldr] or null (may alias C)
  int ldr;
  int tiles_m, tiles_n;
};

template <bool F16>
struct Mfma32;
template <>
struct Mfma32<false> {
  static __device__ __forceinline__ f32x16 run(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(uint16_t v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ uint16_t from_f32(float v) { return f32_to_bf16(v); }
  static __device__ __forceinline__ float round(float v) { return round_bf16(v); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
  static __device__ __forceinline__ float lo_f32(uint32_t u) { return __uint_as_float(u << 16); }
  static __device__ __forceinline__ float hi_f32(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
};
template <>
struct Mfma32<true> {
  static __device__ __forceinline__ f32x16 run(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ float to_f32(uint16_t v) { return f16_to_f32(v); }
  static __device__ __forceinline__ uint16_t from_f32(float v) { return f32_to_f16(v); }
  static __device__ __forceinline__ float round(float v) { return round_f16(v); }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
  static __device__ __forceinline__ float lo_f32(uint32_t u) { return f16_to_f32((uint16_t)(u & 0xffff)); }
  static __device__ __forceinline__ float hi_f32(uint32_t u) { return f16_to_f32((uint16_t)(u >> 16)); }
};

// LDS tile geometry shared by the GEMM and the implicit-GEMM conv kernels.
template <int BM, int BN, int BK, int WM, int WN>
struct TileCfg {
  static constexpr int NW = WM * WN;
  static constexpr int NT = NW * 64;
  static constexpr int CH = BK / 8;              // 16-byte chunks per tile row
  static constexpr int RPI = 64 / CH;            // rows written by one wave-wide DMA instruction
  static constexpr int RPB = 16 / CH;            // tile rows per 256-byte LDS bank row
  static constexpr int A_INST = BM / RPI / NW;   // DMA instructions per wave per stage (A)
  static constexpr int B_INST = BN / RPI / NW;
  static constexpr int A_BYTES = BM * BK * 2;
  static constexpr int B_BYTES = BN * BK * 2;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int TM = BM / WM / 32;        // 32x32 blocks per wave
  static constexpr int TN = BN / WN / 32;
  static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "tile/wave mismatch");
  static_assert(BK % 16 == 0 && (CH == 4 || CH == 8), "BK must be 32 or 64");
  // swizzled chunk position inside a row (involution)
  static __device__ __forceinline__ int swz(int row, int chunk) {
    return chunk ^ ((row / RPB) & (CH - 1));
  }
};

// one wave-wide 16-byte-per-lane global -> LDS DMA. `lds_wave_base` must be wave-uniform.
__device__ __forceinline__ void dma16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const RTV_GLOBAL void*)gsrc, (RTV_LDS void*)lds_wave_base, 16, 0,
                                   0);
}

// MFMA over one staged K-slab: acc[mi][ni] += W_tile . A_tile^T   (swapped operands)
template <bool F16, typename Cfg, int BK>
__device__ __forceinline__ void mma_stage(const char* sA, const char* sB, int a_row0, int b_row0,
                                          int lane, f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  const int l31 = lane & 31, g = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < BK / 16; ++ks) {
    u32x4 af[Cfg::TM], bf[Cfg::TN];
#pragma unroll
    for (int mi = 0; mi < Cfg::TM; ++mi) {
      int row = a_row0 + mi * 32 + l31;
      af[mi] = *(const u32x4*)(sA + row * (BK * 2) + Cfg::swz(row, ks * 2 + g) * 16);
    }
#pragma unroll
    for (int ni = 0; ni < Cfg::TN; ++ni) {
      int row = b_row0 + ni * 32 + l31;
      bf[ni] = *(const u32x4*)(sB + row * (BK * 2) + Cfg::swz(row, ks * 2 + g) * 16);
    }
#pragma unroll
    for (int mi = 0; mi < Cfg::TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < Cfg::TN; ++ni)
        acc[mi][ni] = Mfma32<F16>::run(bf[ni], af[mi], acc[mi][ni]);
  }
}

// Fused epilogue for one lane-owned quad of 4 consecutive n at row m.
// Rounding points reproduce the reference's bf16 eager chain:
//   y = bf16(acc + bias); y = bf16(act(y)); t = bf16(y * gate); out = bf16(res + t)
template <bool F16>
__device__ __forceinline__ void epilogue_quad(const GemmParams& p, int m, int n, const float* v4) {
  typedef Mfma32<F16> T;
  float v[4] = {v4[0], v4[1], v4[2], v4[3]};
  if (p.bias) {
    u32x2 b = *(const u32x2*)(p.bias + n);
    v[0] += T::to_f32(b[0] & 0xffff);
    v[1] += T::to_f32(b[0] >> 16);
    v[2] += T::to_f32(b[1] & 0xffff);
    v[3] += T::to_f32(b[1] >> 16);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = T::round(v[i]);
  if (p.act == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = T::round(gelu_tanh(v[i]));
  } else if (p.act == 2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = T::round(silu(v[i]));
  }
  if (p.gate) {
    const uint16_t* gp = p.gate + (size_t)((p.row_offset + m) / p.rows_per_frame) * p.gate_stride + n;
    u32x2 g = *(const u32x2*)gp;
    v[0] = T::round(v[0] * T::to_f32(g[0] & 0xffff));
    v[1] = T::round(v[1] * T::to_f32(g[0] >> 16));
    v[2] = T::round(v[2] * T::to_f32(g[1] & 0xffff));
    v[3] = T::round(v[3] * T::to_f32(g[1] >> 16));
  }
  if (p.residual) {
    u32x2 r = *(const u32x2*)(p.residual + (size_t)m * p.ldr + n);
    v[0] += T::to_f32(r[0] & 0xffff);
    v[1] += T::to_f32(r[0] >> 16);
    v[2] += T::to_f32(r[1] & 0xffff);
    v[3] += T::to_f32(r[1] >> 16);
  }
  u32x2 o;
  o[0] = (uint32_t)T::from_f32(v[0]) | ((uint32_t)T::from_f32(v[1]) << 16);
  o[1] = (uint32_t)T::from_f32(v[2]) | ((uint32_t)T::from_f32(v[3]) << 16);
  *(u32x2*)(p.C + (size_t)m * p.ldc + n) = o;
}

template <bool F16, typename Cfg>
__device__ __forceinline__ void store_tile(const GemmParams& p, int m_base, int n_base, int lane,
                                           f32x16 (&acc)[Cfg::TM][Cfg::TN]) {
  const int l31 = lane & 31, g = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < Cfg::TM; ++mi) {
    int m = m_base + mi * 32 + l31;
    if (m >= p.M) continue;
#pragma unroll
    for (int ni = 0; ni < Cfg::TN; ++ni) {
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        int n = n_base + ni * 32 + rq * 8 + g * 4;
        if (n < p.N) {
          float v4[4] = {acc[mi][ni][rq * 4 + 0], acc[mi][ni][rq * 4 + 1], acc[mi][ni][rq * 4 + 2],
                         acc[mi][ni][rq * 4 + 3]};
          epilogue_quad<F16>(p, m, n, v4);
        }
      }
    }
  }
}

// Epilogue through LDS: coalesced global accesses for the output and the fused residual.
//
// The MFMA register layout gives every lane ONE output row and 4 consecutive columns per quad, so storing it directly
// (store_tile above) issues 8-byte accesses at a row stride: 32 different cache lines per instruction.  Measured on the
// 256x256 tile (scripts/gemm_lab.py, profiles/r02_gemm_lab_v2_vs_v1.log): that store tail costs 85-130k cycles per tile,
// a third of the kernel.  Here each wave first writes its TM*32 x 64 block to a wave-private LDS image of 128-byte rows
// (16-byte chunk c of row r at chunk c ^ (r & 7), its 8-byte halves swapped when (r >> 3) & 1: conflict-free ds_write_b64 and
// ds_read_b128), then every lane reads 16 contiguous bytes of a row, so residual loads and output stores are whole
// 128-byte lines per 8 lanes: 18k cycles per tile.
// Same rounding points as epilogue_quad (y = bf16(acc + bias); y = bf16(act(y)); t = bf16(y * gate); out = bf16(res + t)):
// a value is only rounded when another operation follows; the staged value is t, the residual is added in f32 afterwards.
// `img`: TM*32*128 bytes of LDS owned by this wave; the caller guarantees nobody still reads that LDS (barrier).
template <bool F16, int TM, int ACT, bool GATE>
__device__ __forceinline__ void store_tile_lds_impl(const GemmParams& p, int m_base, int n_base, int lane, char* img,
                                                    f32x16 (&acc)[TM][2], unsigned long long* mid_stamp) {
  typedef Mfma32<F16> T;
  const int l31 = lane & 31, g = lane >> 5;
  // the bias (and gate) quads of this lane's 8 column slots, fetched as ONE batch of loads up front: left inside the loop
  // they are 32 dependent round trips to L2 per lane (measured 7 us of an 10 us epilogue, profiles/r02_gemm_timeline_*.log)
  int ncol[8];
  u32x2 bq[8];
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {
    ncol[s8] = min(n_base + (s8 >> 2) * 32 + (s8 & 3) * 8 + g * 4, p.N - 4);
    bq[s8] = p.bias ? *(const u32x2*)(p.bias + ncol[s8]) : u32x2{0u, 0u};
  }
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int row = mi * 32 + l31;
    u32x2 gq[8];
    if (GATE) {
      const uint16_t* gp = p.gate + (size_t)((p.row_offset + min(m_base + row, p.M - 1)) / p.rows_per_frame) * p.gate_stride;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) gq[s8] = *(const u32x2*)(gp + ncol[s8]);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int s8 = ni * 4 + rq;
        float v[4] = {acc[mi][ni][rq * 4 + 0], acc[mi][ni][rq * 4 + 1], acc[mi][ni][rq * 4 + 2], acc[mi][ni][rq * 4 + 3]};
        v[0] += T::lo_f32(bq[s8][0]);   // (zero quads without a bias: + 0 is exact)
        v[1] += T::hi_f32(bq[s8][0]);
        v[2] += T::lo_f32(bq[s8][1]);
        v[3] += T::hi_f32(bq[s8][1]);
        if (ACT == 1) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = gelu_tanh(T::round(v[i]));
        } else if (ACT == 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = silu(T::round(v[i]));
        }
        if (GATE) {
          v[0] = T::round(v[0]) * T::lo_f32(gq[s8][0]);
          v[1] = T::round(v[1]) * T::hi_f32(gq[s8][0]);
          v[2] = T::round(v[2]) * T::lo_f32(gq[s8][1]);
          v[3] = T::round(v[3]) * T::hi_f32(gq[s8][1]);
        }
        u32x2 o;
        o[0] = T::pack2(v[0], v[1]);
        o[1] = T::pack2(v[2], v[3]);
        const int chunk = s8 ^ (row & 7);
        const int half = g ^ ((row >> 3) & 1);
        *(u32x2*)(img + row * 128 + chunk * 16 + half * 8) = o;
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  if (mid_stamp) *mid_stamp = __builtin_amdgcn_s_memrealtime();   // lab builds: end of the register -> LDS pass
  const int rsub = lane >> 3, c = lane & 7;
  const int n = n_base + c * 8;
  const bool n_ok = n < p.N;
  constexpr int PASSES = TM * 4;   // 8 rows per pass
  u32x4 res[PASSES];
  if (p.residual) {
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = min(m_base + ps * 8 + rsub, p.M - 1);
      res[ps] = *(const u32x4*)(p.residual + (size_t)m * p.ldr + min(n, p.N - 8));
    }
  }
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int row = ps * 8 + rsub;
    const int m = m_base + row;
    u32x4 t = *(const u32x4*)(img + row * 128 + ((c ^ (row & 7)) << 4));
    if ((row >> 3) & 1) t = u32x4{t[2], t[3], t[0], t[1]};
    if (p.residual) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        t[i] = T::pack2(T::lo_f32(t[i]) + T::lo_f32(res[ps][i]), T::hi_f32(t[i]) + T::hi_f32(res[ps][i]));
    }
    if (m < p.M && n_ok) *(u32x4*)(p.C + (size_t)m * p.ldc + n) = t;
  }
}

// One branch on the epilogue kind, then straight-line code: with the activation / gate tests inside the 32 quads of a
// lane the epilogue was 100+ tiny basic blocks that the scheduler could not overlap (7 us of the 10 us epilogue).
template <bool F16, int TM>
__device__ __forceinline__ void store_tile_lds(const GemmParams& p, int m_base, int n_base, int lane, char* img,
                                               f32x16 (&acc)[TM][2], unsigned long long* mid_stamp = nullptr) {
  if (p.gate) {
    if (p.act == 0) store_tile_lds_impl<F16, TM, 0, true>(p, m_base, n_base, lane, img, acc, mid_stamp);
    else if (p.act == 1) store_tile_lds_impl<F16, TM, 1, true>(p, m_base, n_base, lane, img, acc, mid_stamp);
    else store_tile_lds_impl<F16, TM, 2, true>(p, m_base, n_base, lane, img, acc, mid_stamp);
  } else {
    if (p.act == 0) store_tile_lds_impl<F16, TM, 0, false>(p, m_base, n_base, lane, img, acc, mid_stamp);
    else if (p.act == 1) store_tile_lds_impl<F16, TM, 1, false>(p, m_base, n_base, lane, img, acc, mid_stamp);
    else store_tile_lds_impl<F16, TM, 2, false>(p, m_base, n_base, lane, img, acc, mid_stamp);
  }
}

}  // namespace rtv
