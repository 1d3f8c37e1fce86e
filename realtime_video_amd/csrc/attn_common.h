// Shared by the attention kernels (attn_fwd.hip: lockstep + four-phase kernels and the launcher; attn_w4.hip: the one-wave-per-SIMD
// kernel): launch parameters, tile constants, MFMA / LDS-read helpers, the KV-split partial store.
#pragma once
#include "rtv_common.h"
#include "rtv_internal.h"

namespace rtv {

struct AttnParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  int B, Lq, Lkv, H;
  int64_t q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;  // element strides; head stride = 128
  float scale_log2e;
  int causal_block, q_offset;
  int n_qtiles;
  // two-segment key window (ring-indexed rolling KV cache, causal_model.py:363-379 without the shift copy): key v of the
  // window is cache row v for v < n0 and row v + delta for v >= n0 (rows relative to k / v); n0 == Lkv: one segment.
  int n0, delta;
  int off0;   // four-phase kernel only: keys v < n0 are cache rows v + off0 (its k / v point at the lowest row of the window)
  int skip_idle;  // four-phase kernel only: waves whose 32 query rows all lie beyond Lq run the idle loop (A/B switch)
  // lockstep kernel only: key `dup_key` stands for dup_count identical keys (the zero-padded text rows of the cross-attention all
  // have the same K and V): its score gets + dup_bias = log2(dup_count) / scale_log2e before the softmax.  -1: none.
  int dup_key;
  float dup_bias;
  // KV split (flash-decoding style, for launches whose query grid cannot fill the chip: the head-parallel phase of a context-
  // parallel rank has 5 heads x 19 query tiles): blockIdx.y = split s works on key tiles [ntiles*s/S, ntiles*(s+1)/S) of the
  // workgroup's OWN tile count (block-causal: the tiles below its largest key limit) and leaves its UNNORMALISED O (fp32), its
  // reference point m and its row sum l in part_o / part_ml; attn_combine_kernel merges.  A row whose keys in a range are all
  // masked leaves (m, l, O) = (-1e30, 0, 0): weight 2^(-1e30 - m) = 0 in the merge.
  int kv_splits;    // S (1: the kernel writes `o` itself)
  float* part_o;    // [S][B*H][Lq][128]
  float* part_ml;   // [S][B*H][Lq][2] = (m, l)
};

constexpr int ATT_D = 128;
constexpr int ATT_QW = 32;            // query rows per wave
constexpr int ATT_KT = 64;            // keys per tile
constexpr int ATT_TILE_BYTES = ATT_KT * ATT_D * 2;  // 16 KiB

template <int V>
struct IntC {
  static constexpr int value = V;
};

template <bool F16>
__device__ __forceinline__ f32x16 mfma32(const u32x4& a, const u32x4& b, const f32x16& c) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b),
                                                  c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ uint32_t pack2(float a, float b) {
  if constexpr (F16) return pack_f16x2(a, b);
  else return pack_bf16x2(a, b);
}

__device__ __forceinline__ u32x2 lds_tr_read(const char* p) {
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTV_LDS s16x4*)p);
  return __builtin_bit_cast(u32x2, t);
}

// KV split: this workgroup's share of the key tiles (wave-uniform).
__device__ __forceinline__ void split_tile_range(const AttnParams& p, int ntiles, int* t_lo, int* t_hi) {
  *t_lo = 0;
  *t_hi = ntiles;
  if (p.kv_splits > 1) {
    const int s = blockIdx.y;
    *t_lo = ntiles * s / p.kv_splits;
    *t_hi = ntiles * (s + 1) / p.kv_splits;
  }
}

// KV split epilogue: O^T unnormalised, lane owns row q and dims db*32 + 8*i + 4*g + {0..3}; (m, l) once per row.
__device__ __forceinline__ void store_partial(const AttnParams& p, int bh, int q_row, int g, const f32x16 (&oacc)[4], float m_run,
                                              float l_tot) {
  if (q_row >= p.Lq) return;
  const size_t row = ((size_t)blockIdx.y * (p.B * p.H) + bh) * p.Lq + q_row;
  float* po = p.part_o + row * ATT_D + 4 * g;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *(f32x4*)(po + db * 32 + i * 8) = f32x4{oacc[db][4 * i], oacc[db][4 * i + 1], oacc[db][4 * i + 2], oacc[db][4 * i + 3]};
  if (g == 0) *(f32x2*)(p.part_ml + row * 2) = f32x2{m_run, l_tot};
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}
// Transpose read as inline asm: the builtin carries no pointer information, so behind a pending LDS DMA the compiler's
// wait-count pass puts `s_waitcnt vmcnt(0)` in front of it (a whole DMA latency per tile).  The asm form is invisible to
// that pass - the consumer side waits with lds_wait_frags() below.
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr_read_at(uint32_t lds_addr) {
  u32x2 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128_at(uint32_t lds_addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
  return r;
}
// s_waitcnt lgkmcnt(0) that the fragment registers depend on (so no consumer can be scheduled above it)
__device__ __forceinline__ void lds_wait_frags(u32x4 (&f)[16]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]));
  asm volatile("" : "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]));
}
// the value of lane ^ 32 without the LDS crossbar (ds_bpermute would queue behind the transpose reads in flight)
__device__ __forceinline__ float xor32_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);   // {lanes: [lo, lo], [hi, hi]}
  return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}


}  // namespace rtv
