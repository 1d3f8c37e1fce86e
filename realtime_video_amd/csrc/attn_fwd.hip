// Flash-style attention forward for gfx950 (head_dim 128, bf16/f16, BLHD, in-place strided K/V).
//
// Replaces the attention backend plugin point wan/modules/attention.py:150-212 (sageattn /
// flash_attn / SDPA), the flex_attention block-causal call causal_model.py:339-348 and the
// cross-attention call model.py:201-223.  K/V are read in place from the rolling KV cache
// (strided BLHD view, causal_model.py:386-390): no transposes, no contiguous copies.
//
// Workgroup = 8 waves, 256 query rows (32 per wave) of one head; KV tiles of 64 keys stream
// HBM -> registers -> LDS (double buffered, issue-early / write-late), shared by the 8 waves.
// A 4-wave / 128-row build of the same kernel serves launches whose 256-row grid cannot fill the
// 256 CUs (the head-sharded attention of the context-parallel path: M rows x H/world heads).
//   S^T = K . Q^T       (MFMA 32x32x16, K rows as the A operand from XOR-swizzled LDS, Q^T kept in
//                        registers) -> every lane owns ONE query column: the softmax row reductions
//                        are in-lane plus one exchange with lane^32.
//   O^T += V^T . P^T    P^T is the lane's own S^T registers converted to 16-bit (no cross-lane
//                        movement); V^T fragments come from row-major V tiles through the gfx950
//                        LDS transpose read (ds_read_b64_tr_b16).
// Online softmax in the exp2 domain, f32 accumulation; per-row prefix limits implement the
// block-causal mask of the KV-recompute pass without materialising a mask.
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

// Occupancy: ~220 unified registers -> 2 waves per SIMD (the second __launch_bounds__ argument), i.e. ONE 8-wave workgroup or
// two 4-wave workgroups per CU.  A 6-wave build cannot help partial rounds: its workgroup would still occupy a CU for the
// time its two doubly-loaded SIMDs need.
template <bool F16, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(AttnParams p) {
  constexpr int ATT_QT = ATT_QW * NW;                    // query rows per workgroup (256 / 128)
  constexpr int ATT_THREADS = NW * 64;
  constexpr int ATT_LD_PER_THREAD = ATT_KT * 16 / ATT_THREADS;  // 16-byte chunks per thread per tile (2 / 4)
  constexpr int ATT_RSTEP = ATT_THREADS / 16;            // tile rows between a thread's chunks (32 / 16)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // smem: K[2][16 KiB] | V[2][16 KiB]
  char* const sK = smem;
  char* const sV = smem + 2 * ATT_TILE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;

  // ---- workgroup -> (batch*head, q tile).  With H % 8 == 0 all q tiles of a head run on ONE XCD
  //      (block b executes on XCD b % 8) so the head's K/V stream is shared through that XCD's L2.
  int bh, qt;
  {
    const int nbh = p.B * p.H;
    const int bid = blockIdx.x;
    if (nbh % 8 == 0) {
      int xcd = bid & 7, slot = bid >> 3;
      bh = xcd + 8 * (slot / p.n_qtiles);
      qt = slot % p.n_qtiles;
    } else {
      bh = bid / p.n_qtiles;
      qt = bid % p.n_qtiles;
    }
  }
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = qt * ATT_QT;

  const uint16_t* qb = p.q + (size_t)b * p.q_bs + (size_t)h * ATT_D;
  const uint16_t* kb = p.k + (size_t)b * p.k_bs + (size_t)h * ATT_D;
  const uint16_t* vb = p.v + (size_t)b * p.v_bs + (size_t)h * ATT_D;
  uint16_t* ob = p.o + (size_t)b * p.o_bs + (size_t)h * ATT_D;

  // ---- Q^T fragments (MFMA B operand): lane holds Q[q][dc*16 + g*8 .. +8]
  const int q_row = q0 + wave * ATT_QW + l31;
  const int q_row_c = min(q_row, p.Lq - 1);
  u32x4 qf[8];
  {
    const uint16_t* qp = qb + (size_t)q_row_c * p.q_rs + g * 8;
#pragma unroll
    for (int dc = 0; dc < 8; ++dc) qf[dc] = *(const u32x4*)(qp + dc * 16);
  }

  // ---- key-prefix limits (block-causal rule kv < ends[q], causal_model.py:134-136)
  int kv_lim = p.Lkv;        // this lane's row
  int wave_min_lim = p.Lkv;  // smallest limit inside this wave (wave-uniform)
  int wg_max_lim = p.Lkv;    // largest limit inside this workgroup (workgroup-uniform)
  if (p.causal_block > 0) {
    const int cb = p.causal_block;
    kv_lim = min(p.Lkv, ((p.q_offset + q_row_c) / cb + 1) * cb);
    int first = min(q0 + wave * ATT_QW, p.Lq - 1);
    wave_min_lim = min(p.Lkv, ((p.q_offset + first) / cb + 1) * cb);
    int last = min(q0 + ATT_QT - 1, p.Lq - 1);
    wg_max_lim = min(p.Lkv, ((p.q_offset + last) / cb + 1) * cb);
  }
  const int ntiles = (wg_max_lim + ATT_KT - 1) / ATT_KT;

  // ---- staging geometry: thread moves chunks id = tid + i*THREADS -> (row = id>>4 = tid>>4 + RSTEP i, chunk = tid&15).
  //      Full tiles: uniform 64-bit base (SGPR, advanced per tile / per i) + one 32-bit lane offset per operand,
  //      so the loads cost no per-tile VALU address arithmetic; the ragged last tile clamps rows per lane.
  u32x4 kreg[ATT_LD_PER_THREAD], vreg[ATT_LD_PER_THREAD];
  const int st_r = tid >> 4, st_c = tid & 15;
  const uint32_t k_goff = (uint32_t)(st_r * (int)p.k_rs + st_c * 8) * 2u;
  const uint32_t v_goff = (uint32_t)(st_r * (int)p.v_rs + st_c * 8) * 2u;
  auto load_tile = [&](int j) {
    const int row0 = j * ATT_KT;
    const bool in0 = row0 + ATT_KT <= p.n0;
    if (in0 || (row0 >= p.n0 && row0 + ATT_KT <= p.Lkv)) {
      const int64_t prow = in0 ? row0 : row0 + p.delta;
      const char* kj = (const char*)(kb + prow * p.k_rs);
      const char* vj = (const char*)(vb + prow * p.v_rs);
#pragma unroll
      for (int i = 0; i < ATT_LD_PER_THREAD; ++i) {
        kreg[i] = *(const u32x4*)(kj + (size_t)(i * ATT_RSTEP) * p.k_rs * 2 + k_goff);
        vreg[i] = *(const u32x4*)(vj + (size_t)(i * ATT_RSTEP) * p.v_rs * 2 + v_goff);
      }
    } else {
#pragma unroll
      for (int i = 0; i < ATT_LD_PER_THREAD; ++i) {
        int kv = min(row0 + st_r + i * ATT_RSTEP, p.Lkv - 1);
        if (kv >= p.n0) kv += p.delta;
        kreg[i] = *(const u32x4*)(kb + (int64_t)kv * p.k_rs + st_c * 8);
        vreg[i] = *(const u32x4*)(vb + (int64_t)kv * p.v_rs + st_c * 8);
      }
    }
  };
  // K: 16-byte chunk index XOR (row & 15)  -> conflict-free ds_read_b128 column reads
  // V: 64-byte group index XOR (row & 3)   -> conflict-free transpose reads
  // (row & 15 and row & 3 do not depend on i: rows advance by 32 or 16)
  char* const k_wr = sK + st_r * 256 + ((st_c ^ (st_r & 15)) << 4);
  char* const v_wr = sV + st_r * 256 + ((st_c << 4) ^ ((st_r & 3) << 6));
  auto write_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < ATT_LD_PER_THREAD; ++i) {
      *(u32x4*)(k_wr + buf * ATT_TILE_BYTES + i * ATT_RSTEP * 256) = kreg[i];
      *(u32x4*)(v_wr + buf * ATT_TILE_BYTES + i * ATT_RSTEP * 256) = vreg[i];
    }
  };

  // ---- per-lane LDS read addresses, hoisted out of the tile loop (tile buffer / key block / k-step are
  //      compile-time offsets that fold into the ds_read immediate)
  // K operand (A): row = kbk*32 + l31, chunk = (dc*2 + g) ^ (row & 15)
  const char* k_rd[8];
#pragma unroll
  for (int dc = 0; dc < 8; ++dc) k_rd[dc] = sK + l31 * 256 + (((dc * 2 + g) ^ (l31 & 15)) << 4);
  // V^T operand (A) via transpose read: 16-lane group gathers a [4 keys][16 dims] block
  const int i16 = lane & 15, h16 = (lane >> 4) & 1;
  const char* v_rd[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
    v_rd[db] = sV + (4 * g + (i16 >> 2)) * 256 + h16 * 32 + (i16 & 3) * 8 + ((db ^ (i16 >> 2)) << 6);

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -1e30f;  // reference point of the exponentials (>= running max - RESCALE_SLACK), log2 domain
  float l_run = 0.f;     // this lane's partial row sum
  const float c = p.scale_log2e;
  const f32x2 c2 = {c, c};
  // Lazy rescaling: O and l of a row are only rescaled when that row's maximum has grown by more than 2^8 since its
  // last rescale (a rare branch after the first tiles).  The result is the same softmax — any reference
  // point cancels in O/l — with P <= 256 in the 16-bit MFMA operand and f32 accumulators.
  constexpr float RESCALE_SLACK = 8.f;

  // (A staggered schedule - waves 4-7 half a tile behind waves 0-3 - measured 2-3 % slower than this plain one,
  // profiles/r01_attn_variants.txt: the compiler already interleaves the exponentials of tile j with the PV MFMAs.)
  auto tile = [&](const int j, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const bool has_next = (j + 1 < ntiles);
    if (has_next) load_tile(j + 1);  // in flight during the MFMAs below

    // ---------------- S^T = K . Q^T   (two independent accumulator chains, interleaved)
    f32x16 sacc[2];
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kbk][r] = 0.f;
#pragma unroll
    for (int dc = 0; dc < 8; ++dc)
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk) {
        u32x4 kf = *(const u32x4*)(k_rd[dc] + buf * ATT_TILE_BYTES + kbk * 32 * 256);
        sacc[kbk] = mfma32<F16>(kf, qf[dc], sacc[kbk]);
      }

    // ---------------- mask (only on tiles that cross a limit of this wave)
    if ((j + 1) * ATT_KT > wave_min_lim) {
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int kv = j * ATT_KT + kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (kv >= kv_lim) sacc[kbk][r] = -INFINITY;
        }
    }

    // ---------------- online softmax (row = lane's query; the row is split over lanes l and l^32)
    float mx = sacc[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_cand = fmaxf(m_run, mx * c);
    const bool need = m_cand - m_run > RESCALE_SLACK;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {
      // the branch is wave-uniform, the decision per row: rows that do not need it multiply by exp2(0) = 1 exactly, so a
      // row's arithmetic never depends on which other rows share its wave (token-sharded == unsharded, bit for bit)
      const float m_new = need ? m_cand : m_run;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
    }
    const f32x2 nm2 = {-m_run, -m_run};
    f32x2 ps2 = {0.f, 0.f};
    u32x4 pf[2][2];  // [kv block][k-step]: 8 x 16-bit = the lane's own 8 keys of that 16-key step
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        f32x2 x = {sacc[kbk][2 * t], sacc[kbk][2 * t + 1]};
        x = __builtin_elementwise_fma(x, c2, nm2);  // v_pk_fma_f32
        f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        ps2 += e;                                   // v_pk_add_f32
        pf[kbk][t >> 2][t & 3] = pack2<F16>(e[0], e[1]);
      }
    l_run += ps2[0] + ps2[1];

    // ---------------- O^T += V^T . P^T
#pragma unroll
    for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const char* vp = v_rd[db] + buf * ATT_TILE_BYTES + (kbk * 32 + s * 16) * 256;
          u32x2 lo = lds_tr_read(vp);            // keys +0..3  (this lane group's first quad)
          u32x2 hi = lds_tr_read(vp + 8 * 256);  // keys +8..11
          u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
          oacc[db] = mfma32<F16>(vf, pf[kbk][s], oacc[db]);
        }
      }

    if (has_next) write_tile(buf ^ 1);
    __syncthreads();
  };

  load_tile(0);
  write_tile(0);
  __syncthreads();
  for (int j = 0; j < ntiles; j += 2) {
    tile(j, IntC<0>{});
    if (j + 1 < ntiles) tile(j + 1, IntC<1>{});
  }

  // ---------------- epilogue: O = O^T / l, lane owns row q and dims db*32 + 8*i + 4*g + {0..3}
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q_row < p.Lq) {
    uint16_t* op = ob + (size_t)q_row * p.o_rs + 4 * g;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x2 w;
        w[0] = pack2<F16>(oacc[db][4 * i + 0] * inv, oacc[db][4 * i + 1] * inv);
        w[1] = pack2<F16>(oacc[db][4 * i + 2] * inv, oacc[db][4 * i + 3] * inv);
        *(u32x2*)(op + db * 32 + i * 8) = w;
      }
  }
}

}  // namespace rtv

using namespace rtv;

static int g_attn_waves = 0;  // 0 = by grid size

extern "C" int rtv_attn_set_waves(int waves) {
  if (waves != 0 && waves != 4 && waves != 8) return set_error(-1, "attn_set_waves: 0 (auto), 4 or 8");
  g_attn_waves = waves;
  return 0;
}

extern "C" int rtv_attn_fwd(const void* q, const void* k, const void* v, void* o, int B, int Lq, int Lkv,
                            int H, int D, int64_t q_batch_stride, int64_t q_row_stride,
                            int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride,
                            int64_t v_row_stride, int64_t o_batch_stride, int64_t o_row_stride, float scale,
                            int causal_block, int q_offset, int dtype, rtv_stream_t stream) {
  return rtv_attn_fwd_win(q, k, v, o, B, Lq, Lkv, 0, 0, H, D, q_batch_stride, q_row_stride, k_batch_stride, k_row_stride,
                          v_batch_stride, v_row_stride, o_batch_stride, o_row_stride, scale, causal_block, q_offset, dtype,
                          stream);
}

extern "C" int rtv_attn_fwd_win(const void* q, const void* k, const void* v, void* o, int B, int Lq, int Lkv0, int Lkv1,
                                int seg1_row, int H, int D, int64_t q_batch_stride, int64_t q_row_stride,
                                int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride,
                                int64_t v_row_stride, int64_t o_batch_stride, int64_t o_row_stride, float scale,
                                int causal_block, int q_offset, int dtype, rtv_stream_t stream) {
  if (Lkv0 < 0 || Lkv1 < 0) return set_error(-1, "attn_fwd: negative segment length");
  if (Lkv1 > 0 && causal_block > 0) return set_error(-1, "attn_fwd: the block-causal mask needs a one-segment window");
  const int Lkv = Lkv0 + Lkv1;
  if (D != ATT_D) return set_error(-1, "attn_fwd: head_dim must be 128");
  if (B <= 0 || H <= 0 || Lq <= 0) return 0;
  if (Lkv <= 0) return set_error(-1, "attn_fwd: Lkv must be positive");
  if ((q_row_stride | k_row_stride | v_row_stride | o_row_stride | q_batch_stride | k_batch_stride |
       v_batch_stride | o_batch_stride) & 7)
    return set_error(-1, "attn_fwd: strides must be multiples of 8 elements (16-byte rows)");
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15)
    return set_error(-1, "attn_fwd: base pointers must be 16-byte aligned");
  if (causal_block < 0 || q_offset < 0) return set_error(-1, "attn_fwd: negative mask parameters");
  if (dtype != RTV_DTYPE_BF16 && dtype != RTV_DTYPE_F16) return set_error(-1, "attn_fwd: dtype");
  AttnParams p;
  p.q = (const uint16_t*)q;
  p.k = (const uint16_t*)k;
  p.v = (const uint16_t*)v;
  p.o = (uint16_t*)o;
  p.B = B;
  p.Lq = Lq;
  p.Lkv = Lkv;
  p.H = H;
  p.q_bs = q_batch_stride;
  p.q_rs = q_row_stride;
  p.k_bs = k_batch_stride;
  p.k_rs = k_row_stride;
  p.v_bs = v_batch_stride;
  p.v_rs = v_row_stride;
  p.o_bs = o_batch_stride;
  p.o_rs = o_row_stride;
  p.scale_log2e = scale * 1.4426950408889634f;
  p.causal_block = causal_block;
  p.q_offset = q_offset;
  p.n0 = Lkv1 > 0 ? Lkv0 : Lkv;
  p.delta = Lkv1 > 0 ? seg1_row - Lkv0 : 0;
  // 256-row workgroups unless their grid leaves most CUs idle: then 128-row ones.  Measured (MI355X, Lkv 14040, ms for 8 / 4
  // waves): 4680 rows x 20 heads 0.90 / 0.88, x 10 heads (190 workgroups) 0.46 / 0.51, x 5 heads (95) 0.40 / 0.32,
  // 585 rows x 40 heads (120) 0.40 / 0.34.
  int waves = g_attn_waves;
  if (waves == 0) waves = (int64_t)B * H * ((Lq + 255) / 256) < 160 ? 4 : 8;
  const int qt_rows = ATT_QW * waves;
  p.n_qtiles = (Lq + qt_rows - 1) / qt_rows;
  const int lds = 4 * ATT_TILE_BYTES;
  const bool f16 = dtype == RTV_DTYPE_F16;
  const void* kerns[4] = {(const void*)attn_fwd_kernel<false, 8>, (const void*)attn_fwd_kernel<true, 8>,
                          (const void*)attn_fwd_kernel<false, 4>, (const void*)attn_fwd_kernel<true, 4>};
  static bool attr_set[4] = {false, false, false, false};
  const int ki = (waves == 4 ? 2 : 0) + (f16 ? 1 : 0);
  if (!attr_set[ki]) {
    hipError_t e = hipFuncSetAttribute(kerns[ki], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(e, "attn_fwd: hipFuncSetAttribute");
    attr_set[ki] = true;
  }
  const int grid = B * H * p.n_qtiles;
  double kv_avg = Lkv;  // dense; block-causal work is smaller (reported as dense upper bound / 1)
  ProfScope prof(PROF_ATTN, (hipStream_t)stream, 4.0 * B * H * (double)Lq * kv_avg * ATT_D);
  const dim3 g(grid), t(waves * 64);
  if (waves == 4) {
    if (f16) hipLaunchKernelGGL((attn_fwd_kernel<true, 4>), g, t, lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, 4>), g, t, lds, (hipStream_t)stream, p);
  } else {
    if (f16) hipLaunchKernelGGL((attn_fwd_kernel<true, 8>), g, t, lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, 8>), g, t, lds, (hipStream_t)stream, p);
  }
  return check_launch("attn_fwd");
}
