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
#include "attn_common.h"

namespace rtv {

// Occupancy: ~220 unified registers -> 2 waves per SIMD (the second __launch_bounds__ argument), i.e. ONE 8-wave workgroup or
// two 4-wave workgroups per CU.  A 6-wave build cannot help partial rounds: its workgroup would still occupy a CU for the
// time its two doubly-loaded SIMDs need.
#ifdef RTV_ATTN_TRACE   // lab builds only (scripts/micro/build_attn_trace.sh): s_memtime stamps of KV tiles 20..23, wave 0 / NW-1
__device__ unsigned* g_attn_trace = nullptr;   // [512 blocks][2 waves][4 tiles][6 stamps]
#define ATT_STAMP(i)                                                                    \
  do {                                                                                  \
    if (tr_on && j >= 20 && j < 24) {                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                \
      const unsigned t_ = (unsigned)__builtin_readcyclecounter();                       \
      if (lane == 0) tr_dst[(j - 20) * 6 + (i)] = t_;                                   \
      __builtin_amdgcn_sched_barrier(0);                                                \
    }                                                                                   \
  } while (0)
#else
#define ATT_STAMP(i)
#endif

// Merge of the S partial results of a row: m = max m_s, w_s = 2^(m_s - m), O = sum w_s O_s / sum w_s l_s.
// 16 threads per row (8 dims each), 16 rows per workgroup.
template <bool F16>
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                                           uint16_t* __restrict__ o, int S, int B, int H, int Lq, int64_t o_bs,
                                                           int64_t o_rs) {
  const int64_t nrows = (int64_t)B * H * Lq;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (row >= nrows) return;
  const int c8 = (threadIdx.x & 15) * 8;
  float m = -INFINITY;
  for (int s = 0; s < S; ++s) m = fmaxf(m, part_ml[((int64_t)s * nrows + row) * 2]);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float l = 0.f;
  for (int s = 0; s < S; ++s) {
    const f32x2 ml = *(const f32x2*)(part_ml + ((int64_t)s * nrows + row) * 2);
    const float w = __builtin_amdgcn_exp2f(ml[0] - m);
    l += w * ml[1];
    const float* po = part_o + ((int64_t)s * nrows + row) * ATT_D + c8;
    const f32x4 a = *(const f32x4*)po, b2 = *(const f32x4*)(po + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] += w * a[i];
      acc[4 + i] += w * b2[i];
    }
  }
  const float inv = 1.0f / l;
  const int bh = (int)(row / Lq), q = (int)(row - (int64_t)bh * Lq);
  const int b = bh / H, h = bh - b * H;
  u32x4 out;
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = pack2<F16>(acc[2 * i] * inv, acc[2 * i + 1] * inv);
  *(u32x4*)(o + (size_t)b * o_bs + (size_t)q * o_rs + (size_t)h * ATT_D + c8) = out;
}

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
  int t_lo, t_hi;
  split_tile_range(p, ntiles, &t_lo, &t_hi);

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

  // Schedules measured against this plain one (one barrier per tile, the compiler interleaving the exponentials of tile j
  // with the PV MFMAs), all equal or slower (profiles/r01_attn_variants.txt, profiles/r02_attn_*.log):
  //  * waves 4-7 half a tile behind waves 0-3: -2..3 %;
  //  * K / V^T fragments software-prefetched 4 MFMAs ahead of their use: +-0 % (the partner wave of the SIMD already fills
  //    the LDS-latency gaps);
  //  * GEMM-style ping-pong (a VALU segment = softmax + staging, an MFMA segment = PV(j) + QK^T(j+1), the two wave groups one
  //    barrier apart, fragments prefetched through rotating windows): equal - phase trace: with the partner in its VALU
  //    segment nobody fills the LDS latency of the MFMA segment (60-68 cycles per MFMA with the 2-deep windows 252 VGPRs
  //    allow), so what the role split gains the exposed latency loses.  The lockstep trace itself: tile period 4300 cycles
  //    for 2 x 1024 cycles of MFMA per SIMD; the older wave of a SIMD finishes its tile in ~2900 and waits ~1100 at the
  //    barrier for the younger one.  Going further needs the fragments of >= 8 MFMAs in flight, i.e. the 512-register
  //    one-wave-per-SIMD layout with hand-placed issue slots (MI355X guide: 1.25-1.4 PF) - not reachable with hipcc's
  //    scheduling (r01: 583 TF/s).
#ifdef RTV_ATTN_TRACE
  const bool tr_on = g_attn_trace != nullptr && (wave == 0 || wave == NW - 1) && blockIdx.x < 512;
  unsigned* const tr_dst = g_attn_trace + ((size_t)(blockIdx.x & 511) * 2 + (wave == 0 ? 0 : 1)) * 24;
#endif
  auto tile = [&](const int j, auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    const bool has_next = (j + 1 < t_hi);
    ATT_STAMP(0);
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

    ATT_STAMP(1);
    // ---------------- a key that stands for several identical ones: + log2(count) in the exponent (kernel-uniform test)
    if (p.dup_key >= 0 && (unsigned)(p.dup_key - j * ATT_KT) < (unsigned)ATT_KT) {
      const int local = p.dup_key - j * ATT_KT;
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * g == local) sacc[kbk][r] += p.dup_bias;
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

    ATT_STAMP(2);
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

    ATT_STAMP(3);
    if (has_next) write_tile(buf ^ 1);
    ATT_STAMP(4);
    __syncthreads();
    ATT_STAMP(5);
  };

  load_tile(t_lo);
  write_tile(0);
  __syncthreads();
  for (int j = t_lo; j < t_hi; j += 2) {
    tile(j, IntC<0>{});
    if (j + 1 < t_hi) tile(j + 1, IntC<1>{});
  }

  // ---------------- epilogue: O = O^T / l, lane owns row q and dims db*32 + 8*i + 4*g + {0..3}
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (p.kv_splits > 1) {
    store_partial(p, bh, q_row, g, oacc, m_run, l_tot);
    return;
  }
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


// =====================================================================================================================
// Four-phase ping-pong schedule (8 waves, 256 query rows): the kernel the 256-row launches use.
//
// Phase trace of the lockstep loop above (profiles/r02_attn_phase_trace_v1.log): a tile period of 4300 cycles for the
// 2 x 1024 MFMA cycles a SIMD owes its two waves - both waves of a SIMD are in the same phase at the same time, and every
// MFMA waits for its own ds_read.  The earlier two-segment ping-pong (profiles/r02_attn_pingpong_experiment.log) moved the
// waves apart but kept the reads inside the MFMA segment (60-68 cycles per MFMA) and paid 750-950 cycles per tile for the
// register-staged global -> LDS copies.  Here:
//   * K / V tiles go HBM/L2 -> LDS by DMA (`buffer_load ... lds`, 16 B per lane, source-side swizzle so the LDS image is
//     the same XOR layout the fragment reads want), two tiles ahead in a 3-slot ring per operand, retired by ONE counted
//     `s_waitcnt vmcnt(4)` per issue point a whole tile before the data is read - no staging registers, no ds_write;
//   * a tile is four phases per wave, separated by s_barrier:
//         LK  read the 16 K fragments of the tile into registers (16 ds_read_b128), issue the K DMA of tile j+2
//         QK  16 MFMAs on register operands, S^T = K . Q^T; behind MFMA n the V^T fragment n of the tile is read (2 transpose
//             reads) into the register MFMA n has just consumed - K and V^T fragments share one 64-register block, and the
//             reads cost issue slots in the shadow of the matrix pipe instead of ~280 cycles in front of the softmax
//         LV  issue the V DMA of tile j+2; mask, row maximum, lazy rescale, first quarter of the exponentials -> P^T (VALU beside
//             the partner's MFMAs)
//         PV  16 MFMAs on register operands, O^T += V^T . P^T, the other three quarters of the exponentials behind the first twelve
//     waves 4-7 run ONE phase behind waves 0-3, so in every phase one wave of each SIMD is in a matrix phase and its
//     partner in a load/VALU phase:
//         phase 4j: g0 LK(j)  g1 PV(j-1) | 4j+1: g0 QK(j)  g1 LK(j) | 4j+2: g0 LV(j)  g1 QK(j) | 4j+3: g0 PV(j)  g1 LV(j)
//   * every fragment read is inline asm: the compiler's wait-count pass puts vmcnt(0) in front of an LDS read it cannot
//     disambiguate from a pending LDS DMA, and its lgkmcnt bookkeeping of the K reads would wait on the younger V^T reads;
//     the consumers wait with one explicit lgkmcnt(0) the fragment registers depend on (lds_wait_frags)
//   * ring hazards: K(j+2) lands in the slot of K(j-1), last read in phase 4j-3; V(j+2) (issued in phase 4j+2 or later) in the
//     slot of V(j-1), last read in phase 4j-2; a tile's DMA is complete (issuer's vmcnt + the phase barrier) >= 2 phases
//     before its first read.
// Registers: Q^T 32 + O^T 64 + S^T 32 + P^T 16 + fragments 64 = 208 of the 256 a wave has at two waves per SIMD.
constexpr int ATT_NB = 3;   // ring slots per operand

#ifdef RTV_ATTN_TRACE
#define ATT_STAMP8(i)                                                                   \
  do {                                                                                  \
    if (tr_on && j >= 20 && j < 24) {                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                \
      const unsigned t_ = (unsigned)__builtin_readcyclecounter();                       \
      if (lane == 0) tr_lds[(j - 20) * 8 + (i)] = t_;   /* LDS, not global: stores would count in vmcnt */ \
      __builtin_amdgcn_sched_barrier(0);                                                \
    }                                                                                   \
  } while (0)
#else
#define ATT_STAMP8(i)
#endif

template <bool F16>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(AttnParams p) {
  constexpr int ATT_QT = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // smem: K[ATT_NB][16 KiB] | V[ATT_NB][16 KiB]
  char* const sK = smem;
  char* const sV = smem + ATT_NB * ATT_TILE_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;   // 0: waves 0-3, 1: waves 4-7 (one phase behind)
  const int l31 = lane & 31, g = lane >> 5;

  int bh, qt;
  {
    const int nbh = p.B * p.H;
    const int bid = blockIdx.x;
    if (nbh % 8 == 0) {   // all q tiles of a head on ONE XCD (block b runs on XCD b % 8): its K/V stream stays in that L2
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

  // ---- key-prefix limits (block-causal rule kv < ends[q], causal_model.py:134-136)
  const int q_row = q0 + wave * ATT_QW + l31;
  const int q_row_c = min(q_row, p.Lq - 1);
  int kv_lim = p.Lkv, wave_min_lim = p.Lkv, wg_max_lim = p.Lkv;
  if (p.causal_block > 0) {
    const int cb = p.causal_block;
    kv_lim = min(p.Lkv, ((p.q_offset + q_row_c) / cb + 1) * cb);
    int first = min(q0 + wave * ATT_QW, p.Lq - 1);
    wave_min_lim = min(p.Lkv, ((p.q_offset + first) / cb + 1) * cb);
    int last = min(q0 + ATT_QT - 1, p.Lq - 1);
    wg_max_lim = min(p.Lkv, ((p.q_offset + last) / cb + 1) * cb);
  }
  const int ntiles = (wg_max_lim + ATT_KT - 1) / ATT_KT;
  int t_lo, t_hi;
  split_tile_range(p, ntiles, &t_lo, &t_hi);

  // ---- DMA staging: wave w moves tile rows 8w .. 8w+7 of K and of V, two 1-KiB pieces (4 rows) each.  Lane l lands at byte
  //      16 l of the piece = row (l >> 4), chunk position (l & 15); it fetches the chunk that belongs there:
  //      K: position = chunk ^ (row & 15);  V: position = chunk ^ ((row & 3) << 2)   (the layouts of the fragment reads).
  //      Full tiles: per-lane byte offset fixed for the whole kernel + a uniform tile offset (SGPR); the ragged last tile and
  //      the tile that straddles the two ranges of a ring window compute a per-lane row.
  const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc((void*)kb, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcV = __builtin_amdgcn_make_buffer_rsrc((void*)vb, 0, 0x7fffffff, 0x00020000);
  const int st_row = wave * 8 + (lane >> 4);   // tile row of piece 0 (piece 1: + 4)
  const int st_cp = lane & 15;
  const int k_rs = (int)p.k_rs, v_rs = (int)p.v_rs;
  // window constants pinned in SGPRs: left to itself the compiler turns `in0 ? p.off0 : p.delta` into an s_load from the kernel
  // argument segment at a selected address plus `s_waitcnt lgkmcnt(0)` in front of every DMA issue (a scalar-memory round trip
  // and a drain of the LDS reads in flight, twice per tile)
  const int w_off0 = __builtin_amdgcn_readfirstlane(p.off0), w_delta = __builtin_amdgcn_readfirstlane(p.delta);
  const int w_n0 = __builtin_amdgcn_readfirstlane(p.n0), w_lkv = __builtin_amdgcn_readfirstlane(p.Lkv);
  int k_ch[2], v_ch[2];
  uint32_t k_fast[2], v_fast[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = st_row + 4 * i;
    k_ch[i] = (st_cp ^ (r & 15)) * 8;
    v_ch[i] = (st_cp ^ ((r & 3) << 2)) * 8;
    k_fast[i] = (uint32_t)(r * k_rs + k_ch[i]) * 2u;
    v_fast[i] = (uint32_t)(r * v_rs + v_ch[i]) * 2u;
  }
  auto stage = [&](const __amdgpu_buffer_rsrc_t& rsrc, char* ring, int j, int rs, const uint32_t (&fast)[2],
                   const int (&ch)[2]) {
    const int row0 = j * ATT_KT;
    char* const dst = ring + (j % ATT_NB) * ATT_TILE_BYTES + wave * (8 * 256);
    const bool in0 = row0 + ATT_KT <= w_n0;
    if (in0 || (row0 >= w_n0 && row0 + ATT_KT <= w_lkv)) {
      const uint32_t so = (uint32_t)((row0 + (in0 ? w_off0 : w_delta)) * rs) * 2u;
#pragma unroll
      for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (RTV_LDS void*)(dst + i * 1024), 16, fast[i], so, 0, 0);
    } else {   // (also every tile past the end: clamped rows into a slot nobody reads - the wait counts stay uniform)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int kv = min(row0 + st_row + 4 * i, w_lkv - 1);
        kv += kv >= w_n0 ? w_delta : w_off0;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (RTV_LDS void*)(dst + i * 1024), 16,
                                                 (uint32_t)(kv * rs + ch[i]) * 2u, 0, 0, 0);
      }
    }
  };
  auto stage_k = [&](int j) { stage(rsrcK, sK, j, k_rs, k_fast, k_ch); };
  auto stage_v = [&](int j) { stage(rsrcV, sV, j, v_rs, v_fast, v_ch); };

  // ---- prologue: tiles 0 and 1 of both operands, Q^T fragments (MFMA B operand): lane holds Q[q][dc*16 + g*8 .. +8]
  stage_k(t_lo);
  stage_v(t_lo);
  stage_k(t_lo + 1);
  stage_v(t_lo + 1);
  u32x4 qf[8];
  {
    const uint16_t* qp = qb + (size_t)q_row_c * p.q_rs + g * 8;
#pragma unroll
    for (int dc = 0; dc < 8; ++dc) qf[dc] = *(const u32x4*)(qp + dc * 16);
  }

  // ---- per-lane LDS read addresses (slot 0); K operand (A): row = kbk*32 + l31, chunk = (dc*2 + g) ^ (row & 15)
  uint32_t k_rd32[8];   // 32-bit LDS addresses: every fragment read of this kernel is inline asm (see lds_tr_read_at)
#pragma unroll
  for (int dc = 0; dc < 8; ++dc)
    k_rd32[dc] = (uint32_t)(uintptr_t)(RTV_LDS const char*)(sK + l31 * 256 + (((dc * 2 + g) ^ (l31 & 15)) << 4));
  // V^T operand (A) via transpose read: 16-lane group gathers a [4 keys][16 dims] block
  const int i16 = lane & 15, h16 = (lane >> 4) & 1;
  uint32_t v_rd32[4];   // 32-bit LDS addresses (the transpose reads are inline asm)
#pragma unroll
  for (int db = 0; db < 4; ++db)
    v_rd32[db] = (uint32_t)(uintptr_t)(RTV_LDS const char*)(sV + (4 * g + (i16 >> 2)) * 256 + h16 * 32 + (i16 & 3) * 8 +
                                                            ((db ^ (i16 >> 2)) << 6));

  f32x16 oacc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[db][r] = 0.f;
  float m_run = -1e30f;  // reference point of the exponentials (>= running max - RESCALE_SLACK), log2 domain
  float l_run = 0.f;     // this lane's partial row sum
  const float c = p.scale_log2e;
  const f32x2 c2 = {c, c};
  constexpr float RESCALE_SLACK = 8.f;   // lazy rescaling, see the lockstep kernel

  f32x16 sacc[2];   // S^T of the tile in flight (two 32-key blocks)
  u32x4 pf[2][2];   // P^T: [kv block][k-step], 8 x 16-bit = the lane's own 8 keys of that 16-key step
  u32x4 frag[16];   // K fragments [dc][kbk] during LK/QK, V^T fragments [kbk][s][db] during LV/PV
  f32x2 ps2 = {0.f, 0.f};   // row-sum partials of the tile in flight
  // one step of the exponentials: keys 2t, 2t+1 of 32-key block kbk -> half a P^T register
  auto exp_step = [&](auto kc, auto tc) {
    constexpr int kbk = decltype(kc)::value, t = decltype(tc)::value;
    const f32x2 nm2 = {-m_run, -m_run};
    f32x2 x = {sacc[kbk][2 * t], sacc[kbk][2 * t + 1]};
    x = __builtin_elementwise_fma(x, c2, nm2);  // v_pk_fma_f32
#ifdef ATT_LAB_T3   // timing experiment: no exponentials
    f32x2 e = x;
#else
    f32x2 e = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
#endif
    ps2 += e;                                   // v_pk_add_f32
    pf[kbk][t >> 2][t & 3] = pack2<F16>(e[0], e[1]);
  };

#ifdef RTV_ATTN_TRACE
  const bool tr_on = g_attn_trace != nullptr && (wave == 0 || wave == 7) && blockIdx.x < 512;
  unsigned* const tr_dst = g_attn_trace + ((size_t)(blockIdx.x & 511) * 2 + (wave == 0 ? 0 : 1)) * 32;
  unsigned* const tr_lds = (unsigned*)(smem + 2 * ATT_NB * ATT_TILE_BYTES) + (wave == 0 ? 0 : 32);   // 256 B behind the rings
#endif
#define PP_BARRIER()                        \
  do {                                      \
    __builtin_amdgcn_sched_barrier(0);      \
    __builtin_amdgcn_s_barrier();           \
    __builtin_amdgcn_sched_barrier(0);      \
  } while (0)

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (grp) PP_BARRIER();   // the stagger

  // A wave whose 32 query rows all lie beyond Lq (Lq = 4680: waves 3-7 of the 19th query tile of every head) has nothing to
  // compute: it keeps its four barriers per tile and its K / V DMA duty (the tiles are shared) and skips the fragment
  // reads, both matrix phases and the softmax - a separate loop, the working waves' loop carries no test for it.
  const bool idle_rows = p.skip_idle && q0 + wave * ATT_QW >= p.Lq;   // wave-uniform
  if (idle_rows) {
    for (int j = t_lo; j < t_hi; ++j) {
      stage_k(j + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      PP_BARRIER();
      PP_BARRIER();
      stage_v(j + 2);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      PP_BARRIER();
      PP_BARRIER();
    }
    if (!grp) PP_BARRIER();
    return;
  }
  for (int j = t_lo; j < t_hi; ++j) {
    const int slot_off = (j % ATT_NB) * ATT_TILE_BYTES;
    // ---------------- LK
    ATT_STAMP8(0);
#ifndef ATT_LAB_T6
    stage_k(j + 2);
#endif
    static_for<0, 16>([&](auto ic) {   // fragment n = [dc = n >> 1][kbk = n & 1]
      constexpr int n = decltype(ic)::value;
      frag[n] = lds_read128_at<(n & 1) * 32 * 256>(k_rd32[n >> 1] + (uint32_t)slot_off);
    });
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // everything but the two youngest issue points: K(j+1) has landed
    ATT_STAMP8(1);
    PP_BARRIER();
    // ---------------- QK: S^T = K . Q^T (two independent accumulator chains).  MFMA n consumes fragment register n; the V^T
    // fragment n of the same tile is read into that register right behind it (a transpose read costs an issue slot in the
    // shadow of the matrix pipe here, and ~9 cycles of LDS queueing when all four waves of a group issue their 32 at once in
    // the LV phase, where the softmax then waits behind them).
    ATT_STAMP8(2);
    lds_wait_frags(frag);
    uint32_t va[4];
#pragma unroll
    for (int db = 0; db < 4; ++db) va[db] = v_rd32[db] + (uint32_t)slot_off;
    auto rd_v = [&](auto ic) {   // V^T fragment idx = (kbk*2 + s)*4 + db
#ifndef ATT_LAB_T4   // (T4, timing experiment: no V reads - PV runs on the K fragments)
      constexpr int idx = decltype(ic)::value;
      constexpr int off = ((idx >> 3) * 32 + ((idx >> 2) & 1) * 16) * 256;
      const u32x2 lo = lds_tr_read_at<off>(va[idx & 3]);             // keys +0..3  (this lane group's first quad)
      const u32x2 hi = lds_tr_read_at<off + 8 * 256>(va[idx & 3]);   // keys +8..11
      frag[idx] = u32x4{lo[0], lo[1], hi[0], hi[1]};
#endif
    };
    static_for<0, 8>([&](auto ic) {
      constexpr int dc = decltype(ic)::value;
      if constexpr (dc == 0) {
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        sacc[0] = mfma32<F16>(frag[0], qf[0], z);
        sacc[1] = mfma32<F16>(frag[1], qf[0], z);
      } else {
        sacc[0] = mfma32<F16>(frag[2 * dc], qf[dc], sacc[0]);
        sacc[1] = mfma32<F16>(frag[2 * dc + 1], qf[dc], sacc[1]);
      }
      __builtin_amdgcn_sched_barrier(0);
      rd_v(IntC<2 * dc>{});
      rd_v(IntC<2 * dc + 1>{});
      __builtin_amdgcn_sched_barrier(0);
    });
    ATT_STAMP8(3);
    PP_BARRIER();
    // ---------------- LV: V DMA, mask + online softmax (first 32-key block of the exponentials)
    ATT_STAMP8(4);
#ifndef ATT_LAB_T6
    stage_v(j + 2);
#endif
    if ((j + 1) * ATT_KT > wave_min_lim) {   // mask: only on tiles that cross a limit of this wave
#pragma unroll
      for (int kbk = 0; kbk < 2; ++kbk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int kv = j * ATT_KT + kbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (kv >= kv_lim) sacc[kbk][r] = -INFINITY;
        }
    }
    {
      float mx = sacc[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sacc[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[1][r]);
      mx = xor32_max(mx);
      const float m_cand = fmaxf(m_run, mx * c);
      const bool need = m_cand - m_run > RESCALE_SLACK;
      if (__builtin_amdgcn_ballot_w64(need) != 0) {
        // wave-uniform branch, per-row decision: rows that do not need it multiply by exp2(0) = 1 exactly, so a row's
        // arithmetic never depends on which other rows share its wave (token-sharded == unsharded, bit for bit)
        const float m_new = need ? m_cand : m_run;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[db][r] *= alpha;
      }
      // exponentials of the first 16 keys here (what the first four PV MFMAs multiply); the others run inside the PV phase, one
      // step behind each of its first twelve MFMAs (the wave's own issue slots while the matrix pipe works)
      static_for<0, 4>([&](auto ic) { exp_step(IntC<0>{}, ic); });
    }
    // the first QUARTER of P^T (keys 0-15: what the first four PV MFMAs multiply) is complete HERE: without this the compiler
    // sinks exponentials / packs below the barrier, into the matrix phase (sched_barrier does not stop its code sinking)
    asm volatile("" : "+v"(pf[0][0]), "+v"(ps2));
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // V(j+1) has landed
    ATT_STAMP8(5);
    PP_BARRIER();
    // ---------------- PV: O^T += V^T . P^T (four independent accumulator chains)
    ATT_STAMP8(6);
    lds_wait_frags(frag);
    // 16 MFMAs in the order (key block, 16-key step, dim block); behind MFMA i (i < 12) one step of the exponentials that are
    // still missing - keys 16-31 of the first block behind MFMAs 0-3 (needed from MFMA 4 on), the second block behind MFMAs 4-11
    // (needed from MFMA 8 / 12 on): the wave's own issue slots while the matrix pipe works
    static_for<0, 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value, kbk = i >> 3, s_ = (i >> 2) & 1, db = i & 3;
      oacc[db] = mfma32<F16>(frag[(kbk * 2 + s_) * 4 + db], pf[kbk][s_], oacc[db]);
      if constexpr (i < 12) {
        constexpr int f = i + 4;
        exp_step(IntC<(f >> 3)>{}, IntC<(f & 7)>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    l_run += ps2[0] + ps2[1];
    ps2 = f32x2{0.f, 0.f};
    asm volatile("" : "+v"(l_run));
    ATT_STAMP8(7);
    PP_BARRIER();
  }
  if (!grp) PP_BARRIER();   // waves 0-3 close the stagger (equal barrier counts)
#undef PP_BARRIER

#ifdef RTV_ATTN_TRACE
  if (tr_on && lane < 32 && ntiles >= 24) tr_dst[lane] = tr_lds[lane];
#endif
  // ---------------- epilogue: O = O^T / l, lane owns row q and dims db*32 + 8*i + 4*g + {0..3}
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (p.kv_splits > 1) {
    store_partial(p, bh, q_row, g, oacc, m_run, l_tot);
    return;
  }
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

#ifdef RTV_ATTN_TRACE
extern "C" int rtv_attn_debug_trace(unsigned* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(rtv::g_attn_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

static std::atomic<int> g_attn_waves{0};     // 0 = by grid size (include/rtv_hip_lab.h)
static std::atomic<bool> g_attn_lockstep{false};  // 256-row launches on the lockstep kernel only (A/B runs, tests)
static std::atomic<bool> g_attn_force_pp{false};  // ... on the four-phase kernel whatever the window length
static std::atomic<bool> g_attn_skip_idle{true};  // rtv_attn_set_skip_idle(0): A/B of the idle-wave loop
// 256-row launches on the one-wave-per-SIMD kernel (attn_w4.hip): >= 0 a forced kernel variant, -2 = where it applies, its
// default variant (W4_DEFAULT), -1 = never (rtv_attn_set_waves(81 / 82) pin the older schedules)
static std::atomic<int> g_attn_w4{-2};
constexpr int W4_DEFAULT = 600;   // plain row sums (bit-identical with the four-phase kernel), one M0 write per operand, output rows through LDS
namespace rtv {
int launch_attn_w4(const AttnParams& p, bool f16, int variant, dim3 grid, hipStream_t stream);   // attn_w4.hip
}

extern "C" int rtv_attn_set_skip_idle(int on) {
  g_attn_skip_idle = on != 0;
  return 0;
}

extern "C" int rtv_attn_set_waves(int waves) {
  if (waves != 0 && waves != 4 && waves != 8 && waves != 81 && waves != 82 && !(waves >= 840 && waves <= 1639))
    return set_error(-1, "attn_set_waves: 0 (auto), 4, 8, 81 (256 rows, lockstep schedule), 82 (256 rows, four-phase schedule) or "
                         "840 + v (256 rows, one wave per SIMD, kernel variant v: attn_w4.hip)");
  g_attn_lockstep = waves == 81;
  g_attn_force_pp = waves == 82;
  g_attn_w4 = waves >= 840 ? waves - 840 : (waves == 81 || waves == 82 ? -1 : -2);
  g_attn_waves = waves > 8 ? 8 : waves;
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

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* o, int B, int Lq, int Lkv0, int Lkv1,
                         int seg1_row, int H, int D, int64_t q_batch_stride, int64_t q_row_stride,
                         int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride,
                         int64_t v_row_stride, int64_t o_batch_stride, int64_t o_row_stride, float scale,
                         int causal_block, int q_offset, int dtype, rtv_stream_t stream, int dup_key, int dup_count,
                         int kv_splits = 1, void* split_ws = nullptr, size_t split_ws_bytes = 0);

extern "C" int rtv_attn_fwd_win(const void* q, const void* k, const void* v, void* o, int B, int Lq, int Lkv0, int Lkv1,
                                int seg1_row, int H, int D, int64_t q_batch_stride, int64_t q_row_stride,
                                int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride,
                                int64_t v_row_stride, int64_t o_batch_stride, int64_t o_row_stride, float scale,
                                int causal_block, int q_offset, int dtype, rtv_stream_t stream) {
  return attn_fwd_impl(q, k, v, o, B, Lq, Lkv0, Lkv1, seg1_row, H, D, q_batch_stride, q_row_stride, k_batch_stride, k_row_stride,
                       v_batch_stride, v_row_stride, o_batch_stride, o_row_stride, scale, causal_block, q_offset, dtype, stream,
                       -1, 0);
}

extern "C" int rtv_attn_fwd_dup(const void* q, const void* k, const void* v, void* o, int B, int Lq, int Lkv, int H, int D,
                                int64_t q_batch_stride, int64_t q_row_stride, int64_t k_batch_stride, int64_t k_row_stride,
                                int64_t v_batch_stride, int64_t v_row_stride, int64_t o_batch_stride, int64_t o_row_stride,
                                float scale, int dup_key, int dup_count, int dtype, rtv_stream_t stream) {
  if (dup_key < 0 || dup_key >= Lkv || dup_count < 1) return set_error(-1, "attn_fwd_dup: dup_key must be a key of the window, dup_count >= 1");
  return attn_fwd_impl(q, k, v, o, B, Lq, Lkv, 0, 0, H, D, q_batch_stride, q_row_stride, k_batch_stride, k_row_stride,
                       v_batch_stride, v_row_stride, o_batch_stride, o_row_stride, scale, 0, 0, dtype, stream,
                       dup_count > 1 ? dup_key : -1, dup_count);
}

extern "C" size_t rtv_attn_split_workspace_bytes(int B, int Lq, int H, int kv_splits) {
  if (B <= 0 || Lq <= 0 || H <= 0 || kv_splits < 2) return 0;
  return (size_t)kv_splits * B * H * Lq * (ATT_D + 2) * sizeof(float);
}

extern "C" int rtv_attn_fwd_split(const void* q, const void* k, const void* v, void* o, int B, int Lq, int Lkv0, int Lkv1,
                                  int seg1_row, int H, int D, int64_t q_batch_stride, int64_t q_row_stride,
                                  int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride,
                                  int64_t v_row_stride, int64_t o_batch_stride, int64_t o_row_stride, float scale,
                                  int causal_block, int q_offset, int kv_splits, void* workspace, size_t workspace_bytes,
                                  int dtype, rtv_stream_t stream) {
  if (kv_splits < 1 || kv_splits > 16) return set_error(-1, "attn_fwd_split: kv_splits must be 1..16");
  return attn_fwd_impl(q, k, v, o, B, Lq, Lkv0, Lkv1, seg1_row, H, D, q_batch_stride, q_row_stride, k_batch_stride, k_row_stride,
                       v_batch_stride, v_row_stride, o_batch_stride, o_row_stride, scale, causal_block, q_offset, dtype, stream, -1,
                       0, kv_splits, workspace, workspace_bytes);
}

static int attn_fwd_impl(const void* q, const void* k, const void* v, void* o, int B, int Lq, int Lkv0, int Lkv1,
                         int seg1_row, int H, int D, int64_t q_batch_stride, int64_t q_row_stride,
                         int64_t k_batch_stride, int64_t k_row_stride, int64_t v_batch_stride,
                         int64_t v_row_stride, int64_t o_batch_stride, int64_t o_row_stride, float scale,
                         int causal_block, int q_offset, int dtype, rtv_stream_t stream, int dup_key, int dup_count,
                         int kv_splits, void* split_ws, size_t split_ws_bytes) {
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
  p.dup_key = dup_key;
  p.dup_bias = dup_key >= 0 ? log2f((float)dup_count) / p.scale_log2e : 0.f;
  // never more splits than key tiles; one split = the plain launch
  if (kv_splits > (Lkv + ATT_KT - 1) / ATT_KT) kv_splits = (Lkv + ATT_KT - 1) / ATT_KT;
  if (kv_splits < 1) kv_splits = 1;
  p.kv_splits = kv_splits;
  p.part_o = nullptr;
  p.part_ml = nullptr;
  if (kv_splits > 1) {
    if (dup_key >= 0) return set_error(-1, "attn_fwd: no KV split of a window with a counted key");
    const size_t rows = (size_t)kv_splits * B * H * Lq;
    if (!split_ws || ((uintptr_t)split_ws & 15) || split_ws_bytes < rows * (ATT_D + 2) * sizeof(float))
      return set_error(-1, "attn_fwd: split workspace missing, misaligned or too small (rtv_attn_split_workspace_bytes)");
    p.part_o = (float*)split_ws;
    p.part_ml = p.part_o + rows * ATT_D;
  }
  p.causal_block = causal_block;
  p.q_offset = q_offset;
  p.n0 = Lkv1 > 0 ? Lkv0 : Lkv;
  p.delta = Lkv1 > 0 ? seg1_row - Lkv0 : 0;
  // 256-row workgroups unless their grid leaves most CUs idle: then 128-row ones.  Measured (MI355X, Lkv 14040, ms for 8 / 4
  // waves): 4680 rows x 20 heads 0.90 / 0.88, x 10 heads (190 workgroups) 0.46 / 0.51, x 5 heads (95) 0.40 / 0.32,
  // 585 rows x 40 heads (120) 0.40 / 0.34.
  int waves = g_attn_waves;
  // 128-row / 4-wave workgroups (two per CU) when the 256-row grid cannot fill the chip, and for short key windows (the
  // text cross-attention, 512 keys = 8 tiles): a workgroup is then mostly prologue and epilogue, which two co-resident
  // workgroups overlap (scripts/cross_attn_ab.py: 74.9 vs 79.0 us at 4680 x 512 x 40 heads; bit-identical)
  // "cannot fill the chip": fewer 256-row workgroups than 5/8 of the device's CUs (160 on the 256-CU MI355X) - the same CU count
  // parallel.attn_kv_splits_for plans its key ranges with (rtv_internal.h: every round rule uses device_num_cus())
  const int cus = device_num_cus() > 0 ? device_num_cus() : 256;
  if (waves == 0) waves = ((int64_t)B * H * ((Lq + 255) / 256) * kv_splits < (int64_t)cus * 5 / 8 || Lkv <= 512) ? 4 : 8;
  const int qt_rows = ATT_QW * waves;
  p.n_qtiles = (Lq + qt_rows - 1) / qt_rows;
  const int lds = 4 * ATT_TILE_BYTES;
  const bool f16 = dtype == RTV_DTYPE_F16;
  const void* kerns[4] = {(const void*)attn_fwd_kernel<false, 8>, (const void*)attn_fwd_kernel<true, 8>,
                          (const void*)attn_fwd_kernel<false, 4>, (const void*)attn_fwd_kernel<true, 4>};
  static LdsAttr lds_attr[4];   // per kernel, per device
  const int ki = (waves == 4 ? 2 : 0) + (f16 ? 1 : 0);
  if (int st = ensure_dynamic_lds(kerns[ki], lds, &lds_attr[ki], "attn_fwd")) return st;
  const int grid = B * H * p.n_qtiles;
  // keys a query row really attends, averaged over the rows: Lkv for a dense launch; under the block-causal mask row r sees the
  // first ((r + q_offset) / causal_block + 1) * causal_block keys (r06: the mask used to be counted as dense, which overstated
  // the attention rate of recompute forwards over more than one block - kv_cache_num_frames 9 / 18 - in bench.py's line)
  double kv_avg = Lkv;
  if (causal_block > 0 && Lq > 0) {
    double keys = 0;
    for (int64_t r0 = 0; r0 < Lq;) {
      const int64_t blk = (r0 + q_offset) / causal_block;
      int64_t r1 = (blk + 1) * (int64_t)causal_block - q_offset;
      if (r1 > Lq) r1 = Lq;
      const int64_t lim = (blk + 1) * (int64_t)causal_block;
      keys += (double)(r1 - r0) * (double)(lim < Lkv ? lim : Lkv);
      r0 = r1;
    }
    kv_avg = keys / Lq;
  }
  ProfScope prof(PROF_ATTN, (hipStream_t)stream, 4.0 * B * H * (double)Lq * kv_avg * ATT_D);
  const dim3 g(grid, kv_splits), t(waves * 64);
  auto combine = [&]() -> int {
    if (int st = check_launch("attn_fwd")) return st;
    const int64_t nrows = (int64_t)B * H * Lq;
    const dim3 cg((unsigned)((nrows + 15) / 16)), ct(256);
    note_kernel(DK_ATTN_SPLIT_COMBINE);
    if (f16) hipLaunchKernelGGL((attn_combine_kernel<true>), cg, ct, 0, (hipStream_t)stream, p.part_o, p.part_ml, p.o, kv_splits, B, H, Lq, p.o_bs, p.o_rs);
    else hipLaunchKernelGGL((attn_combine_kernel<false>), cg, ct, 0, (hipStream_t)stream, p.part_o, p.part_ml, p.o, kv_splits, B, H, Lq, p.o_bs, p.o_rs);
    return check_launch("attn_combine");
  };
  // Four-phase kernel: its DMA addresses K / V rows with non-negative 32-bit byte offsets from a buffer base, so the base is
  // the lowest row of the window (the second range of a ring window lies BELOW the first one).
  p.off0 = 0;
  p.skip_idle = g_attn_skip_idle ? 1 : 0;
  const int base_shift = (Lkv1 > 0 && seg1_row < 0) ? seg1_row : 0;
  const int64_t top0 = (int64_t)p.n0 - base_shift, top1 = (int64_t)Lkv + p.delta - base_shift;
  const int64_t rs_max = k_row_stride > v_row_stride ? k_row_stride : v_row_stride;
  const bool offsets_fit = k_row_stride > 0 && v_row_stride > 0 &&
                           ((top0 > top1 ? top0 : top1) + ATT_KT) * rs_max * 2 < 0x7fffffffLL;
  // Short key windows (the 512-key cross-attention) stay on the lockstep kernel: the four-phase one stages two tiles before its
  // first MFMA and pays four barriers per tile (measured 93 vs 84 us at 4680 x 512 x 40; +1..4 % from 4680 keys on).
  // One-wave-per-SIMD kernel (attn_w4.hip; r05): the default for 256-row bf16 launches over one row range of >= 1024 keys (a ring
  // window's two ranges, f16 and counted keys stay on the four-phase / lockstep kernels): 748 vs 790-815 us on 4680 x 9360 x 40,
  // 390 vs 410 on the block-causal recompute shape, bit-identical (profiles/r05_attn_w4_*.log).
  int w4v = g_attn_w4;
  if (w4v == -2) w4v = Lkv >= 1024 ? W4_DEFAULT : -1;
  if (waves == 8 && w4v >= 0 && !f16 && Lkv1 == 0 && dup_key < 0 && offsets_fit) {
    if (int st = launch_attn_w4(p, f16, w4v, g, (hipStream_t)stream)) return st;
    note_kernel(DK_ATTN_W4);
    if (kv_splits > 1) return combine();
    return check_launch("attn_w4");
  }
  if (waves == 8 && !g_attn_lockstep && offsets_fit && dup_key < 0 && (Lkv >= 1024 || g_attn_force_pp)) {
    p.k += (int64_t)base_shift * k_row_stride;
    p.v += (int64_t)base_shift * v_row_stride;
    p.off0 = -base_shift;
    p.delta -= base_shift;
#ifdef RTV_ATTN_TRACE
    const int lds_pp = 2 * ATT_NB * ATT_TILE_BYTES + 256;   // + the stamp area of the lab build
#else
    const int lds_pp = 2 * ATT_NB * ATT_TILE_BYTES;
#endif
    static LdsAttr pp_attr[2];
    const void* kp = f16 ? (const void*)attn_fwd_pp_kernel<true> : (const void*)attn_fwd_pp_kernel<false>;
    if (int st = ensure_dynamic_lds(kp, lds_pp, &pp_attr[f16], "attn_fwd")) return st;
    note_kernel(DK_ATTN_FOUR_PHASE);
    if (f16) hipLaunchKernelGGL((attn_fwd_pp_kernel<true>), g, t, lds_pp, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_fwd_pp_kernel<false>), g, t, lds_pp, (hipStream_t)stream, p);
    if (kv_splits > 1) return combine();
    return check_launch("attn_fwd");
  }
  note_kernel(waves == 4 ? DK_ATTN_LOCKSTEP_128ROW : DK_ATTN_LOCKSTEP_256ROW);
  if (waves == 4) {
    if (f16) hipLaunchKernelGGL((attn_fwd_kernel<true, 4>), g, t, lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, 4>), g, t, lds, (hipStream_t)stream, p);
  } else {
    if (f16) hipLaunchKernelGGL((attn_fwd_kernel<true, 8>), g, t, lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<false, 8>), g, t, lds, (hipStream_t)stream, p);
  }
  if (kv_splits > 1) return combine();
  return check_launch("attn_fwd");
}
