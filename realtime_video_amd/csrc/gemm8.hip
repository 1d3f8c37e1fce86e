// 256x256x64 "ping-pong" projection GEMM for gfx950 (tile configs 4 / 5): the high-throughput variant of gemm.hip for
// the large DiT linears (same math, same fused epilogue, same C ABI).
//
// One workgroup = 8 waves on one CU, two waves per SIMD.  Waves 0-3 ("group 0", output rows 0-127) and waves 4-7
// ("group 1", rows 128-255) run the SAME phase program staggered by one s_barrier, so on every SIMD one wave is in its
// MFMA segment while its partner is in its LDS segment.  A K-tile (64 deep) is TWO phases of 16 MFMA 32x32x16 each:
//
//   interval 4t+0:  g0  LDS  X(t): W[n0], W[n1], A[m0]  (16 ds_read_b128)        g1  MFMA Y(t-1)
//   interval 4t+1:  g0  MFMA X(t): (m0; n0, n1) + stage A0, A1 of tile t+2       g1  LDS  X(t)
//   interval 4t+2:  g0  LDS  Y(t): A[m1] (8 reads), s_waitcnt vmcnt(4)           g1  MFMA X(t)
//   interval 4t+3:  g0  MFMA Y(t): (m1; n0, n1) + stage W0, W1 of tile t+2       g1  LDS  Y(t)
//
// Round-2 measurements that shaped it (scripts/gemm_lab.py, profiles/r02_gemm_*): with 8 MFMAs per segment (the round-1
// schedule) a barrier interval cost ~365 cycles for 256 cycles of MFMA issue (K-tile 3000-3300 cycles for 2048 of MFMA); with
// 16 per segment the fixed cost per interval (barrier, role switch, DMA issue) is paid half as often: 2470 cycles per
// K-tile.  And the direct register-layout epilogue (8-byte stores at a row stride) took a third of the kernel: the epilogue
// now goes through LDS (store_tile_lds, gemm_core.h).  Together 986 -> 1335 TF/s on 4096x16384x5120 (hipBLASLt: 1538).
//
// LDS (160 KiB): A has THREE K-tile buffers, W two, each {rows 0-127, rows 128-255} x 16 KiB.  Half-tiles are staged with
// `buffer_load_dwordx4 ... lds` (lane offset in a VGPR that never changes, K offset in an SGPR: no per-piece VALU), 4 pieces
// per MFMA segment, issued between MFMAs.  A(t+2) goes to A buffer (t+2)%3 - last read in Y(t-1) - and W(t+2) to W buffer
// t&1 - last read in X(t) - so every piece is in flight for >= 3 barrier intervals (~1700 cycles) before the ONE counted
// wait per K-tile (vmcnt(4) in Y's LDS segment: everything but the 4 youngest pieces) retires it; it is read one phase
// later (after a barrier every wave has passed).  The LDS image is lane-linear per DMA piece; the bank swizzle is applied
// to the per-lane source chunk and again on the ds_read_b128 (same involution).
//
// Tail-wave quantisation (e.g. 380 tiles on 256 CUs = 1.48 rounds), three forms chosen by plan_split_k (r03):
//   * a tail that would be cut in two, K <= 8192: its tiles run as 128 x 256 HALF TILES with full K through the 128-row body
//     (gemm8m_body) inside this kernel - no slabs, no reduction, the unsplit summation order;
//   * else the K range of the tail tiles is split over S workgroups; partial accumulators go through an fp32 slab in a
//     caller-provided workspace and the last arriver (agent-scope release / acquire + arrival counter) sums them - for S > 2 all
//     S slabs in index order, so the arrival order never shows in the result - and runs the epilogue;
//   * a handful of split units (R x S <= CUs / 4) is dispatched FIRST, beside the first full tiles, instead of as a last round.
#include <atomic>
#include <mutex>
#include <type_traits>

#include "gemm_core.h"
#include "gemm_split.h"
#include "rtv_internal.h"

namespace rtv {

namespace g8 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;     // 16 KiB: 128 rows of one operand
constexpr int A_OFF = 0;                     // 3 K-tile buffers x 2 halves
constexpr int W_OFF = 6 * HALF_BYTES;        // 2 K-tile buffers x 2 halves
constexpr int LDS_BYTES = 10 * HALF_BYTES;   // 160 KiB
constexpr int THREADS = 512;
// swizzled 16-byte chunk position inside a 128-byte row (involution; conflict-free ds_read_b128)
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
}  // namespace g8

#ifdef RTV_GEMM_TIMELINE   // lab builds only (scripts/micro/build_gemm_timeline.sh): per-workgroup start / end stamps
__device__ unsigned long long* g_timeline = nullptr;   // [grid][4]: realtime start, after K loop, end, (seg << 32 | tile)
#endif

// tile id -> (row tile, column tile): 8-row supertiles, so that the 32 tiles an XCD runs at a time (consecutive ids, xcd_remap)
// share A / W panels through its L2.  RTV_G8_BALANCED_GROUPS (A/B build): the rows balanced over ceil(tiles_m / 8) groups (19 row
// tiles: 7 + 6 + 6 instead of 8 + 8 + 3) - a 3-row group makes an XCD-round 3 x 10.7 tiles = 13.7 operand panels instead of 12
// (scripts/gemm_traffic_model.py: qk 3.74 -> 3.54 x, ffn-in 4.40 -> 4.04 x the algorithmic bytes).  Same arithmetic per tile.
__device__ __forceinline__ void g8_tile_mn(int tiles_m, int tiles_n, int tile_id, int* mt, int* nt) {
#if RTV_G8_BALANCED_GROUPS
  const int ngrp = (tiles_m + 7) >> 3;
  const int gq = tiles_m / ngrp, gr = tiles_m - gq * ngrp;   // the first gr groups have gq + 1 rows
  const int big = (gq + 1) * tiles_n;
  int first_m, gm, in_group;
  if (tile_id < gr * big) {
    const int group = tile_id / big;
    gm = gq + 1;
    first_m = group * gm;
    in_group = tile_id - group * big;
  } else {
    const int t = tile_id - gr * big, sm = gq * tiles_n, group = t / sm;
    gm = gq;
    first_m = gr * (gq + 1) + group * gq;
    in_group = t - group * sm;
  }
#else
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * tiles_n;
  const int group = tile_id / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(tiles_m - first_m, GROUP_M);
  const int in_group = tile_id - group * per_group;
#endif
  *mt = first_m + in_group % gm;
  *nt = in_group / gm;
}

// (the 128 x 256 body, defined below: gemm8_kernel runs the half-tile units of its tail round through it)
template <bool F16>
__device__ __forceinline__ void gemm8m_body(const GemmParams& p, const SplitArgs& sp, int m0, int n0, int tile_id, int seg, int unit,
                                            int kt_begin, int kt_end, bool is_split);

template <bool F16, bool SKIP_IDLE>
__global__ __launch_bounds__(g8::THREADS, 2) void gemm8_kernel(GemmParams p, SplitArgs sp) {
  using namespace g8;
  int bid0 = blockIdx.x;
  // ---- ragged last row of tiles as 128 x 512 strips (sp.pair_units): the FIRST block ids - equal units that start together, then
  //      the tile grid runs in lockstep as before (a workgroup that finishes a cheaper unit in the middle of a round runs ahead of
  //      the tiles it shares A / W panels with and streams them alone: profiles/r04_gemm_ragged_row.log)
  if (sp.pair_units) {
    if (bid0 < sp.pair_pad) {
      if (bid0 >= sp.pair_units) return;   // padding id
#pragma unroll 1
      for (int j = 0; j < 2; ++j) {
        const int tn = 2 * bid0 + j;
        if (tn >= p.tiles_n) break;
        if (j) __syncthreads();            // the first half tile's epilogue has left the LDS
        gemm8m_body<F16>(p, sp, sp.pair_m0, tn * BN, 0, 0, -1, 0, p.K / BK, false);
      }
      return;
    }
    bid0 -= sp.pair_pad;
  }
  // ---- tail round as half tiles (sp.half_tail): block ids >= first_unit (after the split_first rotation) are 128 x 256 units
  if (sp.half_tail) {
    const int nhu = 2 * sp.tail_tiles;
    if (sp.split_first) bid0 = bid0 < nhu ? sp.first_unit + bid0 : bid0 - nhu;
    if (bid0 >= sp.first_unit) {
      const int v = xcd_remap(bid0 - sp.first_unit, nhu);   // the two halves of a tile side by side on one XCD (they share W)
      const int tile_id = sp.first_unit + (v >> 1);
      int mt, nt;
      g8_tile_mn(p.tiles_m, p.tiles_n, tile_id, &mt, &nt);
      const int m0h = mt * BM + (v & 1) * 128;
      if (m0h >= p.M) return;                                // the lower half of a ragged last row of tiles has no rows
      gemm8m_body<F16>(p, sp, m0h, nt * BN, tile_id, 0, -1, 0, p.K / BK, false);
      return;
    }
    sp.S = 1;             // the full tiles below: plain mapping of bid0
    sp.split_first = 0;
  }
#ifdef RTV_GEMM_TIMELINE
  const unsigned long long tl_t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long tl_t1 = 0;
#endif
  typedef TileCfg<256, 256, 64, 2, 4> Cfg;  // epilogue geometry: 4 x 2 blocks of 32x32 per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, g = lane >> 5;

  // ---- workgroup -> (tile, K segment)
  const int nk_total = p.K / BK;
  int tile_id, seg, unit, kt_begin, kt_end;
  const bool is_split = split_unit_of_block(sp, bid0, nk_total, &tile_id, &unit, &seg, &kt_begin, &kt_end);
  // tile id -> (m, n): GROUP_M-row supertiles so concurrently running tiles share A / W panels in L2
  int mt_, nt_;
  g8_tile_mn(p.tiles_m, p.tiles_n, tile_id, &mt_, &nt_);
  const int m0 = mt_ * BM;
  const int n0 = nt_ * BN;
  // ---- DMA geometry: a half-tile is 2 wave-wide pieces per wave; piece j of wave w fills rows j*64 + w*8 .. +8
  //      (lane -> row + lane/8, chunk slot lane%8), source chunk pre-swizzled.  BYTE offsets at k = 0.
  uint32_t src_off[4][2];  // [A0, A1, W0, W1][j]
  {
    const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 64 + wave * 8 + rsub;
      const int ch = swz(row, cpos) * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gm_row = min(m0 + h * 128 + row, p.M - 1);
        const int gn_row = min(n0 + h * 128 + row, p.N - 1);
        src_off[h][j] = ((uint32_t)gm_row * (uint32_t)p.lda + ch) * 2u;
        src_off[2 + h][j] = ((uint32_t)gn_row * (uint32_t)p.ldw + ch) * 2u;
      }
    }
  }
  // raw buffer descriptors over the whole operands (no bounds clamp needed: rows are clamped above, K is a multiple of 64)
  __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  // `chk` (std::true_type / false_type): whether the K-tile index still has to be compared with kt_end.  The steady-state
  // iterations (kt + 2 < kt_end) stage unconditionally - no scalar compare + branch around every DMA piece inside the loop.
  auto stage_piece = [&](int kt, int a3, int h, int j, auto chk) {  // h: 0..3 = A0, A1, W0, W1; a3 = kt % 3
    if (decltype(chk)::value && kt >= kt_end) return;
    const int slot = h < 2 ? A_OFF + (a3 * 2 + h) * HALF_BYTES : W_OFF + ((kt & 1) * 2 + (h - 2)) * HALF_BYTES;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(h < 2 ? rsrcA : rsrcW, (RTV_LDS void*)(smem + slot + (j * 64 + wave * 8) * 128), 16,
                                             src_off[h][j], (unsigned)kt * (BK * 2), 0, 0);
  };

  // ---- fragment addressing: A rows of this wave live in half `wr`, W rows in half wc >> 1
  const int b_row0 = (wc & 1) * 64;
  u32x4 af[2][4];   // current 64-row A half: [m-block][k-step]
  u32x4 bfr[2][4];  // both 32-column W blocks of the K-tile: [n-block][k-step]
  auto read_a = [&](int a3, int mq) {
    const char* s = smem + A_OFF + (a3 * 2 + wr) * HALF_BYTES;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int row = mq * 64 + mb * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[mb][ks] = *(const u32x4*)(s + row * 128 + (swz(row, ks * 2 + g) << 4));
    }
  };
  auto read_w = [&](int wb) {
    const char* s = smem + W_OFF + (wb * 2 + (wc >> 1)) * HALF_BYTES;
#pragma unroll
    for (int nq = 0; nq < 2; ++nq) {
      const int row = b_row0 + nq * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bfr[nq][ks] = *(const u32x4*)(s + row * 128 + (swz(row, ks * 2 + g) << 4));
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

#define G8_FENCE() __builtin_amdgcn_sched_barrier(0)
#define G8_BARRIER()                  \
  do {                                \
    G8_FENCE();                       \
    __builtin_amdgcn_s_barrier();     \
    G8_FENCE();                       \
  } while (0)
#define G8_LDS_DONE()                                   \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    G8_FENCE();                                         \
  } while (0)

  // MFMA segment: 16 MFMA on the 64 x 64 half (mq; n0, n1) with 4 DMA pieces between them (halves h0, h0 + 1 of K-tile
  // st_kt): an LDS-DMA costs 60-180 issue cycles inside an LDS segment but ~10 behind an MFMA.
  // A wave whose 128 rows all lie beyond M (M = 4680: waves 4-7 of the last row of tiles) has nothing to multiply: it keeps
  // its barriers and its DMA duty (the pieces it stages are other waves' operands) and skips fragment reads and MFMAs.
  const bool idle_rows = SKIP_IDLE && m0 + wr * 128 >= p.M;   // wave-uniform
  auto mma_half = [&](int mq, int st_kt, int st_a3, int h0, auto chk) {
    __builtin_amdgcn_s_setprio(1);
    int n = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          acc[mq * 2 + mb][nq] = Mfma32<F16>::run(bfr[nq][ks], af[mb][ks], acc[mq * 2 + mb][nq]);
          ++n;
          if ((n & 3) == 2) {
            G8_FENCE();
            stage_piece(st_kt, st_a3, h0 + (n >> 3), (n >> 2) & 1, chk);
            G8_FENCE();
          }
        }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue: the first two K-tiles complete (what X / Y of two earlier tiles would have staged)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int hh = 0; hh < 4; ++hh) {
      const int h = (hh + 2) & 3;   // W0, W1, A0, A1
      stage_piece(kt_begin + t, t, h, 0, std::true_type{});
      stage_piece(kt_begin + t, t, h, 1, std::true_type{});
    }
  if (kt_begin + 1 < kt_end) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G8_BARRIER();
  if (wr == 1) G8_BARRIER();  // stagger: group 1 runs one barrier behind group 0

  int a3 = 0;  // (kt - kt_begin) % 3
  auto k_tile = [&](const int kt, auto chk) {
    constexpr bool CHK = decltype(chk)::value;
    const int a3n = a3 == 0 ? 2 : a3 - 1;   // A buffer of K-tile kt + 2
    // ---------------- phase X: (m0; n0, n1); stages A0, A1 of K-tile kt + 2 (their buffer was last read in Y(kt - 1))
    read_w(kt & 1);
    read_a(a3, 0);
    G8_LDS_DONE();
    G8_BARRIER();
    mma_half(0, kt + 2, a3n, 0, chk);
    G8_BARRIER();
    // ---------------- phase Y: (m1; n0, n1); stages W0, W1 of K-tile kt + 2 (this tile's W buffer, last read in X(kt)).
    //                  The counted wait leaves the 4 pieces of X(kt) in flight and retires W(kt + 1) and everything older:
    //                  X(kt + 1) reads A(kt + 1) / W(kt + 1) one phase later.
    read_a(a3, 1);
    if (!CHK || kt + 2 < kt_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G8_LDS_DONE();
    G8_BARRIER();
    mma_half(1, kt + 2, a3n, 2, chk);
    G8_BARRIER();
    a3 = a3 == 2 ? 0 : a3 + 1;
  };
  // The same K-tile for a wave with nothing to multiply: barriers, DMA pieces and counted waits at the same points, no
  // fragment reads, no MFMAs.  (A separate loop: with the test inside k_tile the six scalar branches per K-tile cost every
  // tile 3-4 %, profiles/r03_gemm_idle_waves_ab.log.)
  auto idle_tile = [&](const int kt) {
    const int a3n = a3 == 0 ? 2 : a3 - 1;
    G8_BARRIER();
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece(kt + 2, a3n, i >> 1, i & 1, std::true_type{});
    G8_BARRIER();
    if (kt + 2 < kt_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G8_BARRIER();
#pragma unroll
    for (int i = 0; i < 4; ++i) stage_piece(kt + 2, a3n, 2 + (i >> 1), i & 1, std::true_type{});
    G8_BARRIER();
    a3 = a3 == 2 ? 0 : a3 + 1;
  };
  int kt = kt_begin;
  if (idle_rows) {
    for (; kt < kt_end; ++kt) idle_tile(kt);
  } else {
    for (; kt + 2 < kt_end; ++kt) k_tile(kt, std::false_type{});   // steady state: K-tile kt + 2 exists
    for (; kt < kt_end; ++kt) k_tile(kt, std::true_type{});        // last two K-tiles
  }
  if (wr == 0) G8_BARRIER();  // group 0 closes the stagger: every LDS read and DMA of the loop is retired

  // ---- split-K fix-up: publish the partial tile, last arriver reduces (placement-independent agent-scope
  //      release/acquire; the slab is a per-lane register image, so the reduce is a plain elementwise add)
#ifdef RTV_GEMM_TIMELINE
  tl_t1 = __builtin_amdgcn_s_memrealtime();
  bool reducer = true;
  if (is_split) reducer = split_k_reduce<4>(acc, sp, unit, seg, tile_id, smem, tid, wave, lane);
  if (!reducer) {
    if (g_timeline && tid == 0) {
      unsigned long long* t = g_timeline + (size_t)blockIdx.x * 4;
      t[0] = tl_t0;
      t[1] = tl_t1;
      t[2] = __builtin_amdgcn_s_memrealtime();
      t[3] = ((unsigned long long)(seg + 1) << 32) | (unsigned)tile_id;
    }
    return;
  }
#else
  if (is_split && !split_k_reduce<4>(acc, sp, unit, seg, tile_id, smem, tid, wave, lane)) return;
#endif

  // ---- epilogue through LDS when the output / residual rows allow 16-byte accesses (always on the DiT path)
  const bool wide = !((p.ldc | (p.residual ? p.ldr : 0)) & 7) && !(((uintptr_t)p.C | (uintptr_t)p.residual) & 15);
  if (wide) {
    if (is_split) __syncthreads();   // the reducer's flag word lives in smem[0..4)
#ifdef RTV_GEMM_TIMELINE
    unsigned long long tl_mid = 0;
    store_tile_lds<F16, 4>(p, m0 + wr * 128, n0 + wc * 64, lane, smem + wave * (128 * 128), acc, &tl_mid);
    if (g_timeline && tid == 0) g_timeline[(size_t)blockIdx.x * 4 + 1] = tl_mid | (1ull << 63);   // replaces the K-loop stamp
    tl_t1 = tl_t1 ? tl_t1 : 0;
    if (g_timeline && tid == 0) {
      unsigned long long* t = g_timeline + (size_t)blockIdx.x * 4;
      t[0] = tl_t0;
      t[2] = __builtin_amdgcn_s_memrealtime();
      t[3] = ((unsigned long long)(is_split ? seg + 1 : 0) << 32) | (unsigned)tile_id;
      g_timeline[(size_t)(gridDim.x + blockIdx.x) * 4] = tl_t1;    // K-loop end in the second half of the buffer
    }
    return;
#else
    store_tile_lds<F16, 4>(p, m0 + wr * 128, n0 + wc * 64, lane, smem + wave * (128 * 128), acc);
#endif
  } else {
    store_tile<F16, Cfg>(p, m0 + wr * 128, n0 + wc * 64, lane, acc);
  }
#ifdef RTV_GEMM_TIMELINE
  if (g_timeline && tid == 0) {
    unsigned long long* t = g_timeline + (size_t)blockIdx.x * 4;
    t[0] = tl_t0;
    t[1] = tl_t1;
    t[2] = __builtin_amdgcn_s_memrealtime();
    t[3] = ((unsigned long long)(is_split ? seg + 1 : 0) << 32) | (unsigned)tile_id | (is_split ? 0x80000000u : 0u);
  }
#endif
}

// ---------------------------------------------------------------- 128-row variant (token shards of context parallelism)
// The same ping-pong on a 128 x 256 tile for problems with few rows (585-row token shards at 8-way context parallelism:
// three 256-row tiles waste 24 % of their rows and leave most CUs idle).  A wave owns 64 x 64 of the tile, so a K-tile is
// ONE 16-MFMA segment per wave: two barrier intervals per K-tile,
//     interval 2t: g0 LDS(t) [A m0,m1 + W n0,n1: 16 reads] | g1 MFMA(t-1)        interval 2t+1: g0 MFMA(t) | g1 LDS(t)
// and every operand has THREE K-tile buffers (3 x {A 16 KiB, W 2 x 16 KiB} = 144 KiB).  A tile read in interval 2t (g0) /
// 2t+1 (g1) is dead from 2t+2 on.  Because group 1 runs one interval later but group 0 reads a tile first, group 1 stages
// one K-tile further ahead: MFMA(t) stages K-tile t + 2 + group (6 pieces per wave) into buffer (t + 2 + group) % 3, then
// waits with vmcnt(6) - everything but the pieces it has just issued - so each piece is in flight for two to three
// intervals before the counted wait retires it, one barrier before its first reader.
static std::atomic<bool> g_gemm8_skip_idle{true};   // rtv_gemm_set_skip_idle(0): A/B (include/rtv_hip_lab.h)
static std::atomic<bool> g_gemm8_half_tail{true};   // rtv_gemm_set_half_tail(0): K-segment tail instead (A/B)
static std::atomic<bool> g_gemm8_ragged_strips{true}; // rtv_gemm_set_ragged_strips(0): the ragged last row of tiles stays in the tile grid
namespace g8m {
constexpr int BM = 128, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;     // 16 KiB
constexpr int LDS_BYTES = 9 * HALF_BYTES;    // 144 KiB: 3 buffers x {A, W rows 0-127, W rows 128-255}
}  // namespace g8m

template <bool F16>
__device__ __forceinline__ void gemm8m_body(const GemmParams& p, const SplitArgs& sp, int m0, int n0, int tile_id, int seg, int unit,
                                            int kt_begin, int kt_end, bool is_split) {
  using namespace g8m;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, g = lane >> 5;

  uint32_t src_off[3][2];  // [A, W0, W1][piece]: byte offsets at k = 0
  {
    const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 64 + wave * 8 + rsub;
      const int ch = g8::swz(row, cpos) * 8;
      src_off[0][j] = ((uint32_t)min(m0 + row, p.M - 1) * (uint32_t)p.lda + ch) * 2u;
      src_off[1][j] = ((uint32_t)min(n0 + row, p.N - 1) * (uint32_t)p.ldw + ch) * 2u;
      src_off[2][j] = ((uint32_t)min(n0 + 128 + row, p.N - 1) * (uint32_t)p.ldw + ch) * 2u;
    }
  }
  __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  auto stage_piece = [&](int kt, int b3, int h, int j, auto chk) {  // h: 0 = A, 1 = W rows 0-127, 2 = W rows 128-255; b3 = buffer
    if (decltype(chk)::value && kt >= kt_end) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(h == 0 ? rsrcA : rsrcW,
                                             (RTV_LDS void*)(smem + (b3 * 3 + h) * HALF_BYTES + (j * 64 + wave * 8) * 128), 16,
                                             src_off[h][j], (unsigned)kt * (BK * 2), 0, 0);
  };

  u32x4 af[2][4], bfr[2][4];
  auto read_frags = [&](int b3) {
    const char* sa = smem + (b3 * 3) * HALF_BYTES;
    const char* sw = smem + (b3 * 3 + 1 + (wc >> 1)) * HALF_BYTES;
#pragma unroll
    for (int nq = 0; nq < 2; ++nq) {
      const int row = (wc & 1) * 64 + nq * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) bfr[nq][ks] = *(const u32x4*)(sw + row * 128 + (g8::swz(row, ks * 2 + g) << 4));
    }
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int row = wr * 64 + mb * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[mb][ks] = *(const u32x4*)(sa + row * 128 + (g8::swz(row, ks * 2 + g) << 4));
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // ---- prologue: K-tiles 0 and 1 by everybody, K-tile 2 by group 1 (what its MFMA(-1) would have staged)
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      stage_piece(kt_begin + t, t, h, 0, std::true_type{});
      stage_piece(kt_begin + t, t, h, 1, std::true_type{});
    }
  if (wr == 1) {
#pragma unroll
    for (int h = 0; h < 3; ++h) {
      stage_piece(kt_begin + 2, 2, h, 0, std::true_type{});
      stage_piece(kt_begin + 2, 2, h, 1, std::true_type{});
    }
  }
  // group 0: K-tile 0 landed, K-tile 1 stays in flight (retired at the end of its MFMA(0)).  Group 1 stands where the end of
  // its MFMA(-1) would be: everything but the youngest tile (2) retired - group 0 reads K-tile 1 in interval 2, before group 1's
  // first counted wait.
  if (wr == 1) {
    if (kt_begin + 2 < kt_end) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (kt_begin + 1 < kt_end) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  G8_BARRIER();
  if (wr == 1) G8_BARRIER();  // stagger: group 1 runs one barrier behind group 0

  int b3 = 0;  // buffer of K-tile kt = (kt - kt_begin) % 3
  auto k_tile = [&](const int kt, auto chk) {
    constexpr bool CHK = decltype(chk)::value;
    // ---------------- LDS segment
    read_frags(b3);
    G8_LDS_DONE();
    G8_BARRIER();
    // ---------------- MFMA segment: 16 MFMA + the 6 pieces of K-tile kt + 2 + group, then the counted wait
    const int st_kt = kt + 2 + wr;
    const int st_b = wr ? b3 : (b3 == 0 ? 2 : b3 - 1);   // (kt + 3) % 3 = b3 for group 1, (kt + 2) % 3 for group 0
    __builtin_amdgcn_s_setprio(1);
    int n = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          acc[mb][nq] = Mfma32<F16>::run(bfr[nq][ks], af[mb][ks], acc[mb][nq]);
          ++n;
          if (n == 2 || n == 4 || n == 7 || n == 10 || n == 12 || n == 15) {
            const int pc = n == 2 ? 0 : n == 4 ? 1 : n == 7 ? 2 : n == 10 ? 3 : n == 12 ? 4 : 5;
            G8_FENCE();
            stage_piece(st_kt, st_b, pc >> 1, pc & 1, chk);
            G8_FENCE();
          }
        }
    __builtin_amdgcn_s_setprio(0);
    G8_FENCE();
    // retire everything but the pieces just issued: K-tile kt + 1 (+ group) becomes readable after the next barrier
    if (!CHK || st_kt < kt_end) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G8_BARRIER();
    b3 = b3 == 2 ? 0 : b3 + 1;
  };
  int kt = kt_begin;
  for (; kt + 3 < kt_end; ++kt) k_tile(kt, std::false_type{});   // steady state: K-tile kt + 3 exists
  for (; kt < kt_end; ++kt) k_tile(kt, std::true_type{});
  if (wr == 0) G8_BARRIER();  // group 0 closes the stagger

  if (is_split && !split_k_reduce<2>(acc, sp, unit, seg, tile_id, smem, tid, wave, lane)) return;

  const bool wide = !((p.ldc | (p.residual ? p.ldr : 0)) & 7) && !(((uintptr_t)p.C | (uintptr_t)p.residual) & 15);
  if (wide) {
    if (is_split) __syncthreads();
    store_tile_lds<F16, 2>(p, m0 + wr * 64, n0 + wc * 64, lane, smem + wave * (64 * 128), acc);
  } else {
    typedef TileCfg<128, 256, 64, 2, 4> CfgM;
    store_tile<F16, CfgM>(p, m0 + wr * 64, n0 + wc * 64, lane, acc);
  }
}

template <bool F16>
__global__ __launch_bounds__(g8::THREADS, 2) void gemm8m_kernel(GemmParams p, SplitArgs sp) {
  using namespace g8m;
  const int nk_total = p.K / BK;
  int tile_id, seg, unit, kt_begin, kt_end;
  const bool is_split = split_unit_of_block(sp, blockIdx.x, nk_total, &tile_id, &unit, &seg, &kt_begin, &kt_end);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = tile_id / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = tile_id - group * per_group;
  gemm8m_body<F16>(p, sp, (first_m + in_group % gm) * BM, (in_group / gm) * BN, tile_id, seg, unit, kt_begin, kt_end, is_split);
}

// ---------------------------------------------------------------- split-K workspaces (caller-owned)
// One workspace = fp32 partial-tile slabs + arrival counters.  A workspace must never be shared by two launches that can run
// at the same time, so they are registered per (device, stream): a launch on stream s of device d uses the workspace attached
// to exactly (d, s), else the device-wide one attached with rtv_gemm_set_workspace (fine while ONE stream per device issues
// split-K GEMMs), else it runs without split-K.
struct SplitWorkspace {
  int device;
  hipStream_t stream;
  bool any_stream;
  float* slabs;
  int* counters;
};
constexpr int MAX_WORKSPACES = 64;
constexpr int MAX_DEVICES = 64;
static SplitWorkspace g_ws[MAX_WORKSPACES];
static int g_num_ws = 0;
static std::mutex g_ws_mu;

static int attach_workspace(bool any_stream, hipStream_t stream, void* ptr, size_t bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return set_error(-1, "gemm_set_workspace: cannot query the device");
  std::lock_guard<std::mutex> lk(g_ws_mu);
  int slot = -1;
  for (int i = 0; i < g_num_ws; ++i)
    if (g_ws[i].device == dev && g_ws[i].any_stream == any_stream && (any_stream || g_ws[i].stream == stream)) slot = i;
  if (!ptr) {  // detach
    if (slot >= 0) g_ws[slot] = g_ws[--g_num_ws];
    return 0;
  }
  if (((uintptr_t)ptr) & 255) return set_error(-1, "gemm_set_workspace: pointer must be 256-byte aligned");
  if (bytes < rtv_gemm_workspace_bytes()) return set_error(-1, "gemm_set_workspace: too small (rtv_gemm_workspace_bytes)");
  if (slot < 0) {
    if (g_num_ws == MAX_WORKSPACES) return set_error(-1, "gemm_set_workspace: too many workspaces attached");
    slot = g_num_ws++;
  }
  SplitWorkspace w{dev, stream, any_stream, (float*)ptr, (int*)((char*)ptr + (size_t)SPLIT_MAX_UNITS * SPLIT_SLAB_FLOATS * 4)};
  // arrival counters start at zero and every launch leaves them at zero (the reducer of a tile resets its counter)
  if (hipMemset(w.counters, 0, (size_t)SPLIT_MAX_UNITS * sizeof(int)) != hipSuccess) {
    if (slot == g_num_ws - 1) --g_num_ws;
    return set_error(-1, "gemm_set_workspace: cannot zero the arrival counters");
  }
  g_ws[slot] = w;
  return 0;
}

}  // namespace rtv

using namespace rtv;

#ifdef RTV_GEMM_TIMELINE
extern "C" int rtv_gemm_debug_timeline(unsigned long long* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int rtv_gemm_set_half_tail(int on) {   // A/B switch (lab, tests): the half-tile tail round of gemm8
  rtv::g_gemm8_half_tail.store(on != 0, std::memory_order_relaxed);
  return 0;
}

extern "C" int rtv_gemm_set_ragged_strips(int on) {
  rtv::g_gemm8_ragged_strips.store(on != 0, std::memory_order_relaxed);
  return 0;
}

extern "C" int rtv_gemm_set_skip_idle(int on) {
  rtv::g_gemm8_skip_idle.store(on != 0, std::memory_order_relaxed);
  return 0;
}

extern "C" size_t rtv_gemm_workspace_bytes(void) {
  return (size_t)SPLIT_MAX_UNITS * SPLIT_SLAB_FLOATS * 4 + (size_t)SPLIT_MAX_UNITS * 4 + 256;
}

extern "C" int rtv_gemm_set_workspace(void* ptr, size_t bytes) { return attach_workspace(true, nullptr, ptr, bytes); }

extern "C" int rtv_gemm_set_stream_workspace(rtv_stream_t stream, void* ptr, size_t bytes) {
  return attach_workspace(false, (hipStream_t)stream, ptr, bytes);
}

namespace rtv {

int plan_split_k(int T, int nk, bool allow_split, SplitArgs* sp, int* grid, hipStream_t stream, bool allow_half, int extra_units) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return set_error(-1, "gemm: cannot query the device");
  float* slabs = nullptr;
  int* counters = nullptr;
  const int G = device_num_cus();
  if (G <= 0) return set_error(-1, "gemm: cannot query the device");
  {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    for (int i = 0; i < g_num_ws && allow_split; ++i) {
      const SplitWorkspace& w = g_ws[i];
      if (w.device != dev) continue;
      if (!w.any_stream && w.stream == stream) {   // exact (device, stream) match wins
        slabs = w.slabs;
        counters = w.counters;
        break;
      }
      if (w.any_stream && !slabs) {
        slabs = w.slabs;
        counters = w.counters;
      }
    }
  }
  *sp = SplitArgs{T, 1, nullptr, nullptr, 0, 0, 0, 0, 0, 0};
  *grid = T;
  const int R = (T + extra_units) % G <= T ? (T + extra_units) % G : 0;   // the tail units must be tiles
  // A tail that split-K would cut in TWO (R between G / 3 and G / 2 tiles): run it as 2 R half tiles (128 x 256, full K) instead -
  // the same parallelism without two prologues, a 256 KiB publish and a slab read per tile (tail cost 0.7-0.9 -> ~0.6 of a tile
  // time at K = 5120: o-projection 201 -> 193 us, QKV unchanged; at K = 13824 the K segments are long enough to win by 2 %:
  // profiles/r03_gemm_half_tail_ab.log); needs no workspace and keeps the unsplit summation order (bit-identical with config 4)
  if (allow_half && g_gemm8_half_tail.load(std::memory_order_relaxed) && R > 0 && T > R && G / R == 2 && nk <= 128) {
    sp->first_unit = T - R;
    sp->tail_tiles = R;
    sp->half_tail = 1;
    *grid = (T - R) + 2 * R;
    return 0;
  }
  if (allow_split && slabs && R > 0) {   // T < G (small M under context parallelism): every tile is a split tile
    int S = G / R;                 // the split units of the partial round still fit one round
    if (S > 8) S = 8;
    if (S > nk / 4) S = nk / 4;    // keep >= 4 K-tiles per segment
    if (S >= 2 && (size_t)R * S <= (size_t)SPLIT_MAX_UNITS) {
      sp->first_unit = T - R;
      sp->S = S;
      sp->slabs = slabs;
      sp->counters = counters;
      sp->tail_tiles = R;
      // a FEW split units (ffn-in at M = 4680: 1026 tiles = 4 rounds + 2 tiles -> 16 units) go first: behind the last full round
      // they are a fifth round on 16 of 256 CUs plus its publish / reduce; in front they run beside the first full tiles
      // (-3 % on that shape; with many units - 232 of the QKV projection - the order of round 2 stays: +3 % the other way)
      sp->split_first = (R * S * 4 <= G) ? 1 : 0;
      *grid = (T - R) + R * S;
    }
  }
  return 0;
}

template <bool F16, bool SKIP_IDLE>
static int launch_gemm8_t(GemmParams p, bool allow_split, hipStream_t stream) {
  p.tiles_m = (p.M + g8::BM - 1) / g8::BM;
  p.tiles_n = (p.N + g8::BN - 1) / g8::BN;
  auto kern = gemm8_kernel<F16, SKIP_IDLE>;
  static LdsAttr lds_attr;   // per device (a second GPU used from this process needs the attribute as well)
  if (int st = ensure_dynamic_lds((const void*)kern, g8::LDS_BYTES, &lds_attr, "gemm8")) return st;
  SplitArgs sp;
  int grid = 0;
  // Ragged last row of tiles (M = 4680: 72 real rows of 256) as 128 x 512 STRIPS in front of the tile grid (r04): two horizontally
  // adjacent half tiles through the 128-row body, one after the other, so the row costs tiles_n / 2 workgroup slots instead of
  // tiles_n.  Used only where that removes the tail round altogether: ffn-in at M = 4680 is 19 x 54 = 1026 tiles = 4 rounds + 2
  // tiles (round 3 ran the two first, as 16 split-K units: 536 us); as 972 tiles + 27 strips it is 1004 slots = four rounds, no
  // split, no slabs: 488-510 us (hipBLASLt 503-514).  Where a tail round remains anyway (QKV, o, ffn-out) the strips cost 3-8 %:
  // the old tail already runs a ragged tile as ONE half tile, and a 1.1-tile-time unit delays its CU's share of the tail.  Every
  // form that lets a workgroup finish a cheaper unit in the MIDDLE of a round (ragged tiles as single half tiles inside the grid)
  // lost 7-18 % on every shape: that workgroup runs ahead of the tiles it shares panels with (profiles/r04_gemm_ragged_row.log).
  // The strips keep the plain kernel's K order per output element (bit-identical with tile config 4).
  const int G = device_num_cus();
  const int last_rows = p.M - (p.tiles_m - 1) * g8::BM;
  const int T0 = p.tiles_m * p.tiles_n;
  int pair_units = 0, pair_pad = 0;
  if (allow_split && g_gemm8_ragged_strips.load(std::memory_order_relaxed) && p.tiles_m >= 2 && last_rows <= 128 && G > 0 && T0 > G) {
    const int pu = (p.tiles_n + 1) / 2, pp = (pu + 7) & ~7;   // ids [0, pp), pp % 8 == 0: a block's XCD is its id % 8 and the
    const int RA = T0 % G;                                    // tile section must start at a multiple of 8 (xcd_remap)
    if (RA > 0 && RA <= p.tiles_n - pp) {                     // the strips' saving swallows the whole tail round
      pair_units = pu;
      pair_pad = pp;
      p.tiles_m -= 1;
    }
  }
  if (int st = plan_split_k(p.tiles_m * p.tiles_n, p.K / g8::BK, allow_split, &sp, &grid, stream, /*allow_half=*/true, pair_pad))
    return st;
  if (pair_units) {
    sp.pair_units = pair_units;
    sp.pair_pad = pair_pad;
    sp.pair_m0 = p.tiles_m * g8::BM;
    grid += pair_pad;
  }
  ProfScope prof(F16 ? PROF_CONV : PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);
  note_kernel(DK_GEMM8_256x256);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(g8::THREADS), g8::LDS_BYTES, stream, p, sp);
  return check_launch("gemm8");
}

template <bool F16>
static int launch_gemm8m_t(GemmParams p, bool allow_split, hipStream_t stream) {
  p.tiles_m = (p.M + g8m::BM - 1) / g8m::BM;
  p.tiles_n = (p.N + g8m::BN - 1) / g8m::BN;
  auto kern = gemm8m_kernel<F16>;
  static LdsAttr lds_attr;   // per device (a second GPU used from this process needs the attribute as well)
  if (int st = ensure_dynamic_lds((const void*)kern, g8m::LDS_BYTES, &lds_attr, "gemm8m")) return st;
  SplitArgs sp;
  int grid = 0;
  if (int st = plan_split_k(p.tiles_m * p.tiles_n, p.K / g8m::BK, allow_split, &sp, &grid, stream)) return st;
  ProfScope prof(F16 ? PROF_CONV : PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);
  note_kernel(DK_GEMM8M_128x256);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(g8::THREADS), g8m::LDS_BYTES, stream, p, sp);
  return check_launch("gemm8m");
}

int launch_gemm8m(const GemmParams& p, bool f16, bool split, hipStream_t stream) {
  if ((size_t)p.M * p.lda * 2 > 0x7fffffffull || (size_t)p.N * p.ldw * 2 > 0x7fffffffull)
    return set_error(-1, "gemm8m: operand larger than 2 GiB");
  return f16 ? launch_gemm8m_t<true>(p, split, stream) : launch_gemm8m_t<false>(p, split, stream);
}

int launch_gemm8(const GemmParams& p, bool f16, bool split, hipStream_t stream) {
  // the buffer descriptors address the operands with 32-bit byte offsets
  if ((size_t)p.M * p.lda * 2 > 0x7fffffffull || (size_t)p.N * p.ldw * 2 > 0x7fffffffull)
    return set_error(-1, "gemm8: operand larger than 2 GiB");
  // the idle-wave build only where it has something to skip: the last row of tiles holds <= 128 real rows
  const int last_rows = p.M - (p.M - 1) / g8::BM * g8::BM;
  if (g_gemm8_skip_idle.load(std::memory_order_relaxed) && !f16 && last_rows <= 128) return launch_gemm8_t<false, true>(p, split, stream);
  return f16 ? launch_gemm8_t<true, false>(p, split, stream) : launch_gemm8_t<false, false>(p, split, stream);
}

}  // namespace rtv
