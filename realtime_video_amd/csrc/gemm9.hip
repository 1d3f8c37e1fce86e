// 256x256 software-pipelined projection GEMM for gfx950 (tile config 6): same math / epilogue / C ABI as gemm.hip and
// gemm8.hip, different schedule.
//
// The K dimension is consumed in 32-deep slabs ("phases").  One workgroup = 8 waves (2 x 4, wave tile 128 x 64), two
// waves per SIMD, all in the same phase.  LDS is a ring of 5 slabs x (256 A rows + 256 W rows) x 64 B = 160 KiB:
//   * slab p+4 is staged (global_load_lds, 4 wave-wide pieces per wave) between the MFMAs of phase p,
//   * the fragments of slab p+1 are read into a second register set at the start of phase p,
//   * phase p itself is 16 MFMA 32x32x16 per wave on registers loaded one phase earlier.
// So every memory operation is issued a full phase (LDS) or three phases (HBM/L2) before its consumer, and the only
// synchronisation is ONE s_barrier per phase (16 MFMAs per wave), preceded by a counted s_waitcnt vmcnt(8) that
// retires the 4 pieces of slab p+1 while slabs p+2 / p+3 stay in flight.
// Rows are 64 B in LDS; 16-byte chunk c of row r lives at chunk c ^ ((r >> 2) & 3) (conflict-free ds_read_b128 for the
// MFMA fragment pattern; the DMA writes lane-linear, so the swizzle is applied to the global source chunk).
#include "gemm_core.h"
#include "gemm_split.h"
#include "rtv_internal.h"

namespace rtv {

namespace g9 {
constexpr int BM = 256, BN = 256, BKS = 32;
constexpr int SLAB_BYTES = (BM + BN) * BKS * 2;  // 32 KiB
constexpr int LDS_BYTES = 5 * SLAB_BYTES;        // 160 KiB (ring of 5 slabs at the default prefetch distance)
constexpr int THREADS = 512;
// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]); the builtin (not
// inline asm) so that the compiler's own wait-count tracking sees them and does not add lgkmcnt(0) before the MFMAs
constexpr int WAIT_LGKM0 = 0xC07F;
constexpr int WAIT_VM0 = 0x0F70, WAIT_VM4 = 0x0F74, WAIT_VM8 = 0x0F78, WAIT_VM12 = 0x0F7C;
}  // namespace g9

template <int V>
struct IntC9 {
  static constexpr int value = V;
};

template <bool F16, int DIST, int ABL = 0>  // ABL (timing ablations, wrong results): 1 no DMA, 2 no LDS reads, 4 no barriers
__global__ __launch_bounds__(g9::THREADS, 2) void gemm9_kernel(GemmParams p, SplitArgs sp) {
  using namespace g9;
  constexpr int NSLOT = DIST + 2;  // slab being consumed (registers) + slab being read + DIST slabs in flight
  typedef TileCfg<256, 256, 64, 2, 4> Cfg;  // epilogue geometry: 4 x 2 blocks of 32x32 per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, g = lane >> 5;

  // ---- workgroup -> (tile, K segment): XCD-contiguous chunks of 8-row supertiles; the tiles of the last partial round
  //      are split along K (gemm_split.h)
  int tile_id, seg, unit, kt_begin, kt_end;
  const bool is_split = split_unit_of_block(sp, blockIdx.x, p.K / 64, &tile_id, &unit, &seg, &kt_begin, &kt_end);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = tile_id / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = tile_id - group * per_group;
  const int m0 = (first_m + in_group % gm) * BM;
  const int n0 = (in_group / gm) * BN;
  const int ph0 = 2 * kt_begin, nph = 2 * kt_end;  // this workgroup's slab range (even count)

  // ---- DMA geometry: a slab is 32 pieces of 16 rows x 64 B; wave w stages pieces 4w .. 4w+3 (waves 0-3: A rows,
  //      waves 4-7: W rows).  lane -> row piece*16 + lane/4, LDS chunk slot lane%4 <- source chunk slot ^ swizzle(row)
  const bool stage_w = wave >= 4;
  const uint16_t* gsrc = stage_w ? p.W : p.A;
  uint32_t src_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (wave * 4 + q) * 16 + (lane >> 2);  // 0..511
    const int r256 = row & 255;
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const int grow = stage_w ? min(n0 + r256, p.N - 1) : min(m0 + r256, p.M - 1);
    src_off[q] = (uint32_t)grow * (uint32_t)(stage_w ? p.ldw : p.lda) + c * 8;
  }
  int st_slab = ph0, st_slot = 0;  // next slab to stage and its ring slot
  auto stage_piece = [&](int q) {
    if (!(ABL & 1) && st_slab < nph) dma16(gsrc + (size_t)st_slab * BKS + src_off[q], smem + st_slot * SLAB_BYTES + (wave * 4 + q) * 1024);
  };
  auto stage_advance = [&]() {
    ++st_slab;
    st_slot = (st_slot == NSLOT - 1) ? 0 : st_slot + 1;
  };

  // ---- fragment addressing (byte offsets inside a slab; (row >> 2) & 3 only depends on the lane)
  const int sw = (l31 >> 2) & 3;
  int a_base[2], b_base[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_base[ks] = (wr * 128 + l31) * 64 + (((ks * 2 + g) ^ sw) << 4);
    b_base[ks] = (256 + wc * 64 + l31) * 64 + (((ks * 2 + g) ^ sw) << 4);
  }
  u32x4 fa[2][4][2], fb[2][2][2];  // [register set][block][k-step]
  if (ABL & 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) fa[i][mb][ks] = u32x4{(unsigned)lane * 2654435761u + mb, 0x3f803f80u, (unsigned)ks, 0x3f80bf80u};
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) fb[i][nb][ks] = u32x4{0x3f803f80u, (unsigned)lane * 40503u + nb, 0xbf803f80u, (unsigned)i};
      }
  }
  auto read_frags = [&](int set, int slot) {
    if (ABL & 2) return;
    const char* s = smem + slot * SLAB_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) fb[set][nb][ks] = *(const u32x4*)(s + b_base[ks] + nb * 2048);
#pragma unroll
      for (int mb = 0; mb < 4; ++mb) fa[set][mb][ks] = *(const u32x4*)(s + a_base[ks] + mb * 2048);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

#define G9_FENCE() __builtin_amdgcn_sched_barrier(0)
  // One phase: 16 MFMA on register set `set`, with ONE memory instruction issued behind each of them, in the shadow of
  // its 32-cycle execution: the 12 fragment reads of the next slab (into the other register set) after MFMA 1..12, the
  // 4 DMA pieces of the slab being staged after MFMA 13..16.  (Issued as bursts the same instructions cost their full
  // issue time: +22 % each for the reads and the DMA, scripts/abl_gemm.py.)
  auto read_one = [&](int set, int slot, int i) {  // i-th of the 12 fragment reads of a slab
    if (ABL & 2) return;
    const char* s = smem + slot * SLAB_BYTES;
    const int ks = i / 6, j = i % 6;
    if (j < 2) fb[set][j][ks] = *(const u32x4*)(s + b_base[ks] + j * 2048);
    else fa[set][j - 2][ks] = *(const u32x4*)(s + a_base[ks] + (j - 2) * 2048);
  };
  // DMA pieces go after MFMA 2, 6, 10, 14 (spread over the phase: 8 waves x 4 pieces share a 64 B/clk path), the 12
  // fragment reads after the other MFMAs.
  auto mma_phase = [&](auto setc, int rd_slot_, bool do_read) {
    constexpr int set = decltype(setc)::value;
    int n = 0, nr = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          acc[mb][nb] = Mfma32<F16>::run(fb[set][nb][ks], fa[set][mb][ks], acc[mb][nb]);
          G9_FENCE();
          if ((n & 3) == 1) {
            stage_piece(n >> 2);
          } else {
            if (do_read) read_one(set ^ 1, rd_slot_, nr);
            ++nr;
          }
          G9_FENCE();
          ++n;
        }
    stage_advance();
  };
  // retire the pieces of slab ph+1 (slabs ph+2, ph+3 may stay in flight), then meet the other waves
  auto phase_sync = [&](int ph) {
    G9_FENCE();
    // in flight after the wait: slabs ph+2 .. ph+DIST (4 pieces each)
    if (DIST >= 3 && ph + 3 < nph) __builtin_amdgcn_s_waitcnt(WAIT_VM8);
    else if (DIST >= 2 && ph + 2 < nph) __builtin_amdgcn_s_waitcnt(WAIT_VM4);
    else __builtin_amdgcn_s_waitcnt(WAIT_VM0);
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
    G9_FENCE();
  };
  auto lds_done = [&]() {
    __builtin_amdgcn_s_waitcnt(WAIT_LGKM0);
    G9_FENCE();
  };

  // ---- prologue: slabs 0..3 in flight, slab 0 landed and read
#pragma unroll
  for (int s = 0; s < DIST + 1; ++s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) stage_piece(q);
    stage_advance();
  }
  if (nph - ph0 >= DIST + 1) __builtin_amdgcn_s_waitcnt(DIST == 3 ? WAIT_VM12 : (DIST == 2 ? WAIT_VM8 : WAIT_VM4));
  else __builtin_amdgcn_s_waitcnt(WAIT_VM0);
  __builtin_amdgcn_s_barrier();
  G9_FENCE();
  read_frags(0, 0);
  lds_done();

  int rd_slot = 1;  // ring slot of slab ph+1
  for (int ph = ph0; ph < nph; ph += 2) {
    phase_sync(ph);
    mma_phase(IntC9<0>{}, rd_slot, true);     // reads slab ph+1 (exists: nph is even) into set 1
    rd_slot = (rd_slot == NSLOT - 1) ? 0 : rd_slot + 1;
    lds_done();

    phase_sync(ph + 1);
    mma_phase(IntC9<1>{}, rd_slot, ph + 2 < nph);
    rd_slot = (rd_slot == NSLOT - 1) ? 0 : rd_slot + 1;
    lds_done();
  }
#undef G9_FENCE

  if (is_split && !split_k_reduce(acc, sp, unit, seg, tile_id, smem, tid, wave, lane)) return;
  store_tile<F16, Cfg>(p, m0 + wr * 128, n0 + wc * 64, lane, acc);
}

template <bool F16, int DIST, int ABL = 0>
static int launch_gemm9_t(GemmParams p, bool allow_split, hipStream_t stream) {
  p.tiles_m = (p.M + g9::BM - 1) / g9::BM;
  p.tiles_n = (p.N + g9::BN - 1) / g9::BN;
  auto kern = gemm9_kernel<F16, DIST, ABL>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, g9::LDS_BYTES);
    if (e != hipSuccess) return set_error(e, "gemm9: hipFuncSetAttribute");
    attr_set = true;
  }
  SplitArgs sp;
  int grid = 0;
  if (int st = plan_split_k(p.tiles_m * p.tiles_n, p.K / 64, allow_split, &sp, &grid, stream)) return st;
  ProfScope prof(F16 ? PROF_CONV : PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(g9::THREADS), g9::LDS_BYTES, stream, p, sp);
  return check_launch("gemm9");
}

int launch_gemm9(const GemmParams& p, bool f16, bool split, int dist, hipStream_t stream) {
  if (f16) return launch_gemm9_t<true, 3>(p, split, stream);
  if (dist >= 70) {
    switch (dist - 70) {
      case 1: return launch_gemm9_t<false, 3, 1>(p, split, stream);
      case 2: return launch_gemm9_t<false, 3, 2>(p, split, stream);
      case 3: return launch_gemm9_t<false, 3, 3>(p, split, stream);
      case 4: return launch_gemm9_t<false, 3, 4>(p, split, stream);
      case 5: return launch_gemm9_t<false, 3, 5>(p, split, stream);
      case 6: return launch_gemm9_t<false, 3, 6>(p, split, stream);
      case 7: return launch_gemm9_t<false, 3, 7>(p, split, stream);
      default: break;
    }
  }
  if (dist == 1) return launch_gemm9_t<false, 1>(p, split, stream);
  if (dist == 2) return launch_gemm9_t<false, 2>(p, split, stream);
  return launch_gemm9_t<false, 3>(p, split, stream);
}

}  // namespace rtv
