// 256x256 projection GEMM with FOUR waves of 128x128 (tile configs 8 / 78..): the gemm9 schedule at one wave per SIMD.
//
// Why: every byte a wave pulls out of LDS competes with its MFMAs for the SIMD; a 128x128 wave tile needs 512 B of
// fragments per MFMA 32x32x16 instead of the 768 B of the 128x64 tile of gemm8 / gemm9 (-33 % LDS traffic), at the price
// of 256 accumulator registers per lane, i.e. one wave per SIMD and no second wave to cover any stall.  The schedule
// therefore leaves nothing to the compiler: K is consumed in 32-deep slabs through the 5-slot 160 KiB LDS ring of
// gemm9; per phase a wave issues 32 MFMAs and, behind each of the first 24, exactly one memory instruction - the 16
// fragment reads of the NEXT slab (second register set) and the 8 global_load_lds pieces of the slab three phases ahead
// - so reads and DMA retire in the shadow of the MFMA pipe (scripts/micro/mfma_mix.hip: 12 reads + 4 DMA per 16 MFMAs
// cost +13 % at one wave per SIMD).  One s_barrier + counted s_waitcnt vmcnt(16) per phase.
#include "gemm_core.h"
#include "gemm_split.h"
#include "rtv_internal.h"

namespace rtv {

namespace g10 {
constexpr int BM = 256, BN = 256, BKS = 32;
constexpr int SLAB_BYTES = (BM + BN) * BKS * 2;  // 32 KiB
constexpr int NSLOT = 5;
constexpr int LDS_BYTES = NSLOT * SLAB_BYTES;    // 160 KiB
constexpr int THREADS = 256;
// s_waitcnt immediates (gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] at [15:14])
constexpr int WAIT_LGKM0 = 0xC07F;
constexpr int WAIT_VM0 = 0x0F70, WAIT_VM8 = 0x0F78, WAIT_VM16 = 0x4F70, WAIT_VM24 = 0x4F78;
}  // namespace g10

template <int V>
struct IntC {
  static constexpr int value = V;
};

template <bool F16, int ABL = 0>  // ABL (timing ablations, wrong results): 1 no DMA, 2 no LDS reads, 4 no barriers
__global__ __launch_bounds__(g10::THREADS) void gemm10_kernel(GemmParams p) {
  using namespace g10;
  typedef TileCfg<128, 256, 64, 2, 2> HalfCfg;  // epilogue geometry of one accumulator half: 2 x 4 blocks of 32x32
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, g = lane >> 5;

  // ---- workgroup -> tile: XCD-contiguous chunks of 8-row supertiles
  const int tile_id = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = tile_id / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = tile_id - group * per_group;
  const int m0 = (first_m + in_group % gm) * BM;
  const int n0 = (in_group / gm) * BN;
  const int nph = p.K / BKS;  // K % 64 == 0 -> even

  // ---- DMA geometry: a slab is 32 pieces of 16 rows x 64 B; wave w stages pieces 8w .. 8w+7 (waves 0,1: A rows,
  //      waves 2,3: W rows).  lane -> row piece*16 + lane/4, LDS chunk slot lane%4 <- source chunk slot ^ swizzle(row)
  const bool stage_w = wave >= 2;
  const uint16_t* gsrc = stage_w ? p.W : p.A;
  uint32_t src_off[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int row = (wave * 8 + q) * 16 + (lane >> 2);  // 0..511
    const int r256 = row & 255;
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const int grow = stage_w ? min(n0 + r256, p.N - 1) : min(m0 + r256, p.M - 1);
    src_off[q] = (uint32_t)grow * (uint32_t)(stage_w ? p.ldw : p.lda) + c * 8;
  }
  int st_slab = 0, st_slot = 0;  // next slab to stage and its ring slot
  auto stage_piece = [&](int q) {
    if (!(ABL & 1) && st_slab < nph)
      dma16(gsrc + (size_t)st_slab * BKS + src_off[q], smem + st_slot * SLAB_BYTES + (wave * 8 + q) * 1024);
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
    b_base[ks] = (256 + wc * 128 + l31) * 64 + (((ks * 2 + g) ^ sw) << 4);
  }
  u32x4 fa[2][4][2], fb[2][4][2];  // [register set][block][k-step]
  if (ABL & 2) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          fa[i][b][ks] = u32x4{(unsigned)lane * 2654435761u + b, 0x3f803f80u, (unsigned)ks, 0x3f80bf80u};
          fb[i][b][ks] = u32x4{0x3f803f80u, (unsigned)lane * 40503u + b, 0xbf803f80u, (unsigned)i};
        }
  }
  auto read_one = [&](int set, int slot, int i) {  // i-th of the 16 fragment reads of a slab
    if (ABL & 2) return;
    const char* s = smem + slot * SLAB_BYTES;
    const int ks = i >> 3, j = i & 7;
    if (j < 4) fb[set][j][ks] = *(const u32x4*)(s + b_base[ks] + j * 2048);
    else fa[set][j - 4][ks] = *(const u32x4*)(s + a_base[ks] + (j - 4) * 2048);
  };

  // two 512-byte accumulator arrays (a single 1-KiB array is not promoted to registers by the compiler)
  f32x16 acc_lo[2][4], acc_hi[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc_lo[mi][ni][r] = 0.f;
        acc_hi[mi][ni][r] = 0.f;
      }

#define G10_FENCE() __builtin_amdgcn_sched_barrier(0)
  auto mma_phase = [&](auto setc, int rd_slot_, bool do_read) {
    constexpr int set = decltype(setc)::value;
    int n = 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          if (mb < 2) acc_lo[mb][nb] = Mfma32<F16>::run(fb[set][nb][ks], fa[set][mb][ks], acc_lo[mb][nb]);
          else acc_hi[mb - 2][nb] = Mfma32<F16>::run(fb[set][nb][ks], fa[set][mb][ks], acc_hi[mb - 2][nb]);
          G10_FENCE();
          if (n < 16) {
            if (do_read) read_one(set ^ 1, rd_slot_, n);
          } else if (n < 24) {
            stage_piece(n - 16);
          }
          G10_FENCE();
          ++n;
        }
    stage_advance();
  };
  // retire the pieces of slab ph+1 (slabs ph+2, ph+3 may stay in flight: 8 pieces each), then meet the other waves
  auto phase_sync = [&](int ph) {
    G10_FENCE();
    if (ph + 3 < nph) __builtin_amdgcn_s_waitcnt(WAIT_VM16);
    else if (ph + 2 < nph) __builtin_amdgcn_s_waitcnt(WAIT_VM8);
    else __builtin_amdgcn_s_waitcnt(WAIT_VM0);
    if (!(ABL & 4)) __builtin_amdgcn_s_barrier();
    G10_FENCE();
  };
  auto lds_done = [&]() {
    __builtin_amdgcn_s_waitcnt(WAIT_LGKM0);
    G10_FENCE();
  };

  // ---- prologue: slabs 0..3 in flight, slab 0 landed and read
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_piece(q);
    stage_advance();
  }
  if (nph >= 4) __builtin_amdgcn_s_waitcnt(WAIT_VM24);
  else __builtin_amdgcn_s_waitcnt(WAIT_VM0);
  __builtin_amdgcn_s_barrier();
  G10_FENCE();
#pragma unroll
  for (int i = 0; i < 16; ++i) read_one(0, 0, i);
  lds_done();

  int rd_slot = 1;  // ring slot of slab ph+1
  for (int ph = 0; ph < nph; ph += 2) {
    phase_sync(ph);
    mma_phase(IntC<0>{}, rd_slot, true);  // reads slab ph+1 (exists: nph is even) into set 1
    rd_slot = (rd_slot == NSLOT - 1) ? 0 : rd_slot + 1;
    lds_done();

    phase_sync(ph + 1);
    mma_phase(IntC<1>{}, rd_slot, ph + 2 < nph);
    rd_slot = (rd_slot == NSLOT - 1) ? 0 : rd_slot + 1;
    lds_done();
  }
#undef G10_FENCE

  store_tile<F16, HalfCfg>(p, m0 + wr * 128, n0 + wc * 128, lane, acc_lo);
  store_tile<F16, HalfCfg>(p, m0 + wr * 128 + 64, n0 + wc * 128, lane, acc_hi);
}

template <bool F16, int ABL>
static int launch_gemm10_t(GemmParams p, hipStream_t stream) {
  p.tiles_m = (p.M + g10::BM - 1) / g10::BM;
  p.tiles_n = (p.N + g10::BN - 1) / g10::BN;
  auto kern = gemm10_kernel<F16, ABL>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, g10::LDS_BYTES);
    if (e != hipSuccess) return set_error(e, "gemm10: hipFuncSetAttribute");
    attr_set = true;
  }
  ProfScope prof(F16 ? PROF_CONV : PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(g10::THREADS), g10::LDS_BYTES, stream, p);
  return check_launch("gemm10");
}

int launch_gemm10(const GemmParams& p, bool f16, int abl, hipStream_t stream) {
  if (f16) return launch_gemm10_t<true, 0>(p, stream);
  switch (abl) {
    case 1: return launch_gemm10_t<false, 1>(p, stream);
    case 2: return launch_gemm10_t<false, 2>(p, stream);
    case 3: return launch_gemm10_t<false, 3>(p, stream);
    case 7: return launch_gemm10_t<false, 7>(p, stream);
    default: return launch_gemm10_t<false, 0>(p, stream);
  }
}

}  // namespace rtv
