// Dense projection GEMM for the DiT block (QKV / O / cross-q / cross-o / FFN / embeddings / head).
// See gemm_core.h for the tile structure.
#include "gemm_core.h"
#include "rtv_internal.h"

namespace rtv {

template <bool F16, int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmParams p) {
  typedef TileCfg<BM, BN, BK, WM, WN> Cfg;
  static_assert(Cfg::TN == 2 && Cfg::NW * Cfg::TM * 32 * 128 <= 2 * Cfg::STAGE_BYTES, "LDS epilogue geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- workgroup -> output tile (XCD-contiguous chunks, then GROUP_M x tiles_n supertiles so the
  //      tiles that run concurrently on one XCD share A and W panels through its L2)
  const int nwg = p.tiles_m * p.tiles_n;
  int id = xcd_remap(blockIdx.x, nwg);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = id / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = id - group * per_group;
  const int tile_m = first_m + in_group % gm;
  const int tile_n = in_group / gm;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-lane DMA source offsets (elements), K offset 0
  const int cpos = lane % Cfg::CH;   // chunk slot this lane fills in LDS
  const int rsub = lane / Cfg::CH;   // row inside the 64-lane DMA group
  uint32_t a_off[Cfg::A_INST], b_off[Cfg::B_INST];
#pragma unroll
  for (int i = 0; i < Cfg::A_INST; ++i) {
    int row = (wave * Cfg::A_INST + i) * Cfg::RPI + rsub;
    int gm_row = min(m0 + row, p.M - 1);
    a_off[i] = (uint32_t)gm_row * (uint32_t)p.lda + Cfg::swz(row, cpos) * 8;
  }
#pragma unroll
  for (int i = 0; i < Cfg::B_INST; ++i) {
    int row = (wave * Cfg::B_INST + i) * Cfg::RPI + rsub;
    int gn_row = min(n0 + row, p.N - 1);
    b_off[i] = (uint32_t)gn_row * (uint32_t)p.ldw + Cfg::swz(row, cpos) * 8;
  }

  auto stage = [&](int kt, int buf) {
    char* sA = smem + buf * Cfg::STAGE_BYTES;
    char* sB = sA + Cfg::A_BYTES;
    const uint16_t* Ak = p.A + (size_t)kt * BK;
    const uint16_t* Wk = p.W + (size_t)kt * BK;
#pragma unroll
    for (int i = 0; i < Cfg::A_INST; ++i)
      dma16(Ak + a_off[i], sA + (wave * Cfg::A_INST + i) * 1024);
#pragma unroll
    for (int i = 0; i < Cfg::B_INST; ++i)
      dma16(Wk + b_off[i], sB + (wave * Cfg::B_INST + i) * 1024);
  };

  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int mi = 0; mi < Cfg::TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < Cfg::TN; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  const int wm = wave / WN, wn = wave % WN;
  const int a_row0 = wm * (BM / WM), b_row0 = wn * (BN / WN);

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // tile kt has landed (every wave drains its own DMA before the barrier) and every wave is done
    // reading buf^1 (it was consumed in iteration kt-1)
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
    const char* sA = smem + buf * Cfg::STAGE_BYTES;
    mma_stage<F16, Cfg, BK>(sA, sA + Cfg::A_BYTES, a_row0, b_row0, lane, acc);
  }

  // epilogue through LDS (coalesced 16-byte rows, gemm_core.h) whenever the output / residual rows allow it
  const bool wide = !((p.ldc | (p.residual ? p.ldr : 0)) & 7) && !(((uintptr_t)p.C | (uintptr_t)p.residual) & 15);
  if (wide) {
    __syncthreads();   // every wave is done reading the staged tiles
    store_tile_lds<F16, Cfg::TM>(p, m0 + a_row0, n0 + b_row0, lane, smem + wave * (Cfg::TM * 32 * 128), acc);
  } else {
    store_tile<F16, Cfg>(p, m0 + a_row0, n0 + b_row0, lane, acc);
  }
}

template <bool F16, int BM, int BN, int BK, int WM, int WN>
static int launch_cfg(GemmParams p, hipStream_t stream) {
  typedef TileCfg<BM, BN, BK, WM, WN> Cfg;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  auto kern = gemm_kernel<F16, BM, BN, BK, WM, WN>;
  static LdsAttr lds_attr;   // per device
  const int lds = 2 * Cfg::STAGE_BYTES;
  if (int st = ensure_dynamic_lds((const void*)kern, lds, &lds_attr, "gemm")) return st;
  ProfScope prof(F16 ? PROF_CONV : PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);  // fp16 GEMMs belong to the VAE
  note_kernel(BM == 128 && BN == 128 ? DK_GEMM_128x128 : DK_GEMM_OTHER_CFG);
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(Cfg::NT), lds, stream, p);
  return check_launch("gemm");
}

int launch_gemm(const GemmParams& p, int dtype, int tile_cfg, hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return set_error(-1, "gemm: empty problem");
  if (p.K % 64 != 0) return set_error(-1, "gemm: K must be a multiple of 64");
  if (p.N % 8 != 0) return set_error(-1, "gemm: N must be a multiple of 8");
  if ((p.lda % 8) || (p.ldw % 8) || (p.ldc % 4) || (p.residual && (p.ldr % 4)))
    return set_error(-1, "gemm: leading dimensions must keep 16-byte (A,W) / 8-byte (C,res) alignment");
  if (p.gate && p.rows_per_frame <= 0) return set_error(-1, "gemm: gate needs rows_per_frame");
  const bool f16 = dtype == RTV_DTYPE_F16;
  if (dtype != RTV_DTYPE_BF16 && dtype != RTV_DTYPE_F16) return set_error(-1, "gemm: dtype");
  if (tile_cfg == 0) {
    // default: the ping-pong kernels (+ split-K of the tail round) once there is enough work for their big tiles - the
    // 256x256 one, or its 128x256 variant when 256-row tiles would waste many rows or leave the chip under-filled (token
    // shards of context parallelism: 585 rows = 3 x 256 wastes 24 %, 5 x 128 8.6 %) - the 128x128 kernel (2 workgroups per
    // CU) otherwise.  Thresholds from profiles/r01_kbench_*, profiles/r02_gemm_shapes_*.
    const long tiles_n = (p.N + 255) / 256;
    const long rt256 = (p.M + 255) / 256, rt128 = (p.M + 127) / 128;
    const long tiles256 = rt256 * tiles_n, tiles128 = rt128 * tiles_n;
    // rows the kernels really multiply: gemm8's waves whose 128 rows lie beyond M run the idle loop (the upper half of the last
    // row of tiles costs barriers and DMA duty only), so 585 rows are 640 for both kernels
    const long last256 = p.M - (rt256 - 1) * 256;
    const long eff256 = (rt256 - 1) * 256 + (last256 <= 128 ? 128 : 256), pad128 = rt128 * 128;
    // a SMALL tail round behind one to three full rounds of 256x256 tiles (300 = 256 + 44, 540 = 2 x 256 + 28) keeps the chip
    // waiting for a few split-K units; the same problem in 128-row tiles has a well filled tail (r03: M = 1170 / 2340 / 3120,
    // profiles/r03_gemm_small_m_dispatch.log: 6-26 %)
    const long G = device_num_cus() > 0 ? device_num_cus() : 256;   // the round size plan_split_k uses as well
    const long tail256 = tiles256 % G;
    const bool small_tail = tiles256 > G && tiles256 < 4 * G && tail256 != 0 && tail256 * 4 < G;
    // r05: up to 8 row tiles of 160 (the token shards of 4- and 8-way context parallelism: 585 / 1170 rows) whose 160 x 256 tiles fill
    // at least 4/5 of a round run on the one-wave-per-SIMD kernel (gemm5.hip): one round of EQUAL units instead of an under-filled
    // round of 256-row tiles or 1.17 rounds of 128-row ones - QKV 111 -> 83 us, ffn-in 93 -> 74 at 585 rows, 183 -> 154 / 180 -> 143
    // at 1170 (hipBLASLt: 95 / 79 / 179 / 236; profiles/r05_gemm5_cp_shapes.log).  Unsplit there: the K order of tile config 4.
    // Only where 160-row tiles multiply no more rows than the 256-row kernel would (585 -> 640 = 640, 1170 -> 1280 = 1280; but
    // 512 -> 640 > 512 and 1024 -> 1120 > 1024: row counts that tile exactly by 256 stay on the ping-pong kernels - the text
    // encoder's 512 x 20480 x 4096 would otherwise run 320 padded tiles = 1.25 rounds instead of 160 exact ones) and where a
    // second round, if any, is at least a quarter full (the rule ignored padding and round count).
    const long rt160 = (p.M + 159) / 160, tiles160 = rt160 * tiles_n, tail160 = tiles160 % G;
    const bool fill160 = tiles160 * 5 >= G * 4 && (tiles160 <= G || tail160 == 0 || tail160 * 4 >= G);
    if (!f16 && p.K >= 1024 && p.M <= 1280 && rt160 * 160 <= eff256 && fill160) return launch_gemm5(p, f16, true, stream);
    if (!f16 && p.K >= 1024 && tiles128 >= 96 && (pad128 * 100 < eff256 * 93 || tiles256 * 8 < G * 5 || small_tail))
      return launch_gemm8m(p, f16, true, stream);
    // (deep-K problems with 64..127 tiles - ffn2 on a context-parallel token shard - also win: every tile is then split
    // along K over the idle CUs, scripts/cp_gemm_shapes.py)
    if (!f16 && ((p.K >= 2048 && tiles256 >= 128) || tiles256 >= 640 || (p.K >= 8192 && tiles256 >= 64)))
      return launch_gemm8(p, f16, true, stream);
    tile_cfg = 1;
  }
  switch (tile_cfg) {
    case 1:
      return f16 ? launch_cfg<true, 128, 128, 64, 2, 2>(p, stream)
                 : launch_cfg<false, 128, 128, 64, 2, 2>(p, stream);
    case 2:
      return f16 ? launch_cfg<true, 256, 128, 64, 4, 2>(p, stream)
                 : launch_cfg<false, 256, 128, 64, 4, 2>(p, stream);
    case 3:
      return f16 ? launch_cfg<true, 256, 256, 64, 2, 4>(p, stream)
                 : launch_cfg<false, 256, 256, 64, 2, 4>(p, stream);
    case 4:
      return launch_gemm8(p, f16, false, stream);   // 256x256 ping-pong kernel, no split-K
    case 5:
    case 50:
      return launch_gemm8(p, f16, true, stream);    // + split-K of the tail round = the default for large problems
#ifdef RTV_LAB   // experimental kernels: only in the lab build (make LAB=1 -> librtv_hip_lab.so); unknown tile configs otherwise
    case 8:
      return launch_gemm4(p, f16, stream);          // 256x256, one wave per SIMD (lab: no split-K yet)
    case 81:
    case 82:
    case 83:
    case 84:
    case 85:
    case 86:
    case 87:
    case 89:
    case 90:
    case 91:
      return launch_gemm4(p, f16, stream, tile_cfg - 80);   // timing experiments (garbage results)
#endif
    case 9:
      return launch_gemm5(p, f16, true, stream);    // 160x256, one wave per SIMD (few-row problems: context-parallel token shards)
    case 19:
      return launch_gemm5(p, f16, false, stream);   // ... without split-K (bit-identical with tile config 4)
    case 6:
      return launch_gemm8m(p, f16, false, stream);  // 128x256 ping-pong kernel (few-row problems), no split-K
    case 7:
      return launch_gemm8m(p, f16, true, stream);   // + split-K
    default:
      return set_error(-1, "gemm: unknown tile config");
  }
}

}  // namespace rtv
