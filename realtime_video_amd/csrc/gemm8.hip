// 256x256x64 "ping-pong" projection GEMM for gfx950 (tile configs 5 / 6): the high-throughput variant of
// gemm.hip for the large DiT linears (same math, same fused epilogue, same C ABI).
//
// One workgroup = 8 waves on one CU, two waves per SIMD.  Waves 0-3 ("group 0", output rows 0-127) and
// waves 4-7 ("group 1", rows 128-255) run the SAME phase program staggered by one s_barrier, so on every
// SIMD one wave is in its MFMA segment while its partner is in its LDS segment:
//
//     group 0:  lds(p) | B | mfma(p) | B | lds(p+1) | B | mfma(p+1) | B ...
//     group 1:       B | lds(p) | B | mfma(p)    | B | lds(p+1) | B ...
//
// A K-tile (64 deep) is 4 phases; a phase computes one 64x32 quadrant of the wave's 128x64 output
// (8 MFMA 32x32x16) and stages ONE 128x64 half-tile of a future K-tile with global_load_lds.  The two
// DMA pieces of a phase are issued BETWEEN MFMAs (an LDS-DMA costs 60-180 issue cycles inside an LDS
// segment but hides behind the 32-cycle MFMA issue slots: +15 % measured); the LDS segments carry only
// ds_read_b128s (8/4/8/4 per phase), retired right after the phase's first barrier.
// LDS: 2 K-tile buffers x {A rows 0-127, A rows 128-255, W rows 0-127, W rows 128-255} x 16 KiB = 128 KiB.
// DMA runs ~1.5 K-tiles ahead and is retired by COUNTED s_waitcnt vmcnt(2) (never 0 in steady state).
// Schedule per K-tile t (slots of buffer t&1 are re-filled for tile t+2):
//     p1: read A[m0]          mfma (m0,n0) + stage A0(t+1)     p2: read W[n1], vmcnt(2)   mfma (m0,n1) + stage A1(t+1)
//     p3: read A[m1]          mfma (m1,n1) + stage W0(t+2)     p4: read W[n0](t+1), vmcnt(2)  mfma (m1,n0) + stage W1(t+2)
//
// Tail-wave quantisation (e.g. 380 tiles on 256 CUs = 1.48 rounds) is removed by splitting the K range
// of the tiles of the last, partial round over S workgroups; partial accumulators go through an fp32 slab
// in a caller-provided workspace and the last arriver (agent-scope release / acquire + arrival counter)
// sums them and runs the epilogue.
#include <mutex>
#include <type_traits>

#include "gemm_core.h"
#include "gemm_split.h"
#include "rtv_internal.h"

namespace rtv {

namespace g8 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_ROWS = 128;
constexpr int HALF_BYTES = HALF_ROWS * BK * 2;  // 16 KiB
constexpr int LDS_BYTES = 8 * HALF_BYTES;       // 128 KiB
constexpr int THREADS = 512;

__device__ __forceinline__ int slot_off(int buf, int h) { return (buf * 4 + h) * HALF_BYTES; }
// swizzled 16-byte chunk position inside a 128-byte row (involution; conflict-free ds_read_b128)
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

}  // namespace g8

template <bool F16, int NL, bool SYNC_FIRST = false, bool PEEL = true>
__global__ __launch_bounds__(g8::THREADS, 2) void gemm8_kernel(GemmParams p, SplitArgs sp) {
  using namespace g8;
  typedef TileCfg<256, 256, 64, 2, 4> Cfg;  // epilogue geometry: 4 x 2 blocks of 32x32 per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, g = lane >> 5;

  // ---- workgroup -> (tile, K segment)
  const int ntiles = p.tiles_m * p.tiles_n;
  const int nk_total = p.K / BK;
  int tile_id, seg, unit, kt_begin, kt_end;
  const bool is_split = split_unit_of_block(sp, blockIdx.x, nk_total, &tile_id, &unit, &seg, &kt_begin, &kt_end);
  // tile id -> (m, n): GROUP_M-row supertiles so concurrently running tiles share A / W panels in L2
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = tile_id / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = tile_id - group * per_group;
  const int m0 = (first_m + in_group % gm) * BM;
  const int n0 = (in_group / gm) * BN;
  (void)ntiles;

  // ---- DMA geometry: a half-tile is 2 wave-wide pieces per wave; piece j of wave w fills rows
  //      j*64 + w*8 .. +8 (lane -> row + lane/8, chunk slot lane%8), source chunk pre-swizzled.
  uint32_t src_off[4][2];  // [A0, A1, W0, W1][j] element offsets at k = 0
  {
    const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 64 + wave * 8 + rsub;
      const int ch = swz(row, cpos) * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gm_row = min(m0 + h * HALF_ROWS + row, p.M - 1);
        const int gn_row = min(n0 + h * HALF_ROWS + row, p.N - 1);
        src_off[h][j] = (uint32_t)gm_row * (uint32_t)p.lda + ch;
        src_off[2 + h][j] = (uint32_t)gn_row * (uint32_t)p.ldw + ch;
      }
    }
  }
  // `chk` (std::true_type / false_type): whether the K-tile index still has to be compared with kt_end.  The steady-state
  // iterations (kt + 2 < kt_end) stage unconditionally - no scalar compare + branch around every DMA piece inside the loop.
  auto stage_piece = [&](int kt, int h, int j, auto chk) {  // kt: K-tile, h: 0..3 = A0, A1, W0, W1, j: piece
    if (decltype(chk)::value && kt >= kt_end) return;
    const uint16_t* base = (h < 2 ? p.A : p.W) + (size_t)kt * BK;
    dma16(base + src_off[h][j], smem + slot_off(kt & 1, h) + (j * 64 + wave * 8) * 128);
  };
  auto stage_half = [&](int kt, int h) {
    stage_piece(kt, h, 0, std::true_type{});
    stage_piece(kt, h, 1, std::true_type{});
  };

  // ---- fragment addressing: A rows of this wave live in half-tile `wr`, W rows in half-tile 2 + (wc >> 1)
  const int a_slot = wr;
  const int b_slot = 2 + (wc >> 1);
  const int b_row0 = (wc & 1) * 64;
  u32x4 af[2][4];   // current 64-row A half: [m-block][k-step]
  u32x4 bfr[2][4];  // both 32-column W blocks of the K-tile: [n-block][k-step]
  u32x4 bnext[4];   // W[n0] fragments of the NEXT K-tile (prefetched in phase 4: load segments are 8/4/8/4 reads)
  auto read_a = [&](int buf, int mq) {
    const char* s = smem + slot_off(buf, a_slot);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const int row = mq * 64 + mb * 32 + l31;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[mb][ks] = *(const u32x4*)(s + row * 128 + (swz(row, ks * 2 + g) << 4));
    }
  };
  auto read_w = [&](int buf, int nq, u32x4 (&dst)[4]) {
    const char* s = smem + slot_off(buf, b_slot);
    const int row = b_row0 + nq * 32 + l31;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) dst[ks] = *(const u32x4*)(s + row * 128 + (swz(row, ks * 2 + g) << 4));
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
  // Phase = LDS segment [fragment reads, NL DMA pieces, counted vmcnt] | barrier | lgkmcnt(0) |
  //         MFMA segment [8 MFMA + (2 - NL) DMA pieces between them] | barrier.
  // SYNC_FIRST (A/B config 52, NL = 0): the phase's first barrier comes BEFORE the lgkmcnt(0) that retires the
  // fragment reads, so the LDS latency of the reads overlaps the barrier wait instead of preceding it; measured equal to
  // the default order (scripts/cold_gemm.py).  It stays WAR-safe: a wave retires its reads before its MFMAs,
  // i.e. before the phase's SECOND barrier; the earliest restage of a slot (W0 of the current buffer, staged in phase 3,
  // last read in phase 2) is issued after the stager's first barrier of phase 3, which every wave of its own group
  // passes after that second barrier and every wave of the other group (one barrier behind) passes as its own second
  // barrier of phase 2.  RAW is unchanged: the counted vmcnt still precedes the first barrier of phases 2 / 4 and the data
  // is read a phase later.  With SYNC_FIRST = false the reads are retired before the barrier (the NL >= 1 variants need
  // that: their LDS-segment DMA restages a slot one phase after its last read).
#define G8_PHASE_SYNC()   \
  do {                    \
    if (SYNC_FIRST) {     \
      G8_BARRIER();       \
      G8_LDS_DONE();      \
    } else {              \
      G8_LDS_DONE();      \
      G8_BARRIER();       \
    }                     \
  } while (0)

  // MFMA segment of one phase: 8 MFMA on one 64x32 quadrant (+ the DMA pieces not issued in the LDS segment)
  auto mma_quadrant = [&](int mq, int nq, int st_kt, int st_h, auto chk) {
    __builtin_amdgcn_s_setprio(1);
    int n = 0;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        acc[mq * 2 + mb][nq] = Mfma32<F16>::run(bfr[nq][ks], af[mb][ks], acc[mq * 2 + mb][nq]);
        ++n;
        if ((NL == 0 && n == 2) || (NL <= 1 && n == 5)) {
          G8_FENCE();
          stage_piece(st_kt, st_h, n == 2 ? 0 : 1, chk);
          G8_FENCE();
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };
  auto stage_lds_seg = [&](int st_kt, int st_h, auto chk) {  // the NL pieces issued in the LDS segment (after the reads)
    if (NL >= 1) {
      G8_FENCE();
      stage_piece(st_kt, st_h, NL == 2 ? 0 : 0, chk);
      if (NL == 2) stage_piece(st_kt, st_h, 1, chk);
      G8_FENCE();
    }
  };
  // NL == 1: piece 0 goes in the LDS segment, piece 1 after MFMA #5.  Counted waits: `KEEP` = pieces of the two
  // youngest half-tiles that may stay in flight at the wait point of phases 2 / 4.
  constexpr int KEEP = (NL == 2) ? 4 : (NL == 1 ? 3 : 2);

  // ---- prologue: first K-tile complete + the W halves of the second (what p3/p4 of the previous tile would stage)
  stage_half(kt_begin, 2);
  stage_half(kt_begin, 3);
  stage_half(kt_begin, 0);
  stage_half(kt_begin, 1);
  stage_half(kt_begin + 1, 2);
  stage_half(kt_begin + 1, 3);
  if (kt_begin + 1 < kt_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G8_BARRIER();
  if (wr == 1) G8_BARRIER();  // stagger: group 1 runs one barrier behind group 0
  read_w(kt_begin & 1, 0, bnext);
  G8_LDS_DONE();

#define G8_WAIT_KEEP()                                                          \
  do {                                                                          \
    if (KEEP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");             \
    else if (KEEP == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");        \
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                       \
  } while (0)

  auto k_tile = [&](const int kt, auto chk) {
    constexpr bool CHK = decltype(chk)::value;
    const int buf = kt & 1;
    // ---------------- phase 1: quadrant (m0, n0), stages A0(kt+1)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bfr[0][ks] = bnext[ks];
    read_a(buf, 0);
    stage_lds_seg(kt + 1, 0, chk);
    G8_PHASE_SYNC();
    mma_quadrant(0, 0, kt + 1, 0, chk);
    G8_BARRIER();
    // ---------------- phase 2: quadrant (m0, n1), stages A1(kt+1); retire W0/W1(kt+1): phase 4 prefetches W[n0] of
    //                  K-tile kt+1 from them.  In flight afterwards: A0(kt+1) and the LDS-segment pieces of A1(kt+1).
    read_w(buf, 1, bfr[1]);
    stage_lds_seg(kt + 1, 1, chk);
    if (!CHK || kt + 1 < kt_end) G8_WAIT_KEEP();
    G8_PHASE_SYNC();
    mma_quadrant(0, 1, kt + 1, 1, chk);
    G8_BARRIER();
    // ---------------- phase 3: quadrant (m1, n1), stages W0(kt+2) (the W slots of this buffer are free now)
    read_a(buf, 1);
    stage_lds_seg(kt + 2, 2, chk);
    G8_PHASE_SYNC();
    mma_quadrant(1, 1, kt + 2, 2, chk);
    G8_BARRIER();
    // ---------------- phase 4: quadrant (m1, n0), stages W1(kt+2); retire A0/A1(kt+1).
    if (!CHK || kt + 1 < kt_end) read_w(buf ^ 1, 0, bnext);
    stage_lds_seg(kt + 2, 3, chk);
    if (!CHK || kt + 2 < kt_end) G8_WAIT_KEEP();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G8_PHASE_SYNC();
    mma_quadrant(1, 0, kt + 2, 3, chk);
    G8_BARRIER();
  };
  int kt = kt_begin;
  if constexpr (PEEL)   // (PEEL = false, A/B config 53: every iteration keeps the checks)
    for (; kt + 2 < kt_end; ++kt) k_tile(kt, std::false_type{});   // steady state: tiles kt+1 and kt+2 exist
  for (; kt < kt_end; ++kt) k_tile(kt, std::true_type{});          // last two K-tiles
  if (wr == 0) G8_BARRIER();  // group 0 closes the stagger

  // ---- split-K fix-up: publish the partial tile, last arriver reduces (placement-independent agent-scope
  //      release/acquire; the slab is a per-lane register image, so the reduce is a plain elementwise add)
  if (is_split && !split_k_reduce(acc, sp, unit, seg, tile_id, smem, tid, wave, lane)) return;

  store_tile<F16, Cfg>(p, m0 + wr * 128, n0 + wc * 64, lane, acc);
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
static int g_num_cus[MAX_DEVICES] = {0};
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

extern "C" size_t rtv_gemm_workspace_bytes(void) {
  return (size_t)SPLIT_MAX_UNITS * SPLIT_SLAB_FLOATS * 4 + (size_t)SPLIT_MAX_UNITS * 4 + 256;
}

extern "C" int rtv_gemm_set_workspace(void* ptr, size_t bytes) { return attach_workspace(true, nullptr, ptr, bytes); }

extern "C" int rtv_gemm_set_stream_workspace(rtv_stream_t stream, void* ptr, size_t bytes) {
  return attach_workspace(false, (hipStream_t)stream, ptr, bytes);
}

namespace rtv {

int plan_split_k(int T, int nk, bool allow_split, SplitArgs* sp, int* grid, hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return set_error(-1, "gemm: cannot query the device");
  float* slabs = nullptr;
  int* counters = nullptr;
  int G;
  {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    if (g_num_cus[dev] == 0) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return set_error(-1, "gemm: cannot query the device");
      g_num_cus[dev] = prop.multiProcessorCount;
    }
    G = g_num_cus[dev];
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
  *sp = SplitArgs{T, 1, nullptr, nullptr};
  *grid = T;
  const int R = T % G;
  if (allow_split && slabs && R > 0) {   // T < G (small M under context parallelism): every tile is a split tile
    int S = G / R;                 // the split units of the partial round still fit one round
    if (S > 8) S = 8;
    if (S > nk / 4) S = nk / 4;    // keep >= 4 K-tiles per segment
    if (S >= 2 && (size_t)R * S <= (size_t)SPLIT_MAX_UNITS) {
      sp->first_unit = T - R;
      sp->S = S;
      sp->slabs = slabs;
      sp->counters = counters;
      *grid = (T - R) + R * S;
    }
  }
  return 0;
}

template <bool F16, int NL, bool SYNC_FIRST = false, bool PEEL = true>
static int launch_gemm8_t(GemmParams p, bool allow_split, hipStream_t stream) {
  p.tiles_m = (p.M + g8::BM - 1) / g8::BM;
  p.tiles_n = (p.N + g8::BN - 1) / g8::BN;
  auto kern = gemm8_kernel<F16, NL, SYNC_FIRST, PEEL>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, g8::LDS_BYTES);
    if (e != hipSuccess) return set_error(e, "gemm8: hipFuncSetAttribute");
    attr_set = true;
  }
  SplitArgs sp;
  int grid = 0;
  if (int st = plan_split_k(p.tiles_m * p.tiles_n, p.K / g8::BK, allow_split, &sp, &grid, stream)) return st;
  ProfScope prof(F16 ? PROF_CONV : PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(g8::THREADS), g8::LDS_BYTES, stream, p, sp);
  return check_launch("gemm8");
}

int launch_gemm8(const GemmParams& p, bool f16, int variant, hipStream_t stream) {
  // variant: bit 0 = split-K of the tail round; bits 1.. = DMA pieces issued in the LDS segment (0, 1, 2)
  const bool split = variant & 1;
  const int nl = (variant >> 1) & 3;
  if (variant & 8) return launch_gemm8_t<false, 0, true>(p, split, stream);  // A/B: barrier before the lgkmcnt wait
  if (variant & 16) return launch_gemm8_t<false, 0, false, false>(p, split, stream);  // A/B: bounds checks kept in the loop
  if (f16) return launch_gemm8_t<true, 2>(p, split, stream);
  if (nl == 0) return launch_gemm8_t<false, 0>(p, split, stream);
  if (nl == 1) return launch_gemm8_t<false, 1>(p, split, stream);
  return launch_gemm8_t<false, 2>(p, split, stream);
}

}  // namespace rtv
