// 256x256x64 projection GEMM, ONE WAVE PER SIMD (tile config 8, opt-in: NOT the production kernel): four waves, each owning a
// 128 x 128 block of the tile (16 accumulator blocks = 256 registers); same math, fused epilogue and output as gemm8.hip
// (bit-identical results, tests/test_kernels_gpu.py).  Measured at 0.92-0.95x of gemm8 on the 14B projection shapes
// (profiles/r03_gemm4_one_wave_per_simd_v2.log has every step below with its numbers); kept as the record of that experiment
// and as the starting point for a deeper operand pipeline.
//
// Why this shape: the matrix kernels run against the power cap (1.65-1.70 GHz, profiles/r03_pmc_hot_kernels.txt) and gemm8's LDS
// port is as busy as its matrix pipes (8 waves x 24 KiB of fragment reads + 64 KiB of DMA writes per K-tile = 2048 cycles at
// 128 B/clk, the MFMA time of a K-tile).  A 128 x 128 register tile reads each staged operand byte twice instead of three times -
// the shape of hipBLASLt's kernel for these problems (MT256x256x64, 256 threads: profiles/r03_hipblaslt_kernels.txt).
//
// What it took (profiles/r03_gemm4_*.log):
//   1. With one wave per SIMD nothing covers a wave's own issue bubbles, so every non-matrix instruction sits in the shadow of an
//      MFMA: a k-step (16 MFMAs on one of two fragment register sets) carries the eight fragment reads of the NEXT k-step (inline
//      asm, into the other set; the consumer waits with counted lgkmcnt right in front of the first MFMA that needs a fragment)
//      and the staging instructions, at most one LDS and one memory operation per MFMA slot.  (Reads in a clump in front of each
//      k-step: 0.70-0.83x of gemm8; interleaved: 0.93x.)
//   2. The kernel was then bound by the LATENCY of its operand stream, not by issue: with the DMA pieces never waited for it ran
//      at 1484 TF/s, with every piece re-reading K-tile 0 (L2 hits) at 1603, with real data at 1230 (4680 x 15360 x 5120).  In
//      160 KiB (A x 3 buffers, W x 2, gemm8's layout) W(t+2) can only be fetched once W(t) has been read: its second half had
//      three k-steps (~0.85 us) to arrive.  A ring of 32-wide K slices gave every piece 2.5 slices but its 64-byte row pieces
//      (16 half lines per DMA instruction) ran at 0.85x even on L2 hits (profiles/r03_gemm4_ring_bk32.log) - full 128-byte lines
//      per 8 lanes it is.  So: A stays on the LDS DMA, three buffers, issued two tiles ahead (its buffer is free that early);
//      W goes global -> REGISTERS (two sets of 8 x 16 bytes per lane, loaded 1.5 tiles before they are written) -> ds_write_b128
//      into the W buffer the moment the K-tile's barrier frees it.  Every operand byte now has >= 1.5 K-tiles (~1.7 us) to arrive
//      and the kernel no longer waits for memory (dropping the counted wait changes nothing) - but the register -> LDS writes and
//      the extra loads cost 30 + 35 us of the 620 (lab variants 9-11), more than the waiting did.
//
// K-tile u, per wave (fragment reads of the next k-step behind the even MFMAs of k-steps 0-2 / behind MFMAs 4-8, 10, 12, 14 of
// k-step 3):
//   k-step 0: A(u+2) rows 128-255 by DMA (4 pieces), W(u+1) registers 0-3 -> LDS
//   k-step 1: W(u+1) registers 4-7 -> LDS, W(u+3) -> registers 0-3 (the set W(u+1) has just left)
//   k-step 2: W(u+3) -> registers 4-7
//   k-step 3: behind MFMA 3 the K-tile's ONE counted wait + barrier: vmcnt(16) retires A(u+1) and the W(u+2) registers (the 8
//             A(u+2) pieces and the 8 W(u+3) loads stay in flight); everybody's A(u+1) and W(u+1) are in LDS, every wave holds
//             its last fragments of A(u) / W(u); then the reads of (u+1, k-step 0) and A(u+3) rows 0-127 into the buffer of A(u).
// The register sets alternate with the parity of u: the tile body is instantiated per parity so that a set is a fixed group of
// registers (data in flight into a register must never be moved by the compiler).
#ifdef RTV_LAB   // experimental (5-7 % slower than gemm8): compiled only into the lab build, see include/rtv_hip_lab.h
#include <type_traits>

#include "gemm_core.h"
#include "gemm_split.h"
#include "rtv_internal.h"

namespace rtv {

namespace g4 {
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;     // 16 KiB: 128 rows of one operand
constexpr int A_OFF = 0;                     // 3 K-tile buffers x 2 halves
constexpr int W_OFF = 6 * HALF_BYTES;        // 2 K-tile buffers x 2 halves
constexpr int LDS_BYTES = 10 * HALF_BYTES;   // 160 KiB
constexpr int THREADS = 256;
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
template <int V>
struct IC {
  static constexpr int value = V;
};
template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(IC<I>{});
    sfor<I + 1, N>(f);
  }
}
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(uint32_t lds_addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
  return r;
}
template <int OFF>
__device__ __forceinline__ void lds_write128(uint32_t lds_addr, const u32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lds_addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ u32x4 global_read128(const u32x4& rsrc, uint32_t voff, uint32_t soff) {
  u32x4 r;
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(r) : "v"(voff), "s"(rsrc), "s"(soff));
  return r;
}
// counted waits the named fragment registers depend on (no consumer can be scheduled above them)
template <int CNT>
__device__ __forceinline__ void lds_wait1(u32x4& a) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT));
}
template <int CNT>
__device__ __forceinline__ void lds_wait5(u32x4 (&b)[4], u32x4& a) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(a) : "n"(CNT));
}
// ---- the issue schedule of a k-step: which MFMA slot (the instructions behind MFMA n) carries what
// fragment read i of the NEXT k-step (0-3 = W blocks, 4-7 = A blocks, the order the consumer needs them in); k-step 3 starts
// them behind the barrier, the five the next k-step needs first back to back
__device__ constexpr int rd_at(int ks, int i) { return ks == 3 ? (i < 5 ? 4 + i : 2 * i) : 2 * i; }
// register -> LDS write w (0-3) of k-steps 0 / 1: slots 1, 5, 9, 13
__device__ constexpr bool has_writes(int ks) { return ks < 2; }
__device__ constexpr int wr_at(int w) { return 1 + 4 * w; }
// LDS operations (reads of the next batch + writes) a k-step issues in slots < n
__device__ constexpr int lds_before(int ks, int n) {
  int c = 0;
  for (int i = 0; i < 8; ++i) c += rd_at(ks, i) < n ? 1 : 0;
  if (has_writes(ks))
    for (int w = 0; w < 4; ++w) c += wr_at(w) < n ? 1 : 0;
  return c;
}
// ... and in slots behind the one of its read i
__device__ constexpr int lds_after_read(int ks, int i) { return lds_before(ks, 16) - lds_before(ks, rd_at(ks, i) + 1); }
}  // namespace g4

// LAB (timing experiments, garbage results): 6 = the counted vmcnt wait of the K-tile dropped, 7 = every piece / load re-reads
// K-tile 0 (L2 hits), 9 = no register -> LDS writes in the steady state, 10 = no W loads, 11 = neither.
template <bool F16, int LAB>
__global__ __launch_bounds__(g4::THREADS, 1) void gemm4_kernel(GemmParams p) {
  using namespace g4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, g = lane >> 5;

  const int nk = p.K / BK;
  // tile id -> (m, n): GROUP_M-row supertiles, tiles of one round share A / W panels in L2 (as gemm8)
  const int tile_id = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = tile_id / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = tile_id - group * per_group;
  const int m0 = (first_m + in_group % gm) * BM;
  const int n0 = (in_group / gm) * BN;

  // ---- staging geometry: a half-tile (128 rows x 128 bytes) is 16 pieces of 8 rows; wave w moves pieces 4 j + w, j = 0..3.
  //      Lane -> row + lane / 8, LDS chunk slot lane % 8; the lane fetches the chunk that belongs in that slot (the swizzle is
  //      applied on the source side), so the LDS image of a piece is 1 KiB in lane order for the DMA and the ds_write alike.
  uint32_t src_off[4][4];  // [A0, A1, W0, W1][j]: byte offsets at k = 0
  {
    const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = (j * 4 + wave) * 8 + rsub;
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
  __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  u32x4 descW;   // the same descriptor as four scalars, for the inline-asm loads
  {
    const uintptr_t wp = (uintptr_t)p.W;
    descW[0] = __builtin_amdgcn_readfirstlane((uint32_t)wp);
    descW[1] = __builtin_amdgcn_readfirstlane((uint32_t)(wp >> 32) & 0xffffu);
    descW[2] = 0x7fffffffu;
    descW[3] = 0x00020000u;
  }
  // `chk` (std::true_type / false_type): whether the K-tile index still has to be compared with nk - the steady-state iterations
  // stage unconditionally (a scalar compare + branch around every piece costs the hot loop several per cent, see gemm8.hip)
  auto dma_a = [&](int kt, int a3, int h, int j, auto chk) {   // A half h of K-tile kt -> A buffer a3
    if (decltype(chk)::value && kt >= nk) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (RTV_LDS void*)(smem + A_OFF + (a3 * 2 + h) * HALF_BYTES + (j * 4 + wave) * 1024),
                                             16, src_off[h][j], LAB == 7 ? 0u : (unsigned)kt * (BK * 2), 0, 0);
  };
  auto dma_w = [&](int kt, int h, int j) {                     // prologue only: W(0) straight into its buffer
    if (kt >= nk) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (RTV_LDS void*)(smem + W_OFF + ((kt & 1) * 2 + h) * HALF_BYTES + (j * 4 + wave) * 1024),
                                             16, src_off[2 + h][j], (unsigned)kt * (BK * 2), 0, 0);
  };

  // ---- fragment read addresses: A rows of this wave = half wr, W rows = half wc; row = block * 32 + l31, so the swizzle key
  //      (row >> 1) & 7 does not depend on the block: one base per k-step, blocks are immediate offsets of 4096 bytes
  const uint32_t lds0 = (uint32_t)(uintptr_t)(RTV_LDS const char*)smem;
  uint32_t a_base[4], w_base[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    a_base[ks] = lds0 + A_OFF + wr * HALF_BYTES + l31 * 128 + (swz(l31, ks * 2 + g) << 4);
    w_base[ks] = lds0 + W_OFF + wc * HALF_BYTES + l31 * 128 + (swz(l31, ks * 2 + g) << 4);
  }
  const uint32_t w_store = lds0 + W_OFF + wave * 1024 + lane * 16;   // + buffer * 32 KiB + (h * 4 + j) * 4 KiB

  f32x16 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  u32x4 wq0[8], wq1[8];   // W in flight: [h * 4 + j]; set 0 holds even K-tiles, set 1 odd ones
  auto load_w = [&](auto pc, auto rc, int kt, auto chk) __attribute__((always_inline)) {   // W(kt) register r <- global
    constexpr int P = decltype(pc)::value, r = decltype(rc)::value;
    if (decltype(chk)::value && kt >= nk) return;
    const u32x4 v = global_read128(descW, src_off[2 + (r >> 2)][r & 3], LAB == 7 ? 0u : (unsigned)kt * (BK * 2));
    if constexpr (P == 0) wq0[r] = v;
    else wq1[r] = v;
  };
  auto store_w = [&](auto pc, auto rc, uint32_t base) __attribute__((always_inline)) {    // register r of set P -> LDS
    constexpr int P = decltype(pc)::value, r = decltype(rc)::value;
    if constexpr (P == 0) lds_write128<r * 4096>(base, wq0[r]);
    else lds_write128<r * 4096>(base, wq1[r]);
  };

#define G4_FENCE() __builtin_amdgcn_sched_barrier(0)
  // ---- prologue, in the order the counted waits assume: A(0), W(0) by DMA; W(1) -> set 1; A(1); W(2) -> set 0; A(2) rows 0-127
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a(0, 0, h, j, std::true_type{});
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_w(0, h, j);
  sfor<0, 8>([&](auto rc) { load_w(IC<1>{}, rc, 1, std::true_type{}); });
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 4; ++j) dma_a(1, 1, h, j, std::true_type{});
  sfor<0, 8>([&](auto rc) { load_w(IC<0>{}, rc, 2, std::true_type{}); });
#pragma unroll
  for (int j = 0; j < 4; ++j) dma_a(2, 2, 0, j, std::true_type{});
  // A(0), W(0) and the W(1) registers: everything but A(1), W(2), A(2) rows 0-127
  if (nk >= 3) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (nk == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  G4_FENCE();
  __builtin_amdgcn_s_barrier();
  G4_FENCE();

  u32x4 af[2][4], bf[2][4];   // [set][block] fragments of one k-step
  auto read_frag = [&](int set, uint32_t aa, uint32_t ww, auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < 4) bf[set][i] = lds_read128<i * 4096>(ww);
    else af[set][i - 4] = lds_read128<(i - 4) * 4096>(aa);
  };
  sfor<0, 8>([&](auto ic) { read_frag(0, a_base[0], w_base[0], ic); });
  int a3 = 0;   // A buffer of the K-tile being multiplied
  auto k_tile = [&](const int kt, auto pc, auto chk) {
    constexpr int P = decltype(pc)::value;          // kt & 1 = W buffer of this K-tile; the set that holds W(kt+2)
    constexpr bool CHK = decltype(chk)::value;
    const int a_off = a3 * (2 * HALF_BYTES);
    const int a3n = a3 == 2 ? 0 : a3 + 1;           // buffer of A(kt+1); A(kt+2) lives in the third one
    const int a3nn = a3n == 2 ? 0 : a3n + 1;
    constexpr int w_off = P * (2 * HALF_BYTES), w_offn = (P ^ 1) * (2 * HALF_BYTES);
    const bool w_next = !CHK || kt + 1 < nk;        // W(kt+1) exists: registers of set P ^ 1 -> W buffer P ^ 1
    sfor<0, 4>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      constexpr int set = ks & 1;
      constexpr int pks = (ks + 3) & 3;            // the k-step that issued this one's fragment reads
      constexpr bool LAST = ks == 3;
      // fragments of this k-step: W blocks + A block 0 now, A block mb in front of MFMA 4 mb.  LDS operations retire in order:
      // the count allows everything issued behind the needed read (steady state; the tail tiles, where some of those
      // operations are skipped, allow only the reads of the same batch)
      lds_wait5<CHK ? 3 : lds_after_read(pks, 4)>(bf[set], af[set][0]);
      G4_FENCE();
      const uint32_t aa = a_base[(ks + 1) & 3] + (LAST ? a3n * (2 * HALF_BYTES) : a_off);
      const uint32_t ww = w_base[(ks + 1) & 3] + (LAST ? w_offn : w_off);
      const bool more = !LAST || !CHK || kt + 1 < nk;
      sfor<0, 16>([&](auto nc) {
        constexpr int n = decltype(nc)::value, mb = n >> 2, nb = n & 3;
        if constexpr (n > 0 && nb == 0) {
          lds_wait1<CHK ? 3 - mb : lds_after_read(pks, 4 + mb) + lds_before(ks, n)>(af[set][mb]);
          G4_FENCE();
        }
        acc[mb][nb] = Mfma32<F16>::run(bf[set][nb], af[set][mb], acc[mb][nb]);
        G4_FENCE();
        if constexpr (LAST && n == 3) {
          // A(kt+1) and the W(kt+2) registers; younger: A(kt+2) (8 pieces), W(kt+3) (8 loads)
          if (!CHK) {
            if constexpr (LAB == 10 || LAB == 11) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (LAB != 6) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          } else if (kt + 3 < nk) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
          } else if (kt + 2 < nk) {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          G4_FENCE();
          __builtin_amdgcn_s_barrier();
          G4_FENCE();
        }
        // ---- LDS slot: a fragment read of the next k-step or a W register on its way into LDS
        sfor<0, 8>([&](auto ic) {
          if constexpr (rd_at(ks, decltype(ic)::value) == n) {
            if (more) read_frag(set ^ 1, aa, ww, ic);
          }
        });
        if constexpr (has_writes(ks) && (n & 3) == 1) {
          if (w_next && (CHK || (LAB != 9 && LAB != 11))) store_w(IC<P ^ 1>{}, IC<ks * 4 + (n >> 2)>{}, w_store + w_offn);
        }
        // ---- memory slot
        if constexpr (ks == 0 && (n & 3) == 3) dma_a(kt + 2, a3nn, 1, n >> 2, chk);
        if constexpr (ks == 1 && (n & 3) == 3 && (CHK || (LAB != 10 && LAB != 11))) load_w(IC<P ^ 1>{}, IC<(n >> 2)>{}, kt + 3, chk);
        if constexpr (ks == 2 && (n & 3) == 1 && (CHK || (LAB != 10 && LAB != 11))) load_w(IC<P ^ 1>{}, IC<4 + (n >> 2)>{}, kt + 3, chk);
        if constexpr (LAST && n >= 9 && (n & 1)) dma_a(kt + 3, a3, 0, (n - 9) >> 1, chk);
        G4_FENCE();
      });
    });
    a3 = a3n;
  };
  int kt = 0;
  for (; kt + 4 < nk; kt += 2) {   // steady state: K-tiles kt + 3, kt + 4 exist
    k_tile(kt, IC<0>{}, std::false_type{});
    k_tile(kt + 1, IC<1>{}, std::false_type{});
  }
  for (; kt < nk; kt += 2) {       // the last three or four K-tiles (nk is even: K % 128 == 0, checked by the launcher)
    k_tile(kt, IC<0>{}, std::true_type{});
    k_tile(kt + 1, IC<1>{}, std::true_type{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  G4_FENCE();
  __builtin_amdgcn_s_barrier();   // every fragment read and staging write is done: LDS becomes the epilogue image
  G4_FENCE();
#undef G4_FENCE

  // ---- epilogue: the 128 x 128 wave block as two 128 x 64 column halves through a wave-private 16 KiB image each pass
  const bool wide = !((p.ldc | (p.residual ? p.ldr : 0)) & 7) && !(((uintptr_t)p.C | (uintptr_t)p.residual) & 15);
  sfor<0, 2>([&](auto hc) {
    constexpr int hn = decltype(hc)::value;
    f32x16 part[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      part[mi][0] = acc[mi][hn * 2];
      part[mi][1] = acc[mi][hn * 2 + 1];
    }
    if (wide) {
      store_tile_lds<F16, 4>(p, m0 + wr * 128, n0 + wc * 128 + hn * 64, lane, smem + wave * (128 * 128), part);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is rewritten by the second half
    } else {
      typedef TileCfg<256, 256, 64, 2, 4> Cfg;   // 4 x 2 blocks per call
      store_tile<F16, Cfg>(p, m0 + wr * 128, n0 + wc * 128 + hn * 64, lane, part);
    }
  });
}

template <bool F16, int LAB>
static int launch_gemm4_t(GemmParams p, hipStream_t stream) {
  p.tiles_m = (p.M + g4::BM - 1) / g4::BM;
  p.tiles_n = (p.N + g4::BN - 1) / g4::BN;
  auto kern = gemm4_kernel<F16, LAB>;
  static LdsAttr lds_attr;   // per device (a second GPU used from this process needs the attribute as well)
  if (int st = ensure_dynamic_lds((const void*)kern, g4::LDS_BYTES, &lds_attr, "gemm4")) return st;
  ProfScope prof(F16 ? PROF_CONV : PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(g4::THREADS), g4::LDS_BYTES, stream, p);
  return check_launch("gemm4");
}

int launch_gemm4(const GemmParams& p, bool f16, hipStream_t stream, int lab) {
  if ((size_t)p.M * p.lda * 2 > 0x7fffffffull || (size_t)p.N * p.ldw * 2 > 0x7fffffffull)
    return set_error(-1, "gemm4: operand larger than 2 GiB");
  if (p.K % 128) return set_error(-1, "gemm4: K must be a multiple of 128 (the K loop runs in pairs of 64-wide tiles)");
  if (f16) return launch_gemm4_t<true, 0>(p, stream);
  switch (lab) {
    case 6: return launch_gemm4_t<false, 6>(p, stream);
    case 7: return launch_gemm4_t<false, 7>(p, stream);
    case 9: return launch_gemm4_t<false, 9>(p, stream);    // no register -> LDS writes in the steady state
    case 10: return launch_gemm4_t<false, 10>(p, stream);  // no W loads
    case 11: return launch_gemm4_t<false, 11>(p, stream);  // neither
    default: return launch_gemm4_t<false, 0>(p, stream);
  }
}

}  // namespace rtv
#endif  // RTV_LAB
