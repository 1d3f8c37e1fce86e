// 160x256x64 projection GEMM, one wave per SIMD: the kernel for FEW-ROW problems - the token shards of context parallelism
// (xdit_context_parallel.py:131-142 in the reference: 4680 / 8 = 585 rows per rank) - whose tile counts the 256- and 128-row
// ping-pong kernels (gemm8.hip) quantise badly: 585 rows are 3 tiles of 256 (the third 71 % empty: 180 tiles on 256 CUs, all
// of them paying a full tile time) or 5 tiles of 128 (300 tiles = 1.17 rounds); 4 tiles of 160 rows make ONE round of EQUAL
// units that start together - 240 tiles for the QKV projection, 216 for ffn-in, 80 x 3 K segments = 240 for the N = 5120
// shapes (a tile height that keeps the tiles of an XCD in lockstep, which a stream-K schedule gives up).
// Same math, same K order and the same fused epilogue as gemm8 (bit-identical without split-K), same C ABI (tile config 9;
// default dispatch: launch_gemm, gemm.hip).
//
// Structure = the one-wave-per-SIMD idiom of attn_w4.hip: FOUR waves, wave w owns all 160 rows x columns [64 w, 64 w + 64) of
// the tile = 5 x 2 accumulator blocks; 7 fragment reads (5 activation + 2 weight blocks) feed 10 MFMAs per 16-deep k-step.
// Accumulators a[0:159] and two fragment register sets a[160:187] / a[188:215] are named LITERALLY in inline asm (the compiler
// sees none of these values; the K loop has no VALU instruction at all), fragment set s + 1 is read behind the MFMAs of k-step
// s.  A K-tile is 20 KiB of A + 32 KiB of W = 52 one-KiB LDS-DMA pieces, 13 per wave, issued behind MFMAs of k-steps 0-2 two
// K-tiles ahead into a 3-slot ring (156 KiB); ONE counted vmcnt + ONE barrier per K-tile (behind k-step 2; k-step 3 then reads
// the first fragments of the next K-tile).  Pieces of an operand that share an M0 value step through the LDS by the
// instruction's immediate offset (applied to both addresses; the per-lane memory offsets carry the compensation).
// Split-K (tile counts below the CU count, tail rounds) and the epilogue through LDS are gemm8's (gemm_split.h, gemm_core.h).
#include "gemm_core.h"
#include "gemm_split.h"
#include "rtv_internal.h"

namespace rtv {

namespace g5 {
constexpr int BM = 160, BN = 256, BK = 64;
constexpr int A_BYTES = BM * BK * 2;          // 20 KiB
constexpr int W_BYTES = BN * BK * 2;          // 32 KiB
constexpr int SLOT_BYTES = A_BYTES + W_BYTES; // 52 KiB: [A | W] of one K-tile
constexpr int NSLOT = 3;
constexpr int LDS_BYTES = NSLOT * SLOT_BYTES; // 156 KiB (the epilogue image - 4 x 20 KiB - reuses it)
constexpr int THREADS = 256;
constexpr int A_PIECES = 5, W_PIECES = 8;     // per wave and K-tile (8 rows of 128 bytes each)
constexpr int PIECES = A_PIECES + W_PIECES;
// swizzled 16-byte chunk position inside a 128-byte row (involution; conflict-free ds_read_b128) - gemm8's
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// accumulation-register map (asm-owned, see attn_w4.hip): accumulator block (mi, ni) a[(mi*2 + ni)*16 .. +15];
// fragment set s: weight blocks ni at a[160 + 28 s + 4 ni .. +3], activation blocks mi at a[168 + 28 s + 4 mi .. +3]
constexpr int A_ACC = 0, A_FRAG = 160, FRAG_SET = 28;
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
// acc(AGPR) += weight fragment (AGPR) . activation fragment (AGPR)   (swapped operands: a lane owns one output row, gemm_core.h)
template <bool F16, int ACC, int WF, int AF>
__device__ __forceinline__ void mfma_aaa() {
  if constexpr (F16)
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], a[%c2:%c3], a[%c4:%c5], a[%c0:%c1]" ::"n"(ACC), "n"(ACC + 15), "n"(WF), "n"(WF + 3),
                 "n"(AF), "n"(AF + 3));
  else
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c0:%c1], a[%c2:%c3], a[%c4:%c5], a[%c0:%c1]" ::"n"(ACC), "n"(ACC + 15), "n"(WF), "n"(WF + 3),
                 "n"(AF), "n"(AF + 3));
}
template <int DST, int OFF>
__device__ __forceinline__ void lds_read128_a(uint32_t lds_addr) {
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%3" ::"v"(lds_addr), "n"(DST), "n"(DST + 3), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(%c0)" ::"n"(CNT));
}
template <int DST>
__device__ __forceinline__ void acc_zero() {
  asm volatile("v_accvgpr_write_b32 a[%c0], 0" ::"n"(DST));
}
template <int SRC>
__device__ __forceinline__ float acc_read() {
  float r;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(r) : "n"(SRC));
  return r;
}
__device__ __forceinline__ void acc_settle() { asm volatile("s_nop 15\n\ts_nop 7"); }
// (the one clobber list of the kernel: makes the kernel descriptor allocate the accumulation registers it names literally)
#define RTV_G5_ACC \
  "a0", "a15", "a31", "a47", "a63", "a79", "a95", "a111", "a127", "a143", "a159", "a175", "a191", "a207", "a215"
}  // namespace g5

// K-loop issue schedule of a K-tile (per wave): k-step s = 10 MFMAs on fragment set s & 1; behind MFMA n of k-step s:
//   n = 0..6          fragment read n of the NEXT k-step (0, 1: weight blocks; 2..6: activation blocks) into the other set
//   s <= 2, n = 7, 8, 9 (+ n = 5, 6 in k-steps 0, 1)   the K-tile's 13 DMA pieces (two K-tiles ahead): 5 + 5 + 3
// behind k-step 2: vmcnt(13) (the next K-tile has landed for this wave) + s_barrier (... for every wave, and everybody has
// issued its last reads of this K-tile's predecessor, whose slot the pieces of k-steps 0-2 are filling).
template <bool F16>
__global__ __launch_bounds__(g5::THREADS, 1) void gemm5_kernel(GemmParams p, SplitArgs sp) {
  using namespace g5;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;

  // ---- workgroup -> (tile, K segment); tile id -> (m, n): GROUP_M-row supertiles (the row tiles of a column share its W panel)
  const int nk_total = p.K / BK;
  int tile_id, seg, unit, kt_begin, kt_end;
  const bool is_split = split_unit_of_block(sp, blockIdx.x, nk_total, &tile_id, &unit, &seg, &kt_begin, &kt_end);
  constexpr int GROUP_M = 8;
  const int per_group = GROUP_M * p.tiles_n;
  const int group = tile_id / per_group;
  const int first_m = group * GROUP_M;
  const int gm = min(p.tiles_m - first_m, GROUP_M);
  const int in_group = tile_id - group * per_group;
  const int m0 = (first_m + in_group % gm) * BM;
  const int n0 = (in_group / gm) * BN;
  kt_begin = __builtin_amdgcn_readfirstlane(kt_begin);
  kt_end = __builtin_amdgcn_readfirstlane(kt_end);

  // ---- DMA geometry.  A piece = 8 rows x 128 bytes: lane -> row + lane / 8, chunk slot lane % 8, source chunk pre-swizzled.
  //      Wave w stages A pieces 5 w .. 5 w + 4 (rows 40 w ..) and W pieces 8 w .. 8 w + 7 (rows 64 w ..).  Pieces i, i + 1, ..
  //      of a group share one M0 value and step by the immediate offset 1024 (i - i0): groups A {0..3}, {4}, W {0..3}, {4..7}.
  //      BYTE offsets at k = 0, minus the immediate.
  uint32_t a_off[A_PIECES], w_off[W_PIECES];
  {
    const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int row = (wave * A_PIECES + i) * 8 + rsub;
      const int gm_row = min(m0 + row, p.M - 1);
      a_off[i] = ((uint32_t)gm_row * (uint32_t)p.lda + swz(row, cpos) * 8) * 2u - 1024u * (i & 3);
    }
#pragma unroll
    for (int i = 0; i < W_PIECES; ++i) {
      const int row = (wave * W_PIECES + i) * 8 + rsub;
      const int gn_row = min(n0 + row, p.N - 1);
      w_off[i] = ((uint32_t)gn_row * (uint32_t)p.ldw + swz(row, cpos) * 8) * 2u - 1024u * (i & 3);
    }
  }
  // (rows are clamped above, K is a multiple of 64: no bounds needed.  Piece i starts at tile row >= 8 i, a row is >= 128 bytes: the
  //  compensated offset of a real row is >= 0; a row clamped to M - 1 < 8 i may wrap - it then reads zeros and its output is discarded)
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0x7fffffff, 0x00020000);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(RTV_LDS char*)smem;
  const uint32_t lds_a = lds0 + (uint32_t)wave * (A_PIECES * 1024), lds_w = lds0 + A_BYTES + (uint32_t)wave * (W_PIECES * 1024);
  // piece I (0..4: A, 5..12: W) of K-tile kt (clamped: tiles past the segment re-stage its last one into a slot nobody reads)
  auto piece = [&](auto ic, uint32_t slot_off, uint32_t koff) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    if constexpr (I < A_PIECES) {
      constexpr int IMM = (I & 3) * 1024;
      RTV_LDS void* dst = (RTV_LDS void*)(uintptr_t)(lds_a + slot_off + (uint32_t)(I * 1024 - IMM));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, dst, 16, a_off[I], koff, IMM, 0);
    } else {
      constexpr int J = I - A_PIECES, IMM = (J & 3) * 1024;
      RTV_LDS void* dst = (RTV_LDS void*)(uintptr_t)(lds_w + slot_off + (uint32_t)(J * 1024 - IMM));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, dst, 16, w_off[J], koff, IMM, 0);
    }
  };

  // ---- fragment read addresses (bytes inside a slot): row l31 of a 32-row block, chunk (2 ks + g) ^ swizzle(row); the block
  //      index is an immediate (32 rows = 4096 bytes: the swizzle term repeats every 16 rows)
  //      (an LDS immediate is 16 bits: the slot base - 0 / 52 / 104 KiB - is part of the address register: 24 of them)
  uint32_t fa[NSLOT][4], fw[NSLOT][4];
#pragma unroll
  for (int sl = 0; sl < NSLOT; ++sl)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      fa[sl][ks] = lds0 + (uint32_t)(sl * SLOT_BYTES + l31 * 128 + (swz(l31, ks * 2 + g) << 4));
      fw[sl][ks] = lds0 + (uint32_t)(sl * SLOT_BYTES + A_BYTES + (wave * 64 + l31) * 128 + (swz(l31, ks * 2 + g) << 4));
    }
  // the 7 reads of k-step KS of the K-tile in slot SLOT into fragment set SET; read N (0, 1 weight; 2..6 activation)
  auto frag_read = [&](auto slotc, auto ksc, auto setc, auto nc) __attribute__((always_inline)) {
    constexpr int SLOT = decltype(slotc)::value, KS = decltype(ksc)::value, SET = decltype(setc)::value, N = decltype(nc)::value;
    if constexpr (N < 2) lds_read128_a<A_FRAG + FRAG_SET * SET + 4 * N, N * 4096>(fw[SLOT][KS]);
    else lds_read128_a<A_FRAG + FRAG_SET * SET + 8 + 4 * (N - 2), (N - 2) * 4096>(fa[SLOT][KS]);
  };

  asm volatile("" ::: RTV_G5_ACC);
  sfor<0, 160>([&](auto ic) __attribute__((always_inline)) { acc_zero<A_ACC + decltype(ic)::value>(); });

  // ---- prologue: K-tiles 0 and 1 of the segment, then the first fragment set
  const int kt_last = kt_end - 1;
  auto koff_of = [&](int kt) __attribute__((always_inline)) { return (uint32_t)min(kt, kt_last) * (uint32_t)(BK * 2); };
  sfor<0, PIECES>([&](auto ic) __attribute__((always_inline)) { piece(ic, 0u, koff_of(kt_begin)); });
  sfor<0, PIECES>([&](auto ic) __attribute__((always_inline)) { piece(ic, (uint32_t)SLOT_BYTES, koff_of(kt_begin + 1)); });
  asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  sfor<0, 7>([&](auto nc) __attribute__((always_inline)) { frag_read(IC<0>{}, IC<0>{}, IC<0>{}, nc); });

  // one K-tile: the tile in ring slot C (compile time: the loop is unrolled by the ring length); kt = its index
  auto k_tile = [&](auto cc, int kt) __attribute__((always_inline)) {
    constexpr int C = decltype(cc)::value, CN = (C + 1) % NSLOT, C2 = (C + 2) % NSLOT;
    const uint32_t koff = koff_of(kt + 2);
    uint32_t lo = 0u;
    asm volatile("" : "+s"(lo));   // (keeps the 13 destination addresses of a slot an s_add each instead of hoisted SGPRs)
    sfor<0, 4>([&](auto ksc) __attribute__((always_inline)) {
      constexpr int KS = decltype(ksc)::value, SET = KS & 1;
      lds_wait<0>();   // the 7 reads of this k-step (issued behind the first MFMAs of the previous one)
      sfor<0, 10>([&](auto nc) __attribute__((always_inline)) {
        constexpr int n = decltype(nc)::value, mi = n >> 1, ni = n & 1;
        mfma_aaa<F16, A_ACC + (mi * 2 + ni) * 16, A_FRAG + FRAG_SET * SET + 4 * ni, A_FRAG + FRAG_SET * SET + 8 + 4 * mi>();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n < 7) {   // fragment read n of the next k-step: k-step KS + 1 of this tile, or k-step 0 of the next one
          if constexpr (KS < 3) frag_read(IC<C>{}, IC<KS + 1>{}, IC<SET ^ 1>{}, nc);
          else frag_read(IC<CN>{}, IC<0>{}, IC<SET ^ 1>{}, nc);
        }
        // DMA pieces of K-tile kt + 2 (slot C2): k-step 0 pieces 0-4 behind MFMAs 5..9, k-step 1 pieces 5-9, k-step 2 pieces 10-12
        if constexpr (KS < 2 && n >= 5) piece(IC<KS * 5 + n - 5>{}, lo + (uint32_t)(C2 * SLOT_BYTES), koff);
        if constexpr (KS == 2 && n >= 7) piece(IC<10 + n - 7>{}, lo + (uint32_t)(C2 * SLOT_BYTES), koff);
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (KS == 2) {
        asm volatile("s_waitcnt vmcnt(13)" ::: "memory");   // everything but this K-tile's 13 pieces: tile kt + 1 has landed
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    });
  };
  for (int kt = kt_begin; kt < kt_end; kt += 3) {
    k_tile(IC<0>{}, kt);
    if (kt + 1 >= kt_end) break;
    k_tile(IC<1>{}, kt + 1);
    if (kt + 2 >= kt_end) break;
    k_tile(IC<2>{}, kt + 2);
  }
  lds_wait<0>();                                      // the surplus fragment reads of the last k-step
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ... and DMA pieces: the LDS becomes the epilogue image
  __builtin_amdgcn_s_barrier();

  // ---- accumulators -> architectural registers; split-K fix-up; fused epilogue through LDS (gemm_core.h)
  acc_settle();
  f32x16 acc[5][2];
  sfor<0, 160>([&](auto ic) __attribute__((always_inline)) {
    constexpr int i = decltype(ic)::value;
    acc[i >> 5][(i >> 4) & 1][i & 15] = acc_read<A_ACC + i>();
  });
  if (is_split && !split_k_reduce<5>(acc, sp, unit, seg, tile_id, smem, tid, wave, lane)) return;
  if (is_split) __syncthreads();   // (the reducer's flag word at smem[0] is dead; the image below overwrites it)
  store_tile_lds<F16, 5>(p, m0, n0 + wave * 64, lane, smem + wave * (5 * 32 * 128), acc);
}

// tile config 9.  `split_k`: tile counts that leave CUs idle are cut along K (plan_split_k, gemm8.hip)
int launch_gemm5(const GemmParams& p_, bool f16, bool split_k, hipStream_t stream) {
  GemmParams p = p_;
  if (p.K % g5::BK || p.K < 2 * g5::BK) return set_error(-1, "gemm5: K must be a multiple of 64, >= 128");
  if (p.lda < 64 || p.ldw < 64) return set_error(-1, "gemm5: operand rows must be >= 128 bytes");
  if (p.N % 8) return set_error(-1, "gemm5: N must be a multiple of 8");
  p.tiles_m = (p.M + g5::BM - 1) / g5::BM;
  p.tiles_n = (p.N + g5::BN - 1) / g5::BN;
  SplitArgs sp;
  int grid = 0;
  if (int st = plan_split_k(p.tiles_m * p.tiles_n, p.K / g5::BK, split_k, &sp, &grid, stream)) return st;
  static LdsAttr lds_attr[2];
  const void* kern = f16 ? (const void*)gemm5_kernel<true> : (const void*)gemm5_kernel<false>;
  if (int st = ensure_dynamic_lds(kern, g5::LDS_BYTES, &lds_attr[f16], "gemm5")) return st;
  ProfScope prof(f16 ? PROF_CONV : PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);
  note_kernel(DK_GEMM5_160x256);
  if (f16) hipLaunchKernelGGL(gemm5_kernel<true>, dim3(grid), dim3(g5::THREADS), g5::LDS_BYTES, stream, p, sp);
  else hipLaunchKernelGGL(gemm5_kernel<false>, dim3(grid), dim3(g5::THREADS), g5::LDS_BYTES, stream, p, sp);
  return check_launch("gemm5");
}

}  // namespace rtv
