// FP8 (OCP e4m3) projection GEMM + dynamic per-tensor activation quantisation for gfx950.
//
// Mirrors the reference's optional fp8 weight path: release_server.py:179-182 runs
//   torchao.quantize_(transformer, Float8DynamicActivationFloat8WeightConfig(granularity=PerTensor()))
// i.e. every nn.Linear computes  y = bf16( (q(x) . q(W)^T) * s_x * s_w + bias )  with
//   s = max|t| / 448 (fp32, over the WHOLE tensor, clamped to >= 1e-12),  q(t) = e4m3( clamp(t / s, -448, 448) ),
// weights quantised once, activations per call ("dynamic"), products accumulated in fp32 (torch._scaled_mm).  torchao is
// a third-party dependency that is not part of the reference tree (unpinned in its pyproject): the algorithm above is
// restated from its published source; oracle/wan_oracle.py carries the same restatement (parity for this mode is
// pinned to that restatement only).
//
// The GEMM is gemm8.hip's 256x256 ping-pong schedule with one change of unit: a K-tile is 128 fp8 elements, so every
// byte-level quantity (128-byte rows, 16 KiB half-tiles, DMA pieces, swizzle, fragment reads per phase) is identical,
// while each phase issues 4 v_mfma_f32_32x32x64_f8f6f4 (64 cycles, 4x the FLOPs of a 32x32x16 bf16 MFMA) instead of
// 8 bf16 MFMAs: same time per K-tile, twice the work.  Fragment layout (scripts/micro/fp8_mfma.hip): lane l holds row
// l & 31 and the 32 consecutive K bytes 32 * (l >> 5) .. +32 of a 64-deep step = two 16-byte LDS chunks.
#include <type_traits>

#include "gemm_core.h"
#include "gemm_split.h"
#include "rtv_internal.h"

namespace rtv {

namespace gf8 {
constexpr int BM = 256, BN = 256, BK = 128;      // BK in fp8 elements = bytes
constexpr int HALF_ROWS = 128;
constexpr int HALF_BYTES = HALF_ROWS * BK;       // 16 KiB
constexpr int LDS_BYTES = 8 * HALF_BYTES;        // 128 KiB
constexpr int THREADS = 512;
constexpr float FP8_MAX = 448.f;

__device__ __forceinline__ int slot_off(int buf, int h) { return (buf * 4 + h) * HALF_BYTES; }
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) int v4i;
__device__ __forceinline__ f32x16 mfma_fp8(const v8i& a, const v8i& b, const f32x16& c) {
  // cbsz = blgp = 0: both operands e4m3.  Zero scale operands select the plain v_mfma_f32_32x32x64_f8f6f4 (no block
  // scaling, no scale registers) instead of v_mfma_scale_*.
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
}
// one MFMA operand = two 16-byte LDS chunks landing in 8 consecutive registers
__device__ __forceinline__ v8i load_frag(const char* lo, const char* hi) {
  const v4i a = *(const v4i*)lo, b = *(const v4i*)hi;
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct Fp8Args {
  const uint8_t* A;      // [M][lda] e4m3
  const uint8_t* W;      // [N][ldw] e4m3
  int lda, ldw;          // bytes
  const float* a_scale;  // device: s_x of this call (written by the quantisation kernel)
  float w_scale;         // s_w
};
}  // namespace gf8

__global__ __launch_bounds__(gf8::THREADS, 2) void gemm_fp8_kernel(GemmParams p, gf8::Fp8Args f, SplitArgs sp) {
  using namespace gf8;
  typedef TileCfg<256, 256, 64, 2, 4> Cfg;  // epilogue geometry: 4 x 2 blocks of 32x32 per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int l31 = lane & 31, g = lane >> 5;

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

  // ---- DMA geometry (bytes): a half-tile is 2 wave-wide pieces per wave; piece j of wave w fills rows
  //      j*64 + w*8 .. +8 (lane -> row + lane/8, chunk slot lane%8), source chunk pre-swizzled.
  uint32_t src_off[4][2];
  {
    const int rsub = lane >> 3, cpos = lane & 7;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = j * 64 + wave * 8 + rsub;
      const int ch = swz(row, cpos) * 16;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int gm_row = min(m0 + h * HALF_ROWS + row, p.M - 1);
        const int gn_row = min(n0 + h * HALF_ROWS + row, p.N - 1);
        src_off[h][j] = (uint32_t)gm_row * (uint32_t)f.lda + ch;
        src_off[2 + h][j] = (uint32_t)gn_row * (uint32_t)f.ldw + ch;
      }
    }
  }
  // chk: std::true_type only in the last two K-tiles (gemm8.hip: the steady-state loop stages without bounds checks)
  auto stage_piece = [&](int kt, int h, int j, auto chk) {
    if (decltype(chk)::value && kt >= kt_end) return;
    const uint8_t* base = (h < 2 ? f.A : f.W) + (size_t)kt * BK;
    dma16(base + src_off[h][j], smem + slot_off(kt & 1, h) + (j * 64 + wave * 8) * 128);
  };
  auto stage_half = [&](int kt, int h) {
    stage_piece(kt, h, 0, std::true_type{});
    stage_piece(kt, h, 1, std::true_type{});
  };

  // ---- fragments:
  //      the operand of 64-deep step st = bytes [64 st + 32 g, +32) of the row = LDS chunks 4 st + 2 g and + 1
  const int a_slot = wr;
  const int b_slot = 2 + (wc >> 1);
  const int b_row0 = (wc & 1) * 64;
  v8i af[2][2], bfr[2][2], bnext[2];   // [block][64-deep step]
  auto frag = [&](const char* s, int row, int st) {
    const int c0 = st * 4 + g * 2;
    return load_frag(s + row * 128 + (swz(row, c0) << 4), s + row * 128 + (swz(row, c0 + 1) << 4));
  };
  auto read_a = [&](int buf, int mq) {
    const char* s = smem + slot_off(buf, a_slot);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int st = 0; st < 2; ++st) af[mb][st] = frag(s, mq * 64 + mb * 32 + l31, st);
  };
  auto read_w = [&](int buf, int nq, v8i (&dst)[2]) {
    const char* s = smem + slot_off(buf, b_slot);
#pragma unroll
    for (int st = 0; st < 2; ++st) dst[st] = frag(s, b_row0 + nq * 32 + l31, st);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

#define GF_FENCE() __builtin_amdgcn_sched_barrier(0)
#define GF_BARRIER()                  \
  do {                                \
    GF_FENCE();                       \
    __builtin_amdgcn_s_barrier();     \
    GF_FENCE();                       \
  } while (0)
#define GF_PHASE_SYNC()                                 \
  do {                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    GF_BARRIER();                                       \
  } while (0)

  // MFMA segment of one phase: 4 MFMA 32x32x64 on one 64x32 quadrant, the phase's 2 DMA pieces between them
  auto mma_quadrant = [&](int mq, int nq, int st_kt, int st_h, auto chk) {
    __builtin_amdgcn_s_setprio(1);
    int n = 0;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        acc[mq * 2 + mb][nq] = mfma_fp8(bfr[nq][s], af[mb][s], acc[mq * 2 + mb][nq]);
        ++n;
        if (n == 1 || n == 3) {
          GF_FENCE();
          stage_piece(st_kt, st_h, n == 1 ? 0 : 1, chk);
          GF_FENCE();
        }
      }
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- prologue / K loop: gemm8.hip's schedule (phases, counted vmcnt(2), one-barrier stagger of the wave groups)
  stage_half(kt_begin, 2);
  stage_half(kt_begin, 3);
  stage_half(kt_begin, 0);
  stage_half(kt_begin, 1);
  stage_half(kt_begin + 1, 2);
  stage_half(kt_begin + 1, 3);
  if (kt_begin + 1 < kt_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GF_BARRIER();
  if (wr == 1) GF_BARRIER();
  read_w(kt_begin & 1, 0, bnext);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  GF_FENCE();

  auto k_tile = [&](const int kt, auto chk) {
    constexpr bool CHK = decltype(chk)::value;
    const int buf = kt & 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) bfr[0][j] = bnext[j];
    read_a(buf, 0);
    GF_PHASE_SYNC();
    mma_quadrant(0, 0, kt + 1, 0, chk);
    GF_BARRIER();
    read_w(buf, 1, bfr[1]);
    if (!CHK || kt + 1 < kt_end) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    GF_PHASE_SYNC();
    mma_quadrant(0, 1, kt + 1, 1, chk);
    GF_BARRIER();
    read_a(buf, 1);
    GF_PHASE_SYNC();
    mma_quadrant(1, 1, kt + 2, 2, chk);
    GF_BARRIER();
    if (!CHK || kt + 1 < kt_end) read_w(buf ^ 1, 0, bnext);
    if (!CHK || kt + 2 < kt_end) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GF_PHASE_SYNC();
    mma_quadrant(1, 0, kt + 2, 3, chk);
    GF_BARRIER();
  };
  int kt = kt_begin;
  for (; kt + 2 < kt_end; ++kt) k_tile(kt, std::false_type{});
  for (; kt < kt_end; ++kt) k_tile(kt, std::true_type{});
  if (wr == 0) GF_BARRIER();
#undef GF_FENCE
#undef GF_BARRIER
#undef GF_PHASE_SYNC

  if (is_split && !split_k_reduce<4>(acc, sp, unit, seg, tile_id, smem, tid, wave, lane)) return;

  // de-quantise: (q(x) . q(W)^T) * s_x * s_w, then the shared bf16 epilogue (bias / activation / gate / residual)
  const float sc = f.a_scale[0] * f.w_scale;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] *= sc;
  const bool wide = !((p.ldc | (p.residual ? p.ldr : 0)) & 7) && !(((uintptr_t)p.C | (uintptr_t)p.residual) & 15);
  if (wide) {
    if (is_split) __syncthreads();
    store_tile_lds<false, 4>(p, m0 + wr * 128, n0 + wc * 64, lane, smem + wave * (128 * 128), acc);   // gemm_core.h
  } else {
    store_tile<false, Cfg>(p, m0 + wr * 128, n0 + wc * 64, lane, acc);
  }
}

// ------------------------------------------------------------------ dynamic per-tensor activation quantisation
// pass 1: amax_bits = max over the tensor of |x| (non-negative floats order like their bit patterns -> integer atomicMax).
// One workgroup per group of rows, 16-byte chunks strided over the threads (no integer division in the loop).
__global__ __launch_bounds__(256) void absmax_bf16_kernel(const uint16_t* __restrict__ x, int64_t ld, int M, int d,
                                                         unsigned* __restrict__ amax_bits) {
  const int cpr = d >> 3;  // 16-byte chunks per row
  float m = 0.f;
  for (int r = blockIdx.x; r < M; r += gridDim.x) {
    const uint16_t* xr = x + (int64_t)r * ld;
    for (int c = threadIdx.x; c < cpr; c += 256) {
      const u32x4 raw = *(const u32x4*)(xr + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t u = raw[j];
        m = fmaxf(m, fmaxf(fabsf(__builtin_bit_cast(float, u << 16)), fabsf(__builtin_bit_cast(float, u & 0xffff0000u))));
      }
    }
  }
  __shared__ float red[4];
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)   // one same-address atomic per workgroup
    atomicMax(amax_bits, __builtin_bit_cast(unsigned, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// pass 2: scale = max(amax, 1e-12) / 448 ; q = e4m3(clamp(x / scale, -448, 448)) ; also publishes `scale`
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const uint16_t* __restrict__ x, int64_t ld, int M, int d,
                                                          const unsigned* __restrict__ amax_bits, uint8_t* __restrict__ q,
                                                          int64_t ldq, float* __restrict__ scale_out) {
  const float amax = fmaxf(__builtin_bit_cast(float, amax_bits[0]), 1e-12f);
  // torch evaluates `amax / 448.0` (tensor / Python scalar) on the GPU as a multiplication by the fp32 reciprocal
  const float scale = amax * (1.0f / gf8::FP8_MAX);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = scale;
  const int cpr = d >> 3;
  for (int r = blockIdx.x; r < M; r += gridDim.x) {
    const uint16_t* xr = x + (int64_t)r * ld;
    uint8_t* qr = q + (int64_t)r * ldq;
    for (int c = threadIdx.x; c < cpr; c += 256) {
      const u32x4 raw = *(const u32x4*)(xr + c * 8);
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t u = raw[j];
        v[2 * j] = __builtin_bit_cast(float, u << 16);
        v[2 * j + 1] = __builtin_bit_cast(float, u & 0xffff0000u);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fminf(fmaxf(v[j] / scale, -gf8::FP8_MAX), gf8::FP8_MAX);
      u32x2 o;
      int w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], w1, true);
      o[0] = (uint32_t)w0;
      o[1] = (uint32_t)w1;
      *(u32x2*)(qr + c * 8) = o;
    }
  }
}

int launch_quantize_fp8(const uint16_t* x, int64_t ld, int M, int d, uint8_t* q, int64_t ldq, float* scale_out,
                        unsigned* amax_scratch, hipStream_t stream) {
  if (M <= 0 || d <= 0) return set_error(-1, "quantize_fp8: empty tensor");
  if ((d & 15) || (ld & 7) || (ldq & 7)) return set_error(-1, "quantize_fp8: d must be a multiple of 16, strides of 8");
  if (hipMemsetAsync(amax_scratch, 0, sizeof(unsigned), stream) != hipSuccess) return set_error(-1, "quantize_fp8: memset failed");
  const int blocks = M < 1024 ? M : 1024;
  ProfScope prof(PROF_MISC, stream, 0.0);
  hipLaunchKernelGGL(absmax_bf16_kernel, dim3(blocks), dim3(256), 0, stream, x, ld, M, d, amax_scratch);
  hipLaunchKernelGGL(quantize_fp8_kernel, dim3(blocks), dim3(256), 0, stream, x, ld, M, d, amax_scratch, q, ldq, scale_out);
  return check_launch("quantize_fp8");
}

int launch_gemm_fp8(GemmParams p, const uint8_t* A, int lda, const uint8_t* W, int ldw, const float* a_scale, float w_scale,
                    hipStream_t stream) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0) return set_error(-1, "gemm_fp8: empty problem");
  if (p.K % gf8::BK) return set_error(-1, "gemm_fp8: K must be a multiple of 128");
  if (p.N % 8) return set_error(-1, "gemm_fp8: N must be a multiple of 8");
  if ((lda % 16) || (ldw % 16) || (p.ldc % 4) || (p.residual && (p.ldr % 4)))
    return set_error(-1, "gemm_fp8: leading dimensions must keep 16-byte (A,W) / 8-byte (C,res) alignment");
  if (p.gate && p.rows_per_frame <= 0) return set_error(-1, "gemm_fp8: gate needs rows_per_frame");
  p.tiles_m = (p.M + gf8::BM - 1) / gf8::BM;
  p.tiles_n = (p.N + gf8::BN - 1) / gf8::BN;
  static LdsAttr lds_attr;   // per device (a second GPU used from this process needs the attribute as well)
  if (int st = ensure_dynamic_lds((const void*)gemm_fp8_kernel, gf8::LDS_BYTES, &lds_attr, "gemm_fp8")) return st;
  SplitArgs sp;
  int grid = 0;
  if (int st = plan_split_k(p.tiles_m * p.tiles_n, p.K / gf8::BK, true, &sp, &grid, stream)) return st;
  gf8::Fp8Args f{A, W, lda, ldw, a_scale, w_scale};
  ProfScope prof(PROF_GEMM, stream, 2.0 * p.M * (double)p.N * p.K);
  note_kernel(DK_GEMM_FP8_256x256);
  hipLaunchKernelGGL(gemm_fp8_kernel, dim3(grid), dim3(gf8::THREADS), gf8::LDS_BYTES, stream, p, f, sp);
  return check_launch("gemm_fp8");
}

}  // namespace rtv

using namespace rtv;

extern "C" int rtv_quantize_fp8(const void* x, int64_t ld, int M, int d, void* q, int64_t ldq, float* scale_out,
                                void* amax_scratch, rtv_stream_t stream) {
  if (!x || !q || !scale_out || !amax_scratch) return set_error(-1, "quantize_fp8: null pointer");
  return launch_quantize_fp8((const uint16_t*)x, ld, M, d, (uint8_t*)q, ldq, scale_out, (unsigned*)amax_scratch,
                             (hipStream_t)stream);
}

extern "C" int rtv_gemm_fp8(const void* A, int lda, const void* W, int ldw, const float* a_scale, float w_scale, void* C,
                            int ldc, int M, int N, int K, const void* bias, int act, const void* gate, int gate_stride,
                            int rows_per_frame, int row_offset, const void* residual, int ldr, rtv_stream_t stream) {
  if (!A || !W || !C || !a_scale) return set_error(-1, "gemm_fp8: null pointer");
  GemmParams p;
  p.A = nullptr;
  p.W = nullptr;
  p.C = (uint16_t*)C;
  p.lda = lda;
  p.ldw = ldw;
  p.ldc = ldc;
  p.M = M;
  p.N = N;
  p.K = K;
  p.bias = (const uint16_t*)bias;
  p.act = act;
  p.gate = (const uint16_t*)gate;
  p.gate_stride = gate_stride;
  p.rows_per_frame = rows_per_frame;
  p.row_offset = row_offset;
  p.residual = (const uint16_t*)residual;
  p.ldr = ldr;
  p.tiles_m = p.tiles_n = 0;
  return launch_gemm_fp8(p, (const uint8_t*)A, lda, (const uint8_t*)W, ldw, a_scale, w_scale, (hipStream_t)stream);
}
