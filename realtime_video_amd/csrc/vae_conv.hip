// Implicit-GEMM causal convolution for the streaming VAE decoder and encoder (gfx950, fp16, channels-last).
//
// Replaces the cuDNN nn.Conv3d / nn.Conv2d calls of the reference decoder: CausalConv3d 3x3x3
// (wan/modules/vae.py:17-36, used by ResidualBlock :175-209 and VAEDecoder3d vae_block3.py:358,:384),
// the temporal-upsampling time_conv (3,1,1) (vae_block3.py:29,:61-67) and the nearest-2x + Conv2d 3x3
// of Resample (vae_block3.py:19-28,:69-72; Upsample vae.py:57-63).
//
// out[pixel][co] = bias[co] + sum_{tap, ci} in[src(pixel, tap)][ci] * W[co][tap][ci]
//   * activations are [T][H][W][C] (channels-last) so the im2col row of one (pixel, tap) is one
//     contiguous run of Cin halves: the A tile is gathered straight into LDS by the same
//     global_load_lds DMA as the dense GEMM (gemm_core.h); out-of-image taps read a zero page, so the
//     spatial zero padding costs no branches in the MMA loop;
//   * the causal time padding is data, not padding: the conv input is a "concat buffer"
//     [2 cached slices | T new slices], output frame t reads slices t..t+2 (fused cache-concat);
//   * nearest-neighbour 2x upsampling is folded into the gather (source = coord >> 1); the encoder's
//     ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2) (vae.py:84-92) and its time_conv (3,1,1)/stride (2,1,1)
//     (vae.py:96, :151-156) are the same gather with a stride on the output grid;
//   * the time_conv epilogue scatters channel halves to frames 2t / 2t+1 (the reshape/stack of
//     vae_block3.py:65-67), the ResidualBlock's `x + h` rides in the epilogue as well.
#include "gemm_core.h"
#include "rtv_internal.h"

namespace rtv {

struct ConvParams {
  const uint16_t* in;   // [Tin][inH][inW][Cin]
  const uint16_t* w;    // [Cout][taps][Cin]
  uint16_t* out;
  const uint16_t* bias;      // [Cout] or null
  const uint16_t* residual;  // [M][res_ld] or null
  const uint16_t* zeros;     // >= 16 bytes of zeros
  int out_ld, res_ld;
  int T, H, W;      // output grid
  int inH, inW;     // input grid (H >> ups)
  int Cin, Cout;
  int kt, kh, kw;
  int w_taps;       // taps per filter row of `w` (its row stride is w_taps * Cin elements): kt * kh * kw, or MORE when the launch uses
                    // a suffix of a wider tap set in place - the last time tap of a 3x3x3 weight as a 1x3x3 convolution (w points at
                    // tap 18 of filter 0, w_taps = 27: the fresh one-frame VAE encode, whose two cached time slices are zeros)
  int ups;          // 1: input is read through a nearest 2x upsampling
  int sy, st;       // spatial / temporal stride of the output grid over the input (1 or 2)
  int pad_h, pad_w; // low-side zero padding (kh/2 for 'same' convs, 0 for the stride-2 downsample: its ZeroPad2d is high-side)
  int limH, limW;   // taps outside [0,limH) x [0,limW) read zeros (input dims, or output dims when ups)
  int y_out0, y_in0; // row windows (spatially sharded decode): output row py is image row y_out0 + py, input buffer row 0
                     // is image row y_in0 (in input resolution); limH is then the IMAGE height
  int in_rows;       // rows held by the input buffer
  int gather;       // 1: keep a 3x3x3 stride-1 conv on the gather kernel (RTV_CONV_GATHER)
  const uint16_t* norm_gamma;  // halo kernel, Cout == 96, no residual: the epilogue writes SiLU(RMS_norm(conv + bias) * gamma)
                               // (wan/modules/vae.py:39-54 + the nn.SiLU behind it, :186-192) instead of conv + bias
  int n_split;      // >0: output channel n -> frame 2t + n / n_split, channel n % n_split
  int M;            // T*H*W
  int tiles_m, tiles_n;
};

// Wave-private LDS image of a TM*32 x TN*32 output block (rows of TN*64 bytes) for the coalesced epilogue: 16-byte
// chunk c of row r is stored at a permuted position so that both the ds_write_b64 of the MFMA layout (32 rows x one
// 8-byte column slot per instruction) and the row-contiguous ds_read_b128 are bank-conflict free.
template <int TN>
__device__ __forceinline__ int conv_img_off(int row, int chunk, int half) {
  if constexpr (TN == 2)
    return row * 128 + ((chunk ^ (row & 7)) << 4) + ((half ^ ((row >> 3) & 1)) << 3);
  else   // 4 or 12 chunks per row: row bases repeat every 4 rows (64 / 192-byte rows), so rows r, r+4, r+8, r+12 get
    return row * (TN * 64) + ((chunk ^ ((row >> 2) & 3)) << 4) + (half << 3);   // different chunks of a group of 4
}

// Implicit-GEMM convolution, one BM-pixel x BN-filter tile per workgroup.
//  * K runs over (tap, channel slab of BK); the im2col rows of a slab are gathered straight into LDS by global_load_lds.
//    The per-lane source pointers are rebuilt only when the TAP changes (row / column offsets of the 3 x 3 neighbours and
//    their in-image bits are precomputed per lane), a slab step just adds the uniform channel offset: ~25 VALU per
//    K-step and wave instead of ~90 in the round-1 kernel, where address arithmetic competed with the MFMAs for issue.
//  * three LDS stages, the DMA two K-steps ahead, ONE raw s_barrier per step and a counted s_waitcnt vmcnt (round 1: two
//    stages, the DMA of step k+1 issued and waited for inside step k behind __syncthreads' full drain);
//  * epilogue through LDS: every lane stores 16 contiguous bytes of an output pixel row (round 1: 8-byte stores at a
//    pixel stride - the same store tail that cost the GEMM a third of its time, profiles/r02_gemm_lab_v2_vs_v1.log).
template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(ConvParams p) {
  typedef TileCfg<BM, BN, BK, WM, WN> Cfg;
  constexpr int STAGES = 3;
  constexpr int PIECES = Cfg::A_INST + Cfg::B_INST;   // DMA instructions per wave and K-step
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // consecutive workgroups walk the (few) N tiles of the same pixel tile: they share the gathered
  // activation rows through L2
  const int id = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tile_m = id / p.tiles_n, tile_n = id % p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int cpos = lane % Cfg::CH;
  const int rsub = lane / Cfg::CH;
  const int HW = p.H * p.W;
  // ---- per-lane gather state of the A rows this lane stages.  The source of tap (dt, dy, dx) is the pixel's own input
  //      position plus a UNIFORM offset: ((dt * inH + dy) * inW + dx) * Cin without upsampling; with the nearest-2x
  //      upsampling folded into the gather the source row of output row y + dy is (y + dy) >> 1 = (y >> 1) + floor((dy +
  //      (y & 1)) / 2), i.e. one of two uniform offsets chosen by the parity of the lane's row (same for columns).  Per lane:
  //      the base offset, the parities and the in-image bits of its 3 x 3 neighbourhood.
  int base_off[Cfg::A_INST], flags[Cfg::A_INST];   // flags: bits 0-2 row k in image, 3-5 column k in image, 6 row odd, 7 col odd
#pragma unroll
  for (int i = 0; i < Cfg::A_INST; ++i) {
    const int row = (wave * Cfg::A_INST + i) * Cfg::RPI + rsub;
    const int m = min(m0 + row, p.M - 1);
    const int pt = m / HW;
    const int rem = m - pt * HW;
    const int py = rem / p.W, px = rem - py * p.W;
    const int yb = p.y_out0 + py * p.sy, xb = px * p.sy;
    int f = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int yy = yb + k - p.pad_h, xx = xb + k - p.pad_w;
      if (yy >= 0 && yy < p.limH) f |= 1 << k;
      if (xx >= 0 && xx < p.limW) f |= 8 << k;
    }
    if (p.ups) f |= ((yb & 1) << 6) | ((xb & 1) << 7);
    flags[i] = f;
    base_off[i] = ((pt * p.st * p.inH + ((yb >> p.ups) - p.y_in0)) * p.inW + (xb >> p.ups)) * p.Cin + Cfg::swz(row, cpos) * 8;
  }
  uint32_t b_off[Cfg::B_INST];
  const int Ktot = p.w_taps * p.Cin;   // weight row stride (elements)
#pragma unroll
  for (int i = 0; i < Cfg::B_INST; ++i) {
    int row = (wave * Cfg::B_INST + i) * Cfg::RPI + rsub;
    int gn = min(n0 + row, p.Cout - 1);
    b_off[i] = (uint32_t)gn * (uint32_t)Ktot + Cfg::swz(row, cpos) * 8;
  }
  const int cpk = p.Cin / BK;  // K-steps per tap
  const int nk = p.kt * p.kh * p.kw * cpk;
  const int slice = p.inH * p.inW * p.Cin;   // elements per input time slice

  // ---- DMA issue cursor (uniform): tap (dt, dy, dx) and the channel slab inside it
  int is_dt = 0, is_dy = 0, is_dx = 0, is_c = 0, is_ks = 0;
  const uint16_t* a_src[Cfg::A_INST];   // source of the current tap's row for this lane (zero page when outside the image)
  int a_live[Cfg::A_INST];              // 1: add the slab offset, 0: the zero page is read at offset 0
  auto tap_setup = [&]() __attribute__((always_inline)) {
    // uniform offsets of this tap for even / odd output rows and columns (equal without upsampling)
    const int ry = is_dy - p.pad_h, rx = is_dx - p.pad_w;
    const int row_e = p.ups ? (ry >> 1) : ry, row_o = p.ups ? ((ry + 1) >> 1) : ry;     // arithmetic shift = floor
    const int col_e = p.ups ? (rx >> 1) : rx, col_o = p.ups ? ((rx + 1) >> 1) : rx;
    const int t_off = is_dt * slice;
    const int d_re = t_off + row_e * p.inW * p.Cin, d_ro = t_off + row_o * p.inW * p.Cin;
    const int d_ce = col_e * p.Cin, d_co = col_o * p.Cin;
#pragma unroll
    for (int i = 0; i < Cfg::A_INST; ++i) {
      const int f = flags[i];
      const bool ok = ((f >> is_dy) & (f >> (3 + is_dx)) & 1) != 0;
      const int off = base_off[i] + ((f & 64) ? d_ro : d_re) + ((f & 128) ? d_co : d_ce);
      a_src[i] = ok ? p.in + (ptrdiff_t)off : p.zeros;
      a_live[i] = ok ? 1 : 0;
    }
  };
  auto issue = [&]() __attribute__((always_inline)) {   // stage K-step is_ks into slot is_ks % STAGES, advance the cursor
    char* sA = smem + (is_ks % STAGES) * Cfg::STAGE_BYTES;
    char* sB = sA + Cfg::A_BYTES;
    if (is_c == 0) tap_setup();
#pragma unroll
    for (int i = 0; i < Cfg::A_INST; ++i) dma16(a_src[i] + a_live[i] * is_c, sA + (wave * Cfg::A_INST + i) * 1024);
    const uint16_t* Wk = p.w + (size_t)is_ks * BK;
#pragma unroll
    for (int i = 0; i < Cfg::B_INST; ++i) dma16(Wk + b_off[i], sB + (wave * Cfg::B_INST + i) * 1024);
    ++is_ks;
    is_c += BK;
    if (is_c == p.Cin) {
      is_c = 0;
      if (++is_dx == p.kw) {
        is_dx = 0;
        if (++is_dy == p.kh) {
          is_dy = 0;
          ++is_dt;
        }
      }
    }
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

#define CV_FENCE() __builtin_amdgcn_sched_barrier(0)
  issue();
  if (nk > 1) issue();
  for (int ks = 0; ks < nk; ++ks) {
    // step ks has landed for this wave (the pieces of step ks + 1 may stay in flight) ...
    if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CV_FENCE();
    __builtin_amdgcn_s_barrier();   // ... and for every wave; everybody is done reading the slot of step ks - 1
    CV_FENCE();
    if (ks + 2 < nk) issue();       // -> slot (ks + 2) % 3 = the slot of step ks - 1
    CV_FENCE();
    const char* sA = smem + (ks % STAGES) * Cfg::STAGE_BYTES;
    mma_stage<true, Cfg, BK>(sA, sA + Cfg::A_BYTES, a_row0, b_row0, lane, acc);
    CV_FENCE();
  }
  __builtin_amdgcn_s_barrier();     // all fragment reads done: the stage buffers become the epilogue image
  CV_FENCE();
#undef CV_FENCE

  // ---- epilogue: bias (+ residual), optional channel-half -> frame scatter, through a wave-private LDS image so that every
  //      lane moves 16 contiguous bytes of a pixel's channels
  constexpr int TM = Cfg::TM, TN = Cfg::TN, CPR = TN * 4;   // 16-byte chunks per image row
  char* img = smem + wave * (TM * 32 * TN * 64);
  const int l31 = lane & 31, g = lane >> 5;
  const bool wide = !((p.out_ld | (p.residual ? p.res_ld : 0) | p.n_split) & 7) &&
                    !(((uintptr_t)p.out | (uintptr_t)p.residual) & 15);
  if (wide) {
    // the residual of this wave's output passes, requested ahead of the accumulator conversion (otherwise every pass waits for its own
    // load: r05, see conv_halo_kernel)
    constexpr int NPASS = TM * 32 * CPR / 64;
    u32x4 res_pre[NPASS];
    if (p.residual) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int q = ps * 64 + lane;
        const int row = q / CPR, c = q - row * CPR;
        const int m = min(m0 + a_row0 + row, p.M - 1), n = min(n0 + b_row0 + c * 8, p.Cout - 8);   // (clamped: rows / chunks past the end are never stored)
        res_pre[ps] = *(const u32x4*)(p.residual + (size_t)m * p.res_ld + n);
      }
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      const int row = mi * 32 + l31;
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int n = min(n0 + b_row0 + ni * 32 + rq * 8 + g * 4, p.Cout - 4);
          float v[4] = {acc[mi][ni][rq * 4 + 0], acc[mi][ni][rq * 4 + 1], acc[mi][ni][rq * 4 + 2], acc[mi][ni][rq * 4 + 3]};
          if (p.bias) {
            const u32x2 bb = *(const u32x2*)(p.bias + n);
            v[0] += f16_to_f32(bb[0] & 0xffff);
            v[1] += f16_to_f32(bb[0] >> 16);
            v[2] += f16_to_f32(bb[1] & 0xffff);
            v[3] += f16_to_f32(bb[1] >> 16);
          }
          u32x2 o;
          o[0] = pack_f16x2(v[0], v[1]);
          o[1] = pack_f16x2(v[2], v[3]);
          *(u32x2*)(img + conv_img_off<TN>(row, ni * 4 + rq, g)) = o;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    constexpr int PASSES = TM * 32 * CPR / 64;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int q = ps * 64 + lane;
      const int row = q / CPR, c = q - row * CPR;
      const int m = m0 + a_row0 + row;
      const int n = n0 + b_row0 + c * 8;
      u32x4 t;
      if constexpr (TN == 2) {
        t = *(const u32x4*)(img + conv_img_off<TN>(row, c, 0) - (((row >> 3) & 1) << 3));   // chunk start; halves swapped on odd octets
        if ((row >> 3) & 1) t = u32x4{t[2], t[3], t[0], t[1]};
      } else {
        t = *(const u32x4*)(img + conv_img_off<TN>(row, c, 0));
      }
      if (m >= p.M || n >= p.Cout) continue;
      if (p.residual) {
        const u32x4 rr = res_pre[ps];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float a0, a1, r0, r1;
          unpack_f16x2(t[i], a0, a1);
          unpack_f16x2(rr[i], r0, r1);
          t[i] = pack_f16x2(a0 + r0, a1 + r1);
        }
      }
      size_t drow = (size_t)m;
      int ch = n;
      if (p.n_split > 0) {
        const int tt = m / HW, pix = m - tt * HW;
        const int half = n >= p.n_split ? 1 : 0;
        drow = (size_t)(2 * tt + half) * HW + pix;
        ch = n - half * p.n_split;
      }
      *(u32x4*)(p.out + drow * p.out_ld + ch) = t;
    }
    return;
  }
  // narrow fallback (channel strides / pointers that do not allow 16-byte accesses): 8-byte stores from the MFMA layout
#pragma unroll
  for (int mi = 0; mi < Cfg::TM; ++mi) {
    const int m = m0 + a_row0 + mi * 32 + l31;
    if (m >= p.M) continue;
    const int t = m / HW, pix = m - t * HW;
#pragma unroll
    for (int ni = 0; ni < Cfg::TN; ++ni)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n = n0 + b_row0 + ni * 32 + rq * 8 + g * 4;
        if (n >= p.Cout) continue;
        float v[4] = {acc[mi][ni][rq * 4 + 0], acc[mi][ni][rq * 4 + 1], acc[mi][ni][rq * 4 + 2],
                      acc[mi][ni][rq * 4 + 3]};
        if (p.bias) {
          u32x2 bb = *(const u32x2*)(p.bias + n);
          v[0] += f16_to_f32(bb[0] & 0xffff);
          v[1] += f16_to_f32(bb[0] >> 16);
          v[2] += f16_to_f32(bb[1] & 0xffff);
          v[3] += f16_to_f32(bb[1] >> 16);
        }
        if (p.residual) {
          u32x2 rr = *(const u32x2*)(p.residual + (size_t)m * p.res_ld + n);
          v[0] = round_f16(v[0]) + f16_to_f32(rr[0] & 0xffff);
          v[1] = round_f16(v[1]) + f16_to_f32(rr[0] >> 16);
          v[2] = round_f16(v[2]) + f16_to_f32(rr[1] & 0xffff);
          v[3] = round_f16(v[3]) + f16_to_f32(rr[1] >> 16);
        }
        size_t drow;
        int ch;
        if (p.n_split > 0) {
          const int half = n >= p.n_split ? 1 : 0;
          drow = (size_t)(2 * t + half) * HW + pix;
          ch = n - half * p.n_split;
        } else {
          drow = (size_t)m;
          ch = n;
        }
        u32x2 o;
        o[0] = pack_f16x2(v[0], v[1]);
        o[1] = pack_f16x2(v[2], v[3]);
        *(u32x2*)(p.out + drow * p.out_ld + ch) = o;
      }
  }
}

// ---------------------------------------------------------------- halo-tile kernel (round 3)
// The 3x3x3 stride-1 convolutions of the decoder's ResidualBlocks (wan/modules/vae.py:175-209) at 96 / 192 channels are 71 % of
// the VAE's FLOPs.  conv_igemm_kernel gathers the im2col rows of every one of the 27 taps from global
// memory again: 7 LDS-DMA instructions per 12 MFMAs and wave, issue-bound on staging (MFMA busy 28 %, profiles/r01_pmc_hot_kernels.txt).
// Here a workgroup owns a 16 x 32 pixel tile of one frame x 96 filters and stages, per (time slice dt, 32-channel chunk), the
// 18 x 34 HALO of that tile once (39 KiB, double buffered); the nine spatial taps of the chunk are then MFMAs whose A fragments
// are ds_read_b128 at tap-shifted addresses of the same LDS image - 5 halo + 9 weight DMA instructions per 108 MFMAs and wave.
//   * 8 waves, wave w = tile rows 2w, 2w+1 (two 32-pixel m-blocks) x 96 filters: 6 accumulator blocks, 12 MFMAs per tap;
//   * weights stream per "tap row" (dt, chunk, dy): [dx][96 filters][32 channels] = 18 KiB through a 3-slot ring;
//   * one raw s_barrier + one counted vmcnt per tap row (36 MFMAs per wave); every wave issues the same number of DMA
//     instructions per step (the few surplus ones repeat a piece), so the counts are compile-time constants;
//   * bank conflicts: a pixel / filter row is 64 bytes, its 16-byte chunk c sits at slot c ^ ((column >> 2) & 3) - keyed by the
//     COLUMN only, so a tap's row shift is an immediate offset and its column shift one of three precomputed per-lane bases; the
//     16 lanes of a quarter wave (16 consecutive columns) then cover all 16 slots of a 256-byte bank row.
// K order per output pixel: (dt, chunk) outer, (dy, dx) inner - independent of the tile position, so row-sharded and
// unsharded decodes stay bit-identical.
namespace ch {
constexpr int TH = 16, TW = 32, HPITCH = 34, HROWS = 18;
constexpr int HPIX = HROWS * HPITCH;        // 612 halo pixels
constexpr int HREAL = 39;                   // 1-KiB pieces (16 pixels x 64 bytes) that cover them
constexpr int HBYTES = HREAL * 1024;
constexpr int WROW_BYTES = 3 * 96 * 64;     // one tap row of weights: [dx][filter][32 channels]
constexpr int WREAL = 18;
constexpr int LDS_BYTES = 2 * HBYTES + 3 * WROW_BYTES;   // 135168
constexpr int THREADS = 512;
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
// Fragment read as inline asm: behind a pending LDS DMA the compiler's wait-count pass waits lgkmcnt(0) in front of the first
// consumer of ANY plain LDS load (measured in the first build of this kernel: two full drains per tap row); the asm form is
// invisible to it and the consumer side waits with a counted lgkmcnt that the fragment registers depend on.
template <int OFF>
__device__ __forceinline__ u32x4 lds_read128(uint32_t lds_addr) {
  u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(lds_addr), "n"(OFF));
  return r;
}
template <int CNT>
__device__ __forceinline__ void lds_wait(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(CNT));
}
}  // namespace ch

__global__ __launch_bounds__(ch::THREADS, 2) void conv_halo_kernel(ConvParams p, int tiles_x, int tiles_y) {
  using namespace ch;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, gl = lane >> 5;

  // workgroup -> (frame, tile, filter tile): the filter tiles of one pixel tile are neighbours (they share the halo through L2)
  const int tiles_n = p.Cout / 96;
  const int id = xcd_remap(blockIdx.x, p.T * tiles_y * tiles_x * tiles_n);
  const int tn = id % tiles_n;
  int rest = id / tiles_n;
  const int tx = rest % tiles_x;
  rest /= tiles_x;
  const int ty = rest % tiles_y, t = rest / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH, n0 = tn * 96;

  const int cpk = p.Cin / 32;                      // channel chunks per time slice
  const int G = p.kt * cpk;                        // (dt, chunk) groups (kt = 3: causal 3x3x3; kt = 1: the 3x3 conv behind the
  const int slice = p.inH * p.inW * p.Cin;         //  nearest-2x upsampling, r05)
  const int taps = p.w_taps;   // weight row stride in taps (9 * kt unless the launch uses a tap suffix in place)

  // ---- halo DMA geometry: wave w issues pieces 5w .. 5w+4 (piece 39 repeats 38); lane -> halo pixel q = 16 piece + lane / 4,
  //      LDS slot lane % 4, source chunk = slot ^ key(halo column)
  int h_off[5], h_ok = 0, h_lds[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int piece = min(wave * 5 + i, HREAL - 1);
    h_lds[i] = piece * 1024;
    const int q = piece * 16 + (lane >> 2);
    const int hr = q / HPITCH, hc = q - hr * HPITCH;
    const int y = y0 - 1 + hr, x = x0 - 1 + hc;
    // p.ups: the conv runs over the nearest-2x upsampled image - halo pixel (y, x) is source pixel (y >> 1, x >> 1); a row window
    // (row-sharded decode) holds image rows from y_out0 on in the output and from y_in0 on (source resolution) in the input, and
    // a tap outside the window but inside the image reads its real source row (the caller supplies them)
    const int yi = p.y_out0 + y;                   // image row
    const bool ok = q < HPIX && yi >= 0 && yi < p.limH && x >= 0 && x < p.limW;
    const int ys = p.ups ? (yi >> 1) - p.y_in0 : y, xs = p.ups ? x >> 1 : x;
    const int c = (lane & 3) ^ ((hc >> 2) & 3);
    h_off[i] = ((t * p.inH + ys) * p.inW + xs) * p.Cin + c * 8;
    if (ok) h_ok |= 1 << i;
  }
  // ---- weight DMA geometry: wave w issues pieces 3w .. 3w+2 of the 18 (pieces 18..23 repeat 10..15); piece = (dx, 16 filters)
  int w_off[3], w_lds[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int piece = wave * 3 + i;
    if (piece >= WREAL) piece -= 8;
    w_lds[i] = piece * 1024;
    const int dx = piece / 6, f = (piece % 6) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((f >> 2) & 3);
    w_off[i] = ((n0 + f) * taps + dx) * p.Cin + c * 8;
  }
  auto issue_h = [&](int g, int i0, int i1) __attribute__((always_inline)) {   // pieces i0..i1-1 of group g's halo
    g = min(g, G - 1);
    const int dt = g / cpk, cb = g - dt * cpk;
    const int uoff = dt * slice + cb * 32;
    char* dst = smem + (g & 1) * HBYTES;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const uint16_t* src = ((h_ok >> i) & 1) ? p.in + (ptrdiff_t)(h_off[i] + uoff) : p.zeros;
      dma16(src, dst + h_lds[i]);
    }
  };
  auto issue_w = [&](int g, int dy, int slot) __attribute__((always_inline)) {  // tap row (g, dy) -> ring slot
    const int gg = min(g, G - 1);
    const int dt = gg / cpk, cb = gg - dt * cpk;
    const int uoff = (dt * 9 + dy * 3) * p.Cin + cb * 32;
    char* dst = smem + 2 * HBYTES + slot * WROW_BYTES;
#pragma unroll
    for (int i = 0; i < 3; ++i) dma16(p.w + (ptrdiff_t)(w_off[i] + uoff), dst + w_lds[i]);
  };

  // ---- fragment read addresses (bytes inside a halo buffer / a weight ring slot)
  int a_base[3][2], b_base[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hc = l31 + dx;
      a_base[dx][ks] = ((2 * wave) * HPITCH + hc) * 64 + (((2 * ks + gl) ^ ((hc >> 2) & 3)) << 4);
    }
    b_base[ks] = l31 * 64 + (((2 * ks + gl) ^ ((l31 >> 2) & 3)) << 4);
  }

  f32x16 acc[2][3];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

#define CH_FENCE() __builtin_amdgcn_sched_barrier(0)
  // prologue: halo(0), W(0, dy 0), W(0, dy 1)
  issue_h(0, 0, 5);
  issue_w(0, 0, 0);
  issue_w(0, 1, 1);
  int wslot = 0;   // ring slot of the current tap row
  const uint32_t lds0 = (uint32_t)(uintptr_t)(RTV_LDS const char*)smem;
  for (int g = 0; g < G; ++g) {
    const uint32_t hb = lds0 + (g & 1) * HBYTES;
    sfor<0, 3>([&](auto dyc) {
      constexpr int dy = decltype(dyc)::value;
      // W(this tap row) [and, at dy 0, this group's halo] have landed for this wave; younger pieces stay in flight:
      //   dy 0: W(next row) = 3;  dy 1: W(prev. issue) 3 + halo part 3 = 6;  dy 2: halo 3 + W 3 + halo 2 = 8
      if (dy == 0) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else if (dy == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      CH_FENCE();
      __builtin_amdgcn_s_barrier();   // ... for every wave; everybody is done with the previous tap row (and group)
      CH_FENCE();
      // the tap row two ahead -> the ring slot of the previous one; the next group's halo -> the other halo buffer
      {
        const int slot2 = wslot == 0 ? 2 : wslot - 1;
        if (dy == 0) issue_w(g, 2, slot2);
        else issue_w(g + 1, dy - 1, slot2);
        if (dy == 0) issue_h(g + 1, 0, 3);
        if (dy == 1) issue_h(g + 1, 3, 5);
      }
      CH_FENCE();
      const uint32_t wb = lds0 + 2 * HBYTES + wslot * WROW_BYTES;
      uint32_t aaddr[3][2], baddr[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        baddr[ks] = wb + b_base[ks];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) aaddr[dx][ks] = hb + a_base[dx][ks];
      }
      // the six (tap, k-step) chunks of the row, software pipelined two chunks deep: the 5 fragment reads of chunk i + 2 are
      // issued in front of the 6 MFMAs of chunk i (three register sets, at most 15 LDS reads in flight)
      u32x4 af[3][2], bf[3][3];   // [set][block]
      auto load_chunk = [&](auto cic) __attribute__((always_inline)) {
        constexpr int ci = decltype(cic)::value;
        constexpr int dx = ci >> 1, ks = ci & 1, set = ci % 3;
        bf[set][0] = lds_read128<dx * (96 * 64) + 0 * (32 * 64)>(baddr[ks]);
        bf[set][1] = lds_read128<dx * (96 * 64) + 1 * (32 * 64)>(baddr[ks]);
        bf[set][2] = lds_read128<dx * (96 * 64) + 2 * (32 * 64)>(baddr[ks]);
        af[set][0] = lds_read128<(0 + dy) * (HPITCH * 64)>(aaddr[dx][ks]);
        af[set][1] = lds_read128<(1 + dy) * (HPITCH * 64)>(aaddr[dx][ks]);
      };
      load_chunk(IC<0>{});
      load_chunk(IC<1>{});
      sfor<0, 6>([&](auto cic) {
        constexpr int ci = decltype(cic)::value;
        constexpr int set = ci % 3;
        if constexpr (ci + 2 < 6) load_chunk(IC<ci + 2>{});
        lds_wait<(ci + 2 < 6) ? 10 : (ci + 1 < 6 ? 5 : 0)>(bf[set][0], bf[set][1], bf[set][2], af[set][0], af[set][1]);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int nb = 0; nb < 3; ++nb) acc[mb][nb] = Mfma32<true>::run(bf[set][nb], af[set][mb], acc[mb][nb]);
        CH_FENCE();   // keep the chunk's MFMAs in front of the next chunk's wait
      });
      CH_FENCE();
      wslot = wslot == 2 ? 0 : wslot + 1;
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus pieces of the last steps
  CH_FENCE();
  __builtin_amdgcn_s_barrier();     // all fragment reads and DMA writes done: LDS becomes the epilogue image
  CH_FENCE();
#undef CH_FENCE

  // ---- the residual of this wave's 12 output passes, requested NOW: the passes below otherwise each wait for their own load
  //      (load -> add -> store, twelve times ~1 us: a third of the kernel on the 96-channel layers, r05 lab builds of the
  //      one-wave-per-SIMD form); the accumulator conversion runs under the loads' latency instead
  u32x4 res_pre[2 * 32 * 12 / 64];
  if (p.residual) {
#pragma unroll
    for (int ps = 0; ps < 2 * 32 * 12 / 64; ++ps) {
      const int q = ps * 64 + lane;
      const int row = q / 12, c = q - row * 12;
      const int y = y0 + 2 * wave + (row >> 5), x = x0 + (row & 31);
      res_pre[ps] = u32x4{0u, 0u, 0u, 0u};
      if (y < p.H && x < p.W) res_pre[ps] = *(const u32x4*)(p.residual + (((size_t)t * p.H + y) * p.W + x) * p.res_ld + n0 + c * 8);
    }
  }
  // ---- epilogue: bias (+ residual) through a wave-private 64 x 96 image (rows of 192 bytes), 16 contiguous bytes per lane.
  //      Fused RMS_norm + SiLU (p.norm_gamma, 96 filters = the whole channel axis of a pixel in this workgroup): lane (l31, gl)
  //      holds 48 of pixel l31's 96 outputs, lane + 32 the other 48 - sum of squares in the lane, one exchange, then
  //      SiLU(y * sqrt(96) / max(|y|, 1e-12) * gamma) on the fp16-rounded y, the arithmetic of rmsnorm_silu_cl_kernel.
  constexpr int TM = 2, TN = 3, CPR = TN * 4;
  char* img = smem + wave * (TM * 32 * TN * 64);
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int row = mi * 32 + l31;
    float yv[TN][4][4];
    float ssq = 0.f;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n = n0 + ni * 32 + rq * 8 + gl * 4;
        float v[4] = {acc[mi][ni][rq * 4 + 0], acc[mi][ni][rq * 4 + 1], acc[mi][ni][rq * 4 + 2], acc[mi][ni][rq * 4 + 3]};
        if (p.bias) {
          const u32x2 bb = *(const u32x2*)(p.bias + n);
          v[0] += f16_to_f32(bb[0] & 0xffff);
          v[1] += f16_to_f32(bb[0] >> 16);
          v[2] += f16_to_f32(bb[1] & 0xffff);
          v[3] += f16_to_f32(bb[1] >> 16);
        }
        u32x2 o;
        o[0] = pack_f16x2(v[0], v[1]);
        o[1] = pack_f16x2(v[2], v[3]);
        if (p.norm_gamma) {
          unpack_f16x2(o[0], yv[ni][rq][0], yv[ni][rq][1]);   // the conv output as the unfused path stores it
          unpack_f16x2(o[1], yv[ni][rq][2], yv[ni][rq][3]);
#pragma unroll
          for (int i = 0; i < 4; ++i) ssq += yv[ni][rq][i] * yv[ni][rq][i];
        } else {
          *(u32x2*)(img + conv_img_off<TN>(row, ni * 4 + rq, gl)) = o;
        }
      }
    if (p.norm_gamma) {
      ssq += __shfl_xor(ssq, 32, 64);
      const float inv = sqrtf(96.f) / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const u32x2 gg = *(const u32x2*)(p.norm_gamma + ni * 32 + rq * 8 + gl * 4);
          float g4[4];
          unpack_f16x2(gg[0], g4[0], g4[1]);
          unpack_f16x2(gg[1], g4[2], g4[3]);
          float z[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) z[i] = silu(yv[ni][rq][i] * inv * g4[i]);
          u32x2 o;
          o[0] = pack_f16x2(z[0], z[1]);
          o[1] = pack_f16x2(z[2], z[3]);
          *(u32x2*)(img + conv_img_off<TN>(row, ni * 4 + rq, gl)) = o;
        }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  constexpr int PASSES = TM * 32 * CPR / 64;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int q = ps * 64 + lane;
    const int row = q / CPR, c = q - row * CPR;
    const int y = y0 + 2 * wave + (row >> 5), x = x0 + (row & 31);
    const int n = n0 + c * 8;
    u32x4 tv = *(const u32x4*)(img + conv_img_off<TN>(row, c, 0));
    if (y >= p.H || x >= p.W) continue;
    const size_t m = ((size_t)t * p.H + y) * p.W + x;
    if (p.residual) {
      const u32x4 rr = res_pre[ps];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a0, a1, r0, r1;
        unpack_f16x2(tv[i], a0, a1);
        unpack_f16x2(rr[i], r0, r1);
        tv[i] = pack_f16x2(a0 + r0, a1 + r1);
      }
    }
    *(u32x4*)(p.out + m * p.out_ld + n) = tv;
  }
}

// ---------------------------------------------------------------- halo-tile kernel, one wave per SIMD (round 5, lab: rtv_conv_set_halo(2))
// The same tile, LDS images, DMA pieces, K order and epilogue arithmetic as conv_halo_kernel (bit-identical results), in the
// one-wave-per-SIMD idiom of attn_w4.hip / gemm5.hip: FOUR waves, wave w = tile rows 4w .. 4w+3 (four 32-pixel m-blocks) x 96
// filters = 12 accumulator blocks a[0:191]; a (tap, k-step) chunk is 12 MFMAs fed by 7 fragment reads (4 halo + 3 weight blocks)
// where the two-waves-per-SIMD kernel reads 2 x 5 for the same 12 - 30 % fewer LDS bytes per flop.  Accumulators and the two
// fragment sets a[192:219] / a[220:247] are named literally in inline asm; the reads of chunk c + 1 sit behind the MFMAs of
// chunk c (across tap rows too: no drain at a row boundary).  One counted vmcnt + one barrier per tap row, placed behind MFMA 2
// of the row's LAST chunk: past it every wave has finished the row's reads, so its weight slot takes the row three ahead
// (two rows of lead) and, after the last row of a group, its halo buffer the group after next (three rows of lead).
namespace ch4 {
using namespace ch;
constexpr int THREADS4 = 256;
constexpr int HP = 10, WP = 5;                 // halo / weight DMA pieces per wave (pieces past the real ones repeat one)
constexpr int A_ACC = 0, A_FRAG = 192, FRAG_SET = 28;
template <int ACC, int WF, int AF>
__device__ __forceinline__ void mfma_aaa() {
  asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], a[%c2:%c3], a[%c4:%c5], a[%c0:%c1]" ::"n"(ACC), "n"(ACC + 15), "n"(WF), "n"(WF + 3),
               "n"(AF), "n"(AF + 3));
}
template <int DST, int OFF>
__device__ __forceinline__ void lds_read128_a(uint32_t lds_addr) {
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%3" ::"v"(lds_addr), "n"(DST), "n"(DST + 3), "n"(OFF));
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
// (the one clobber list of the kernel: makes the kernel descriptor allocate the accumulation registers it names literally)
#define RTV_CH4_ACC \
  "a0", "a15", "a31", "a47", "a63", "a79", "a95", "a111", "a127", "a143", "a159", "a175", "a191", "a207", "a223", "a239", "a247"
}  // namespace ch4

template <int LAB>   // lab builds (timing only, garbage results): 1 no DMA in the loop, 2 no epilogue, 3 DMA issue staggered by wave (not yet)
__global__ __launch_bounds__(ch4::THREADS4, 1) void conv_halo4_kernel(ConvParams p, int tiles_x, int tiles_y) {
  using namespace ch4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, gl = lane >> 5;

  const int tiles_n = p.Cout / 96;
  const int id = xcd_remap(blockIdx.x, p.T * tiles_y * tiles_x * tiles_n);
  const int tn = id % tiles_n;
  int rest = id / tiles_n;
  const int tx = rest % tiles_x;
  rest /= tiles_x;
  const int ty = rest % tiles_y, t = rest / tiles_y;
  const int x0 = tx * TW, y0 = ty * TH, n0 = tn * 96;

  const int cpk = p.Cin / 32;
  const int G = p.kt * cpk;
  const int slice = p.inH * p.inW * p.Cin;
  const int taps = p.w_taps;   // weight row stride in taps (9 * kt unless the launch uses a tap suffix in place)

  // ---- halo DMA geometry (conv_halo_kernel's pieces, ten per wave): piece -> 16 halo pixels x 64 bytes.  Buffer addressing: the
  //      per-lane BYTE offset of the pixel's chunk at (dt 0, channel chunk 0) in a VGPR, the group's offset in an SGPR - no vector
  //      instruction per piece; a pixel outside the image gets an offset beyond the descriptor's end and reads zeros.
  uint32_t h_voff[HP], h_lds[HP];
#pragma unroll
  for (int i = 0; i < HP; ++i) {
    const int piece = min(wave * HP + i, HREAL - 1);
    h_lds[i] = piece * 1024;
    const int q = piece * 16 + (lane >> 2);
    const int hr = q / HPITCH, hc = q - hr * HPITCH;
    const int y = y0 - 1 + hr, x = x0 - 1 + hc;
    const int yi = p.y_out0 + y;
    const bool ok = q < HPIX && yi >= 0 && yi < p.limH && x >= 0 && x < p.limW;
    const int ys = p.ups ? (yi >> 1) - p.y_in0 : y, xs = p.ups ? x >> 1 : x;
    const int c = (lane & 3) ^ ((hc >> 2) & 3);
    h_voff[i] = ok ? (uint32_t)(((t * p.inH + ys) * p.inW + xs) * p.Cin + c * 8) * 2u : 0x80000000u;
  }
  // ---- weight DMA geometry: piece = (dx, 16 filters); wave w issues pieces 5w .. 5w+4 of the 18 (18, 19 repeat 16, 17)
  uint32_t w_voff[WP], w_lds[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    int piece = wave * WP + i;
    if (piece >= WREAL) piece -= 2;
    w_lds[i] = piece * 1024;
    const int dx = piece / 6, f = (piece % 6) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ ((f >> 2) & 3);
    w_voff[i] = (uint32_t)(((n0 + f) * taps + dx) * p.Cin + c * 8) * 2u;
  }
  const __amdgpu_buffer_rsrc_t rsrcI = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0x80000000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(RTV_LDS const char*)smem;
  // piece I of a halo (byte offset `soff` of its (dt, channel chunk)) -> halo buffer `buf`
  auto issue_h = [&](uint32_t soff, uint32_t buf, auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    RTV_LDS void* dst = (RTV_LDS void*)(uintptr_t)(lds0 + buf * (uint32_t)HBYTES + h_lds[I]);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcI, dst, 16, h_voff[I], soff, 0, 0);
  };
  // piece I of a tap row of weights (byte offset `soff` of its (dt, dy, channel chunk)) -> ring slot SLOT
  auto issue_w = [&](uint32_t soff, auto slotc, auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value, SLOT = decltype(slotc)::value;
    RTV_LDS void* dst = (RTV_LDS void*)(uintptr_t)(lds0 + (uint32_t)(2 * HBYTES + SLOT * WROW_BYTES) + w_lds[I]);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, dst, 16, w_voff[I], soff, 0, 0);
  };
  // scalar byte offsets of group g (clamped: groups past the end re-stage the last one into the buffer / slot of THEIR index,
  // which nobody reads any more): halo (dt, chunk) and tap row (dt, dy, chunk)
  auto h_soff = [&](int dt, int cb) __attribute__((always_inline)) { return (uint32_t)(dt * slice + cb * 32) * 2u; };
  auto w_soff = [&](int dt, int cb, int dy) __attribute__((always_inline)) {
    return (uint32_t)((dt * 9 + dy * 3) * p.Cin + cb * 32) * 2u;
  };

  // ---- fragment read addresses: halo pixel (row 4 wave + mi + dy [immediate], column l31 + dx), chunk 2 ks + gl; weights:
  //      ring slot dy (part of the address: an LDS immediate is 16 bits), filter l31, (dx, filter block) immediates
  uint32_t a_base[3][2], b_addr[3][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hc = l31 + dx;
      a_base[dx][ks] = lds0 + (uint32_t)(((4 * wave) * HPITCH + hc) * 64 + (((2 * ks + gl) ^ ((hc >> 2) & 3)) << 4));
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
      b_addr[dy][ks] = lds0 + (uint32_t)(2 * HBYTES + dy * WROW_BYTES + l31 * 64 + (((2 * ks + gl) ^ ((l31 >> 2) & 3)) << 4));
  }
  uint32_t a_cur[3][2], a_nxt[3][2];
  // the 7 reads of chunk (DY, CI) [CI = 2 dx + ks] into fragment set SET: N = 0..2 weight blocks, 3..6 halo rows
  auto frag_read = [&](auto dyc, auto cic, auto setc, auto nc, const uint32_t (&aa)[3][2]) __attribute__((always_inline)) {
    constexpr int DY = decltype(dyc)::value, CI = decltype(cic)::value, SET = decltype(setc)::value, N = decltype(nc)::value;
    constexpr int dx = CI >> 1, ks = CI & 1;
    if constexpr (N < 3) lds_read128_a<A_FRAG + FRAG_SET * SET + 4 * N, dx * (96 * 64) + N * (32 * 64)>(b_addr[DY][ks]);
    else lds_read128_a<A_FRAG + FRAG_SET * SET + 12 + 4 * (N - 3), ((N - 3) + DY) * (HPITCH * 64)>(aa[dx][ks]);
  };

  asm volatile("" ::: RTV_CH4_ACC);
  sfor<0, 192>([&](auto ic) __attribute__((always_inline)) { acc_zero<A_ACC + decltype(ic)::value>(); });

  // ---- prologue: halo(0), W rows 0 and 1 (row 0 then issues W(2) and halo(1) like every first row of a group); row 0's operands
  //      landed -> barrier -> the first fragment set
  sfor<0, HP>([&](auto ic) __attribute__((always_inline)) { issue_h(h_soff(0, 0), 0u, ic); });
  sfor<0, WP>([&](auto ic) __attribute__((always_inline)) { issue_w(w_soff(0, 0, 0), IC<0>{}, ic); });
  sfor<0, WP>([&](auto ic) __attribute__((always_inline)) { issue_w(w_soff(0, 0, 1), IC<1>{}, ic); });
  asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // W(1) may still be in flight
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_cur[dx][ks] = a_base[dx][ks];
  sfor<0, 7>([&](auto nc) __attribute__((always_inline)) { frag_read(IC<0>{}, IC<0>{}, IC<0>{}, nc, a_cur); });

  int dt = 0, cb = 0;   // (time slice, channel chunk) of group g
  for (int g = 0; g < G; ++g) {
    const uint32_t nsel = (g & 1) ? 0u : (uint32_t)HBYTES;   // the NEXT group's halo buffer
    // what this group's three rows stage: W(g, 2), W(g + 1, 0), W(g + 1, 1) and the halo of group g + 1 (clamped to the last)
    const bool more = g + 1 < G;
    const int cb1 = cb + 1 == cpk ? 0 : cb + 1, dt1 = cb + 1 == cpk ? dt + 1 : dt;
    uint32_t ws[3], hs;
    ws[0] = w_soff(dt, cb, 2);
    ws[1] = more ? w_soff(dt1, cb1, 0) : ws[0];
    ws[2] = more ? w_soff(dt1, cb1, 1) : ws[0];
    hs = more ? h_soff(dt1, cb1) : h_soff(dt, cb);
    const uint32_t nbuf = (uint32_t)((g + 1) & 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) a_nxt[dx][ks] = a_base[dx][ks] + nsel;
    sfor<0, 3>([&](auto dyc) __attribute__((always_inline)) {
      constexpr int DY = decltype(dyc)::value;
      sfor<0, 6>([&](auto cic) __attribute__((always_inline)) {
        constexpr int CI = decltype(cic)::value, SET = CI & 1;
        asm volatile("s_waitcnt lgkmcnt(0)");   // the 7 reads of this chunk (issued behind MFMAs of the previous one)
        sfor<0, 12>([&](auto nc) __attribute__((always_inline)) {
          constexpr int n = decltype(nc)::value, mi = n / 3, ni = n % 3;
          mfma_aaa<A_ACC + (mi * 3 + ni) * 16, A_FRAG + FRAG_SET * SET + 4 * ni, A_FRAG + FRAG_SET * SET + 12 + 4 * mi>();
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (CI < 5) {
            if constexpr (n < 7) frag_read(dyc, IC<CI + 1>{}, IC<SET ^ 1>{}, nc, a_cur);
            // the DMA pieces of this epoch (behind the barrier of the previous row): W(row + 2) -> the slot of row - 1 (chunk 0), and
            // behind the barrier of a group's last row the halo of group g + 1 (row 3g: pieces 0-4 in chunk 1, 5-9 in chunk 2)
            if constexpr (LAB != 1) {
              if constexpr (CI == 0 && n >= 7) issue_w(ws[DY], IC<(DY + 2) % 3>{}, IC<n - 7>{});
              if constexpr (DY == 0 && (CI == 1 || CI == 2) && n >= 7) issue_h(hs, nbuf, IC<(CI - 1) * 5 + n - 7>{});
            }
          } else {
            if constexpr (n == 2) {
              // every wave's pieces for the next row (and, behind a group's last row, the next group's halo) have landed; younger
              // pieces stay in flight: DY 0: W(row + 2) 5 + halo(g + 1) 10;  DY 1: the same 15;  DY 2: W(row + 2) 5
              if constexpr (LAB == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              else if constexpr (DY == 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
              else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (n >= 3 && n < 10) {   // chunk 0 of the next tap row
              if constexpr (DY < 2) frag_read(IC<DY + 1>{}, IC<0>{}, IC<SET ^ 1>{}, IC<n - 3>{}, a_cur);
              else frag_read(IC<0>{}, IC<0>{}, IC<SET ^ 1>{}, IC<n - 3>{}, a_nxt);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      });
    });
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) a_cur[dx][ks] = a_nxt[dx][ks];
    dt = dt1;
    cb = cb1;
  }
  asm volatile("s_waitcnt lgkmcnt(0)");               // the surplus fragment reads of the last chunk
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ... and DMA pieces: the LDS becomes the epilogue image
  __builtin_amdgcn_s_barrier();
  asm volatile("s_nop 15\n\ts_nop 7");                // MFMA -> v_accvgpr_read distance
  if constexpr (LAB == 2) {
    if (acc_read<0>() == 12345.678f) p.out[tid] = 1;
    return;
  }

  // ---- epilogue: conv_halo_kernel's arithmetic, per 32-pixel row block of the wave; wave-private 128 x 96 image.  The residual
  //      of all 24 output passes is requested first (see conv_halo_kernel).
  constexpr int TM = 4, TN = 3, CPR = TN * 4;
  u32x4 res_pre[TM * 32 * CPR / 64];
  if (p.residual) {
#pragma unroll
    for (int ps = 0; ps < TM * 32 * CPR / 64; ++ps) {
      const int q = ps * 64 + lane;
      const int row = q / CPR, c = q - row * CPR;
      const int y = y0 + 4 * wave + (row >> 5), x = x0 + (row & 31);
      res_pre[ps] = u32x4{0u, 0u, 0u, 0u};
      if (y < p.H && x < p.W) res_pre[ps] = *(const u32x4*)(p.residual + (((size_t)t * p.H + y) * p.W + x) * p.res_ld + n0 + c * 8);
    }
  }
  char* img = smem + wave * (TM * 32 * TN * 64);
  sfor<0, TM>([&](auto mic) __attribute__((always_inline)) {
    constexpr int mi = decltype(mic)::value;
    f32x16 accv[TN];
    sfor<0, TN * 16>([&](auto ic) __attribute__((always_inline)) {
      constexpr int i = decltype(ic)::value;
      accv[i >> 4][i & 15] = acc_read<A_ACC + mi * 48 + i>();
    });
    const int row = mi * 32 + l31;
    float yv[TN][4][4];
    float ssq = 0.f;
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const int n = n0 + ni * 32 + rq * 8 + gl * 4;
        float v[4] = {accv[ni][rq * 4 + 0], accv[ni][rq * 4 + 1], accv[ni][rq * 4 + 2], accv[ni][rq * 4 + 3]};
        if (p.bias) {
          const u32x2 bb = *(const u32x2*)(p.bias + n);
          v[0] += f16_to_f32(bb[0] & 0xffff);
          v[1] += f16_to_f32(bb[0] >> 16);
          v[2] += f16_to_f32(bb[1] & 0xffff);
          v[3] += f16_to_f32(bb[1] >> 16);
        }
        u32x2 o;
        o[0] = pack_f16x2(v[0], v[1]);
        o[1] = pack_f16x2(v[2], v[3]);
        if (p.norm_gamma) {
          unpack_f16x2(o[0], yv[ni][rq][0], yv[ni][rq][1]);
          unpack_f16x2(o[1], yv[ni][rq][2], yv[ni][rq][3]);
#pragma unroll
          for (int i = 0; i < 4; ++i) ssq += yv[ni][rq][i] * yv[ni][rq][i];
        } else {
          *(u32x2*)(img + conv_img_off<TN>(row, ni * 4 + rq, gl)) = o;
        }
      }
    if (p.norm_gamma) {
      ssq += __shfl_xor(ssq, 32, 64);
      const float inv = sqrtf(96.f) / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const u32x2 gg = *(const u32x2*)(p.norm_gamma + ni * 32 + rq * 8 + gl * 4);
          float g4[4];
          unpack_f16x2(gg[0], g4[0], g4[1]);
          unpack_f16x2(gg[1], g4[2], g4[3]);
          float z[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) z[i] = silu(yv[ni][rq][i] * inv * g4[i]);
          u32x2 o;
          o[0] = pack_f16x2(z[0], z[1]);
          o[1] = pack_f16x2(z[2], z[3]);
          *(u32x2*)(img + conv_img_off<TN>(row, ni * 4 + rq, gl)) = o;
        }
    }
  });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  constexpr int PASSES = TM * 32 * CPR / 64;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const int q = ps * 64 + lane;
    const int row = q / CPR, c = q - row * CPR;
    const int y = y0 + 4 * wave + (row >> 5), x = x0 + (row & 31);
    const int n = n0 + c * 8;
    u32x4 tv = *(const u32x4*)(img + conv_img_off<TN>(row, c, 0));
    if (y >= p.H || x >= p.W) continue;
    const size_t m = ((size_t)t * p.H + y) * p.W + x;
    if (p.residual) {
      const u32x4 rr = res_pre[ps];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float a0, a1, r0, r1;
        unpack_f16x2(tv[i], a0, a1);
        unpack_f16x2(rr[i], r0, r1);
        tv[i] = pack_f16x2(a0 + r0, a1 + r1);
      }
    }
    *(u32x4*)(p.out + m * p.out_ld + n) = tv;
  }
}

// ---------------------------------------------------------------- ... and PERSISTENT (round 5): one workgroup per CU walks its tiles
// (virtual block ids b, b + gridDim, ..: the ids the hardware would have dispatched to this CU's XCD) and the operand stream never
// stops: where conv_halo4_kernel re-stages its last group past the end of a tile, this kernel stages the first group of its NEXT tile,
// so a tile boundary is just another group boundary of the (halo double buffer, weight ring) pipeline - no prologue, no dispatch gap,
// and the output stores of a tile drain under the next tile's matrix instructions.  At a boundary the wave reads its 12 accumulator
// blocks out (and zeroes them) one 32-pixel row at a time through a wave-private 6 KiB image in the 24 KiB of LDS the pipeline leaves
// free.  Same K order and epilogue arithmetic: bit-identical with the other two forms.
// Stores count in vmcnt and are not ordered against loads: the first barrier of a tile waits for vmcnt(0).
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(1, 2))) void conv_halo4p_kernel(ConvParams p, int tiles_x,
                                                                                                                      int tiles_y) {
  using namespace ch4;
  // accumulation-register map of THIS kernel: a[0:7] are left to the compiler (with a tile loop around the K loop hipcc parks a few
  // loop-carried values there; scripts/micro/h4_audit.sh checks that it touches nothing above a7), accumulators a[8:199], fragment sets
  // a[200:227] / a[228:255]
  constexpr int A_ACC = 8, A_FRAG = 200;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, gl = lane >> 5;

  const int tiles_n = p.Cout / 96;
  const int total = p.T * tiles_y * tiles_x * tiles_n;
  const int stride = gridDim.x;
  const int cpk = p.Cin / 32;
  const int G = p.kt * cpk;
  const int slice = p.inH * p.inW * p.Cin;
  const int taps = p.w_taps;   // weight row stride in taps (9 * kt unless the launch uses a tap suffix in place)

  auto decode = [&](int vid, int& t, int& y0, int& x0, int& n0) __attribute__((always_inline)) {
    const int id = xcd_remap(vid, total);
    const int tn = id % tiles_n;
    int rest = id / tiles_n;
    const int tx = rest % tiles_x;
    rest /= tiles_x;
    const int ty = rest % tiles_y;
    t = rest / tiles_y;
    x0 = tx * TW;
    y0 = ty * TH;
    n0 = tn * 96;
  };
  // DMA geometry of a tile (conv_halo4_kernel's): per-lane byte offsets of the ten halo pieces and five weight pieces of this wave
  uint32_t h_lds[HP], w_lds[WP];
#pragma unroll
  for (int i = 0; i < HP; ++i) h_lds[i] = min(wave * HP + i, HREAL - 1) * 1024;
#pragma unroll
  for (int i = 0; i < WP; ++i) w_lds[i] = (wave * WP + i >= WREAL ? wave * WP + i - 2 : wave * WP + i) * 1024;
  auto tile_voff = [&](int t, int y0, int x0, int n0, uint32_t (&hv)[HP], uint32_t (&wv)[WP]) __attribute__((always_inline)) {
    int lane_v = lane, wave_v = wave;   // (laundered: see the epilogue)
    asm volatile("" : "+v"(lane_v), "+s"(wave_v));
#pragma unroll
    for (int i = 0; i < HP; ++i) {
      const int piece = min(wave_v * HP + i, HREAL - 1);
      const int q = piece * 16 + (lane_v >> 2);
      const int hr = q / HPITCH, hc = q - hr * HPITCH;
      const int y = y0 - 1 + hr, x = x0 - 1 + hc;
      const int yi = p.y_out0 + y;
      const bool ok = q < HPIX && yi >= 0 && yi < p.limH && x >= 0 && x < p.limW;
      const int ys = p.ups ? (yi >> 1) - p.y_in0 : y, xs = p.ups ? x >> 1 : x;
      const int c = (lane_v & 3) ^ ((hc >> 2) & 3);
      hv[i] = ok ? (uint32_t)(((t * p.inH + ys) * p.inW + xs) * p.Cin + c * 8) * 2u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
      int piece = wave_v * WP + i;
      if (piece >= WREAL) piece -= 2;
      const int dx = piece / 6, f = (piece % 6) * 16 + (lane_v >> 2);
      const int c = (lane_v & 3) ^ ((f >> 2) & 3);
      wv[i] = (uint32_t)(((n0 + f) * taps + dx) * p.Cin + c * 8) * 2u;
    }
  };
  const __amdgpu_buffer_rsrc_t rsrcI = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0x80000000, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0x7fffffff, 0x00020000);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(RTV_LDS const char*)smem;
  auto issue_h = [&](uint32_t soff, uint32_t buf, uint32_t voff, auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value;
    RTV_LDS void* dst = (RTV_LDS void*)(uintptr_t)(lds0 + buf * (uint32_t)HBYTES + h_lds[I]);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcI, dst, 16, voff, soff, 0, 0);
  };
  auto issue_w = [&](uint32_t soff, auto slotc, uint32_t voff, auto ic) __attribute__((always_inline)) {
    constexpr int I = decltype(ic)::value, SLOT = decltype(slotc)::value;
    RTV_LDS void* dst = (RTV_LDS void*)(uintptr_t)(lds0 + (uint32_t)(2 * HBYTES + SLOT * WROW_BYTES) + w_lds[I]);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, dst, 16, voff, soff, 0, 0);
  };
  auto h_soff = [&](int dt, int cb) __attribute__((always_inline)) { return (uint32_t)(dt * slice + cb * 32) * 2u; };
  auto w_soff = [&](int dt, int cb, int dy) __attribute__((always_inline)) {
    return (uint32_t)((dt * 9 + dy * 3) * p.Cin + cb * 32) * 2u;
  };

  // fragment read addresses (conv_halo4_kernel's), recomputed per tile from a laundered lane id: nothing but scalars stays live
  // across a tile's epilogue, which then has the whole architectural file to itself (see the note there)
  uint32_t a_base[3][2], b_addr[3][2];
  auto frag_addrs = [&]() __attribute__((always_inline)) {
    int lane_k = lane, wave_k = wave;
    asm volatile("" : "+v"(lane_k), "+s"(wave_k));
    const int l31k = lane_k & 31, glk = lane_k >> 5;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int hc = l31k + dx;
        a_base[dx][ks] = lds0 + (uint32_t)(((4 * wave_k) * HPITCH + hc) * 64 + (((2 * ks + glk) ^ ((hc >> 2) & 3)) << 4));
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
        b_addr[dy][ks] = lds0 + (uint32_t)(2 * HBYTES + dy * WROW_BYTES + l31k * 64 + (((2 * ks + glk) ^ ((l31k >> 2) & 3)) << 4));
    }
  };
  frag_addrs();
  uint32_t a_cur[3][2], a_nxt[3][2];
  auto frag_read = [&](auto dyc, auto cic, auto setc, auto nc, const uint32_t (&aa)[3][2]) __attribute__((always_inline)) {
    constexpr int DY = decltype(dyc)::value, CI = decltype(cic)::value, SET = decltype(setc)::value, N = decltype(nc)::value;
    constexpr int dx = CI >> 1, ks = CI & 1;
    if constexpr (N < 3) lds_read128_a<A_FRAG + FRAG_SET * SET + 4 * N, dx * (96 * 64) + N * (32 * 64)>(b_addr[DY][ks]);
    else lds_read128_a<A_FRAG + FRAG_SET * SET + 12 + 4 * (N - 3), ((N - 3) + DY) * (HPITCH * 64)>(aa[dx][ks]);
  };

  asm volatile("" ::: RTV_CH4_ACC, "a255");
  sfor<0, 192>([&](auto ic) __attribute__((always_inline)) { acc_zero<A_ACC + decltype(ic)::value>(); });

  int vid = blockIdx.x;
  int ct, cy0, cx0, cn0;                 // the tile being accumulated
  uint32_t hv_c[HP], wv_c[WP], hv_n[HP], wv_n[WP];   // its DMA geometry, and the next tile's
  decode(vid, ct, cy0, cx0, cn0);
  tile_voff(ct, cy0, cx0, cn0, hv_c, wv_c);

  // ---- prologue of the FIRST tile: halo(0), W rows 0 and 1 -> barrier -> the first fragment set
  sfor<0, HP>([&](auto ic) __attribute__((always_inline)) { issue_h(h_soff(0, 0), 0u, hv_c[decltype(ic)::value], ic); });
  sfor<0, WP>([&](auto ic) __attribute__((always_inline)) { issue_w(w_soff(0, 0, 0), IC<0>{}, wv_c[decltype(ic)::value], ic); });
  sfor<0, WP>([&](auto ic) __attribute__((always_inline)) { issue_w(w_soff(0, 0, 1), IC<1>{}, wv_c[decltype(ic)::value], ic); });
  asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  uint32_t par = 0u;                      // halo buffer of the current group
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) a_cur[dx][ks] = a_base[dx][ks];
  sfor<0, 7>([&](auto nc) __attribute__((always_inline)) { frag_read(IC<0>{}, IC<0>{}, IC<0>{}, nc, a_cur); });

  for (bool first_tile = true;; first_tile = false) {
    const bool has_next = vid + stride < total;
    if (!first_tile) {   // (everything per-lane is rebuilt here rather than carried through the previous tile's epilogue)
      frag_addrs();
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a_cur[dx][ks] = a_base[dx][ks] + par * (uint32_t)HBYTES;
      tile_voff(ct, cy0, cx0, cn0, hv_c, wv_c);
    }
    {
      int nt = ct, ny0 = cy0, nx0 = cx0, nn0 = cn0;   // (no next tile: its slots are filled with this tile's first group again; nobody reads them)
      if (has_next) decode(vid + stride, nt, ny0, nx0, nn0);
      tile_voff(nt, ny0, nx0, nn0, hv_n, wv_n);
    }
    int dt = 0, cb = 0;   // (time slice, channel chunk) of group g
    for (int g = 0; g < G; ++g) {
      // what this group's three rows stage: W(g, 2), then rows 0 and 1 and the halo of the NEXT group - of this tile, or group 0 of the next
      const bool last = g + 1 == G;
      const int cb1 = last ? 0 : (cb + 1 == cpk ? 0 : cb + 1), dt1 = last ? 0 : (cb + 1 == cpk ? dt + 1 : dt);
      uint32_t ws[3], hs;
      ws[0] = w_soff(dt, cb, 2);
      ws[1] = w_soff(dt1, cb1, 0);
      ws[2] = w_soff(dt1, cb1, 1);
      hs = h_soff(dt1, cb1);
      const uint32_t nbuf = par ^ 1u;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a_nxt[dx][ks] = a_base[dx][ks] + nbuf * (uint32_t)HBYTES;
      sfor<0, 3>([&](auto dyc) __attribute__((always_inline)) {
        constexpr int DY = decltype(dyc)::value;
        sfor<0, 6>([&](auto cic) __attribute__((always_inline)) {
          constexpr int CI = decltype(cic)::value, SET = CI & 1;
          asm volatile("s_waitcnt lgkmcnt(0)");   // the 7 reads of this chunk (issued behind MFMAs of the previous one)
          sfor<0, 12>([&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value, mi = n / 3, ni = n % 3;
            mfma_aaa<A_ACC + (mi * 3 + ni) * 16, A_FRAG + FRAG_SET * SET + 4 * ni, A_FRAG + FRAG_SET * SET + 12 + 4 * mi>();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (CI < 5) {
              if constexpr (n < 7) frag_read(dyc, IC<CI + 1>{}, IC<SET ^ 1>{}, nc, a_cur);
              if constexpr (CI == 0 && n >= 7) {
                if constexpr (DY == 0) issue_w(ws[0], IC<2>{}, wv_c[n - 7], IC<n - 7>{});
                else issue_w(ws[DY], IC<DY - 1>{}, last ? wv_n[n - 7] : wv_c[n - 7], IC<n - 7>{});
              }
              if constexpr (DY == 0 && (CI == 1 || CI == 2) && n >= 7)
                issue_h(hs, nbuf, last ? hv_n[(CI - 1) * 5 + n - 7] : hv_c[(CI - 1) * 5 + n - 7], IC<(CI - 1) * 5 + n - 7>{});
            } else {
              if constexpr (n == 2) {
                if constexpr (DY == 2) {
                  asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                } else if constexpr (DY == 1) {
                  asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                } else {
                  if (g == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the previous tile's output stores are in the count
                  else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
              }
              if constexpr (n >= 3 && n < 10) {   // chunk 0 of the next tap row
                if constexpr (DY < 2) frag_read(IC<DY + 1>{}, IC<0>{}, IC<SET ^ 1>{}, IC<n - 3>{}, a_cur);
                else frag_read(IC<0>{}, IC<0>{}, IC<SET ^ 1>{}, IC<n - 3>{}, a_nxt);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          });
        });
      });
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a_cur[dx][ks] = a_nxt[dx][ks];
      par = nbuf;
      dt = dt1;
      cb = cb1;
    }

    // ---- the tile's epilogue (conv_halo_kernel's arithmetic), one 32-pixel row of the wave at a time through a wave-private 32 x 96
    //      image; the fragment reads of the next tile's first chunk and its DMA pieces stay in flight
    asm volatile("s_nop 15\n\ts_nop 7");   // MFMA -> v_accvgpr_read distance
    {
      constexpr int TN = 3, CPR = TN * 4, PPM = 32 * CPR / 64;   // 6 store passes per row block
      // (lane and wave laundered: everything the epilogue derives from them is invariant over the tile loop, and hipcc would hoist
      //  some 150 registers of store addresses out of it - over the whole K loop, into the accumulation file this kernel owns)
      int lane_e = lane, wave_e = wave;
      asm volatile("" : "+v"(lane_e), "+s"(wave_e));
      const int l31e = lane_e & 31, gle = lane_e >> 5;
      char* img = smem + LDS_BYTES + wave_e * (32 * TN * 64);
      u32x2 bias_pre[TN * 4];
      if (p.bias) {
#pragma unroll
        for (int j = 0; j < TN * 4; ++j) bias_pre[j] = *(const u32x2*)(p.bias + cn0 + (j >> 2) * 32 + (j & 3) * 8 + gle * 4);
      }
      u32x4 res_pre[PPM];   // the residual of the row block being converted (requested before its conversion)
      auto res_load = [&](auto mic) __attribute__((always_inline)) {
        constexpr int mi = decltype(mic)::value;
#pragma unroll
        for (int ps = 0; ps < PPM; ++ps) {
          const int q = ps * 64 + lane_e;
          const int row = q / CPR, c = q - row * CPR;
          const int y = min(cy0 + 4 * wave_e + mi, p.H - 1), x = min(cx0 + row, p.W - 1);   // (pixels past the image: clamped, never stored)
          res_pre[ps] = *(const u32x4*)(p.residual + (((size_t)ct * p.H + y) * p.W + x) * p.res_ld + cn0 + c * 8);
        }
      };
      if (p.residual) res_load(IC<0>{});
      sfor<0, 4>([&](auto mic) __attribute__((always_inline)) {
        constexpr int mi = decltype(mic)::value;
        f32x16 accv[TN];
        sfor<0, TN * 16>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          accv[i >> 4][i & 15] = acc_read<A_ACC + mi * 48 + i>();
        });
        sfor<0, TN * 16>([&](auto ic) __attribute__((always_inline)) { acc_zero<A_ACC + mi * 48 + decltype(ic)::value>(); });
        const int row = l31e;
        u32x2 ypk[TN * 4];   // norm path: the fp16-rounded conv outputs, packed (unpacked again for the second pass: 24 registers, not 48)
        float ssq = 0.f;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            float v[4] = {accv[ni][rq * 4 + 0], accv[ni][rq * 4 + 1], accv[ni][rq * 4 + 2], accv[ni][rq * 4 + 3]};
            if (p.bias) {
              const u32x2 bb = bias_pre[ni * 4 + rq];
              v[0] += f16_to_f32(bb[0] & 0xffff);
              v[1] += f16_to_f32(bb[0] >> 16);
              v[2] += f16_to_f32(bb[1] & 0xffff);
              v[3] += f16_to_f32(bb[1] >> 16);
            }
            u32x2 o;
            o[0] = pack_f16x2(v[0], v[1]);
            o[1] = pack_f16x2(v[2], v[3]);
            if (p.norm_gamma) {
              float y4[4];
              unpack_f16x2(o[0], y4[0], y4[1]);
              unpack_f16x2(o[1], y4[2], y4[3]);
#pragma unroll
              for (int i = 0; i < 4; ++i) ssq += y4[i] * y4[i];
              ypk[ni * 4 + rq] = o;
            } else {
              *(u32x2*)(img + conv_img_off<TN>(row, ni * 4 + rq, gle)) = o;
            }
          }
        if (p.norm_gamma) {
          ssq += __shfl_xor(ssq, 32, 64);
          const float inv = sqrtf(96.f) / fmaxf(sqrtf(ssq), 1e-12f);
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
              const u32x2 gg = *(const u32x2*)(p.norm_gamma + ni * 32 + rq * 8 + gle * 4);
              float g4[4], y4[4];
              unpack_f16x2(gg[0], g4[0], g4[1]);
              unpack_f16x2(gg[1], g4[2], g4[3]);
              unpack_f16x2(ypk[ni * 4 + rq][0], y4[0], y4[1]);
              unpack_f16x2(ypk[ni * 4 + rq][1], y4[2], y4[3]);
              float z[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) z[i] = silu(y4[i] * inv * g4[i]);
              u32x2 o;
              o[0] = pack_f16x2(z[0], z[1]);
              o[1] = pack_f16x2(z[2], z[3]);
              *(u32x2*)(img + conv_img_off<TN>(row, ni * 4 + rq, gle)) = o;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ps = 0; ps < PPM; ++ps) {
          const int q = ps * 64 + lane_e;
          const int r = q / CPR, c = q - r * CPR;
          const int y = cy0 + 4 * wave_e + mi, x = cx0 + r;
          const int n = cn0 + c * 8;
          u32x4 tv = *(const u32x4*)(img + conv_img_off<TN>(r, c, 0));
          if (y >= p.H || x >= p.W) continue;
          const size_t m = ((size_t)ct * p.H + y) * p.W + x;
          if (p.residual) {
            const u32x4 rr = res_pre[ps];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float a0, a1, r0, r1;
              unpack_f16x2(tv[i], a0, a1);
              unpack_f16x2(rr[i], r0, r1);
              tv[i] = pack_f16x2(a0 + r0, a1 + r1);
            }
          }
          *(u32x4*)(p.out + m * p.out_ld + n) = tv;
        }
        if constexpr (mi + 1 < 4) {
          if (p.residual) res_load(IC<mi + 1>{});
        }
      });
    }
    if (!has_next) break;
    vid += stride;
    decode(vid, ct, cy0, cx0, cn0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no DMA piece may land in this LDS after the workgroup has left
}

static int launch_conv_halo(ConvParams p, hipStream_t stream, int form) {   // form 0: two waves per SIMD; 1: one; 4: one, persistent (lab build: 2, 3 = timing forms of 1)
  const int tiles_x = (p.W + ch::TW - 1) / ch::TW, tiles_y = (p.H + ch::TH - 1) / ch::TH;
  static LdsAttr lds_attr[5];   // per device (a second GPU used from this process needs the attribute as well)
  const void* kern = form == 1 ? (const void*)conv_halo4_kernel<0> : form == 4 ? (const void*)conv_halo4p_kernel : (const void*)conv_halo_kernel;
#ifdef RTV_LAB
  if (form == 2) kern = (const void*)conv_halo4_kernel<1>;
  if (form == 3) kern = (const void*)conv_halo4_kernel<2>;
#endif
  const int lds = form == 4 ? ch::LDS_BYTES + 4 * 32 * 192 : ch::LDS_BYTES;   // persistent form: + four wave-private 32 x 96 epilogue images
  if (int st = ensure_dynamic_lds(kern, lds, &lds_attr[form], "conv")) return st;
  ProfScope prof(PROF_CONV, stream, 2.0 * p.M * (double)p.Cout * 9 * p.kt * p.Cin);
  const dim3 grid(p.T * tiles_y * tiles_x * (p.Cout / 96));
  note_kernel(form == 4 ? DK_CONV_HALO4P : form == 0 ? DK_CONV_HALO : DK_CONV_HALO4);
  if (form == 4) {   // one workgroup per CU, rounded down to whole XCD rows so that a virtual id keeps its XCD (id % 8)
    const int cus = device_num_cus() > 0 ? device_num_cus() : 256;
    const int nwg = (int)grid.x < cus ? (int)grid.x : cus / 8 * 8;
    hipLaunchKernelGGL(conv_halo4p_kernel, dim3(nwg), dim3(ch4::THREADS4), lds, stream, p, tiles_x, tiles_y);
    return check_launch("conv_halo");
  }
  if (form == 1) hipLaunchKernelGGL(conv_halo4_kernel<0>, grid, dim3(ch4::THREADS4), ch::LDS_BYTES, stream, p, tiles_x, tiles_y);
#ifdef RTV_LAB
  else if (form == 2) hipLaunchKernelGGL(conv_halo4_kernel<1>, grid, dim3(ch4::THREADS4), ch::LDS_BYTES, stream, p, tiles_x, tiles_y);
  else if (form == 3) hipLaunchKernelGGL(conv_halo4_kernel<2>, grid, dim3(ch4::THREADS4), ch::LDS_BYTES, stream, p, tiles_x, tiles_y);
#endif
  else hipLaunchKernelGGL(conv_halo_kernel, grid, dim3(ch::THREADS), ch::LDS_BYTES, stream, p, tiles_x, tiles_y);
  return check_launch("conv_halo");
}

template <int BM, int BN, int BK, int WM, int WN>
static int launch_conv_cfg(ConvParams p, hipStream_t stream) {
  typedef TileCfg<BM, BN, BK, WM, WN> Cfg;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  const int lds = 3 * Cfg::STAGE_BYTES;
  static_assert(Cfg::NW * Cfg::TM * 32 * Cfg::TN * 64 <= 3 * Cfg::STAGE_BYTES, "epilogue image must fit the stage buffers");
  auto kern = conv_igemm_kernel<BM, BN, BK, WM, WN>;
  static LdsAttr lds_attr;   // per device (a second GPU used from this process needs the attribute as well)
  if (int st = ensure_dynamic_lds((const void*)kern, lds, &lds_attr, "conv")) return st;
  ProfScope prof(PROF_CONV, stream, 2.0 * p.M * (double)p.Cout * p.kt * p.kh * p.kw * p.Cin);
  note_kernel(DK_CONV_IGEMM);
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(Cfg::NT), lds, stream, p);
  return check_launch("conv");
}

// rtv_conv_set_halo: 0 = conv_igemm_kernel everywhere (A/B, tests); 1 = default = 6: the persistent one-wave-per-SIMD halo kernel;
// 2 / 3 = the one- / two-waves-per-SIMD form with one workgroup per tile (lab build: 4, 5 = timing-only forms of 2)
static std::atomic<int> g_conv_halo{1};

int launch_conv(const ConvParams& p, hipStream_t stream) {
  if (p.M <= 0) return 0;
  if (p.Cin % 32) return set_error(-1, "conv: Cin must be a multiple of 32 (pad channels)");
  if (p.Cout % 8) return set_error(-1, "conv: Cout must be a multiple of 8 (pad filters)");
  if (p.out_ld % 4 || (p.residual && p.res_ld % 4)) return set_error(-1, "conv: channel strides must be multiples of 4");
  if (p.n_split && (p.n_split % 4 || p.Cout != 2 * p.n_split)) return set_error(-1, "conv: bad n_split");
  // 3x3x3 stride-1 convs at 96 / 192 / 384 channels: the halo-tile kernel (unless the caller marks the layer RTV_CONV_GATHER).  The choice depends on the layer (channels, taps, layout)
  // only, never on T / H / W: a row-sharded decode must run every layer on the kernel the unsharded one uses (bit parity).
  // r05: also the 3x3 conv behind the nearest-2x upsampling of a Resample (vae_block3.py:19-28, :69-72: 384 -> 192, 384 -> 192, 192 -> 96),
  // the upsampling folded into the halo gather, with the row windows of the sharded decode.
  const bool halo_common = g_conv_halo && !p.gather && p.kh == 3 && p.kw == 3 && p.sy == 1 && p.st == 1 && !p.n_split && p.pad_h == 1 &&
                           p.pad_w == 1 && p.Cin % 32 == 0 && p.Cin <= 384 && p.Cout % 96 == 0 && p.Cout <= 384 &&
                           !((p.out_ld | (p.residual ? p.res_ld : 0)) & 7) && !(((uintptr_t)p.out | (uintptr_t)p.residual) & 15) &&
                           (size_t)(p.T + 2) * p.inH * p.inW * p.Cin < 0x7fffffffull;
  const bool halo3 = halo_common && p.kt == 3 && !p.ups && p.y_out0 == 0 && p.y_in0 == 0 && p.in_rows == p.inH && p.limH == p.inH &&
                     p.limW == p.inW;
  const bool halo_up = halo_common && p.kt == 1 && p.ups && p.limW == p.W;
  // r06: the LAST time tap of a 3x3x3 layer run as a 1x3x3 convolution over the newest slice (w_taps = 27: the weight is used in
  // place): the same layer on the same kernel, one (time slice) group per channel chunk instead of three
  const bool halo1 = halo_common && p.kt == 1 && !p.ups && p.w_taps == 27 && p.y_out0 == 0 && p.y_in0 == 0 && p.in_rows == p.inH &&
                     p.limH == p.inH && p.limW == p.inW;
  const bool halo = halo3 || halo_up || halo1;
  if (p.norm_gamma && !(halo && p.Cout == 96 && !p.residual && !((uintptr_t)p.norm_gamma & 7)))
    return set_error(-1, "conv: the fused RMS_norm + SiLU epilogue needs a halo-kernel layer with 96 filters and no residual");
  if (halo) {
    // default: the persistent one-wave-per-SIMD form on every layer a halo kernel takes (-8..11 % against the two-waves form on
    // all thirteen decoder layer shapes, profiles/r05_conv_halo4.log).  The choice is by LAYER only - never by T / H / W (see above) -
    // and the three forms are bit-identical anyway.
    const int mode = g_conv_halo;
    int form = mode == 1 || mode == 6 ? 4 : mode == 2 ? 1 : mode == 3 ? 0 : mode - 2;
    // the one-wave forms address the input through a buffer descriptor with 32-bit BYTE offsets and an out-of-image sentinel at 2^31
    if (form != 0 && (size_t)(p.T + 2) * p.inH * p.inW * p.Cin * 2 >= 0x80000000ull) form = 0;
    return launch_conv_halo(p, stream, form);
  }
  if (p.Cout % 96 == 0 && p.Cout % 128 != 0) return launch_conv_cfg<128, 96, 32, 2, 1>(p, stream);
  if (p.Cout <= 32) return launch_conv_cfg<128, 32, 32, 2, 1>(p, stream);
  return launch_conv_cfg<128, 128, 32, 2, 2>(p, stream);
}

}  // namespace rtv

using namespace rtv;

extern "C" int rtv_conv_set_halo(int on) {
#ifdef RTV_LAB
  g_conv_halo = on < 0 || on > 6 ? 1 : on;
#else
  g_conv_halo = on < 0 || (on > 3 && on != 6) ? 1 : on;
#endif
  return 0;
}

/* Row-window variant (spatially sharded decode): the output buffer holds image rows [y_out0, y_out0 + H), the input
 * buffer `in_rows` rows starting at image row y_in0 (input resolution), the image has img_rows rows at OUTPUT resolution.
 * Only meaningful with RTV_CONV_UPSAMPLE2X (the stage transition); other modes treat the window as the image. */
extern "C" int rtv_conv_cl_win(const void* in, const void* w, const void* bias, const void* residual, int res_ld,
                               void* out, int out_ld, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                               int resample, int n_split, const void* zeros, int y_out0, int y_in0, int in_rows,
                               int img_rows, rtv_stream_t stream);

/* Standalone C entry (used by the tests): one convolution launch. */
extern "C" int rtv_conv_cl(const void* in, const void* w, const void* bias, const void* residual, int res_ld,
                           void* out, int out_ld, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                           int resample, int n_split, const void* zeros, rtv_stream_t stream) {
  return rtv_conv_cl_win(in, w, bias, residual, res_ld, out, out_ld, T, H, W, Cin, Cout, kt, kh, kw, resample, n_split,
                         zeros, 0, 0, -1, -1, stream);
}

static std::atomic<bool> g_conv_fuse_norm{true};   // rtv_conv_set_fuse_norm(0): A/B against the separate rmsnorm_silu pass (tests)
extern "C" int rtv_conv_set_fuse_norm(int on) {
  g_conv_fuse_norm = on != 0;
  return 0;
}

/* 3x3x3 causal conv (input = the concat buffer, as rtv_conv_cl with kt = kh = kw = 3) whose epilogue applies the RMS_norm * gamma
 * + SiLU that follows it in a ResidualBlock (wan/modules/vae.py:186-192): out = SiLU(RMS_norm(conv + bias) * gamma).
 * Returns 1 (nothing launched, no error) when the layer is not one the halo-tile kernel takes with all 96 channels of a pixel in
 * one workgroup - the caller then runs the conv and rtv_rmsnorm_silu_cl separately. */
extern "C" int rtv_conv3_norm_silu_cl(const void* in, const void* w, const void* bias, const void* gamma, void* out, int out_ld,
                                      int T, int H, int W, int Cin, int Cout, int flags, const void* zeros, rtv_stream_t stream) {
  if (!in || !w || !out || !zeros || !gamma) return set_error(-1, "conv: null pointer");
  if (!g_conv_fuse_norm || !g_conv_halo || (flags & RTV_CONV_GATHER) || Cout != 96 || Cin % 32 || Cin > 384 || (out_ld & 7) ||
      ((uintptr_t)out & 15) || ((uintptr_t)gamma & 7) || (size_t)(T + 2) * H * W * Cin >= 0x7fffffffull)
    return 1;
  ConvParams p;
  p.in = (const uint16_t*)in;
  p.w = (const uint16_t*)w;
  p.out = (uint16_t*)out;
  p.bias = (const uint16_t*)bias;
  p.residual = nullptr;
  p.zeros = (const uint16_t*)zeros;
  p.out_ld = out_ld;
  p.res_ld = 0;
  p.T = T;
  p.H = H;
  p.W = W;
  p.inH = H;
  p.inW = W;
  p.sy = p.st = 1;
  p.pad_h = p.pad_w = 1;
  p.limH = H;
  p.limW = W;
  p.y_out0 = p.y_in0 = 0;
  p.in_rows = H;
  p.gather = 0;
  p.norm_gamma = (const uint16_t*)gamma;
  p.Cin = Cin;
  p.Cout = Cout;
  p.kt = p.kh = p.kw = 3;
  p.w_taps = 27;
  p.ups = 0;
  p.n_split = 0;
  p.M = T * H * W;
  p.tiles_m = p.tiles_n = 0;
  return launch_conv(p, (hipStream_t)stream);
}

/* The 3x3x3 causal convolution of ONE new frame over two all-zero cached slices (a fresh stream: vae.py:17-36 pads zeros in front of
 * the first chunk) = the 1x3x3 convolution of that frame with the LAST time tap of the weight: taps 0-17 multiply zeros.  `frame` =
 * the new slice [H][W][Cin] (slice 2 of the concat buffer), `w3` = the 3x3x3 weight [Cout][27][Cin] used in place.  Same kernel per
 * layer as rtv_conv_cl / rtv_conv3_norm_silu_cl (gamma != null: the fused RMS_norm + SiLU epilogue, 96 filters on the halo kernel),
 * the surviving products in the same order: bit-identical with the full launch, a third of its matrix work.  Returns 1 (nothing
 * launched) where a fused epilogue is asked for on a layer that has none. */
int rtv::conv3_last_tap(const void* frame, const void* w3, const void* bias, const void* gamma, const void* residual, int res_ld,
                        void* out, int out_ld, int H, int W, int Cin, int Cout, int flags, const void* zeros, hipStream_t stream) {
  if (!frame || !w3 || !out || !zeros) return set_error(-1, "conv: null pointer");
  const int gather = (flags & RTV_CONV_GATHER) ? 1 : 0;
  if (gamma && (!g_conv_fuse_norm || !g_conv_halo || gather || Cout != 96 || Cin % 32 || Cin > 384 || (out_ld & 7) || residual ||
                ((uintptr_t)out & 15) || ((uintptr_t)gamma & 7)))
    return 1;
  ConvParams p;
  p.in = (const uint16_t*)frame;
  p.w = (const uint16_t*)w3 + (size_t)18 * Cin;
  p.out = (uint16_t*)out;
  p.bias = (const uint16_t*)bias;
  p.residual = (const uint16_t*)residual;
  p.zeros = (const uint16_t*)zeros;
  p.out_ld = out_ld;
  p.res_ld = res_ld;
  p.T = 1;
  p.H = H;
  p.W = W;
  p.inH = H;
  p.inW = W;
  p.sy = p.st = 1;
  p.pad_h = p.pad_w = 1;
  p.limH = H;
  p.limW = W;
  p.y_out0 = p.y_in0 = 0;
  p.in_rows = H;
  p.gather = gather;
  p.norm_gamma = (const uint16_t*)gamma;
  p.Cin = Cin;
  p.Cout = Cout;
  p.kt = 1;
  p.kh = p.kw = 3;
  p.w_taps = 27;
  p.ups = 0;
  p.n_split = 0;
  p.M = H * W;
  p.tiles_m = p.tiles_n = 0;
  return launch_conv(p, stream);
}

extern "C" int rtv_conv_cl_win(const void* in, const void* w, const void* bias, const void* residual, int res_ld,
                               void* out, int out_ld, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                               int resample, int n_split, const void* zeros, int y_out0, int y_in0, int in_rows,
                               int img_rows, rtv_stream_t stream) {
  if (!in || !w || !out || !zeros) return set_error(-1, "conv: null pointer");
  const int gather = (resample & RTV_CONV_GATHER) ? 1 : 0;
  resample &= ~RTV_CONV_GATHER;
  if ((kt != 1 && kt != 3) || (kh != 1 && kh != 3) || kh != kw) return set_error(-1, "conv: kernel must be 1 or 3 per axis");
  if (resample < 0 || resample > 3) return set_error(-1, "conv: resample must be 0..3");
  const int ups = resample == RTV_CONV_UPSAMPLE2X;
  if (ups && ((W & 1) || (in_rows < 0 && (H & 1)))) return set_error(-1, "conv: upsampled output dims must be even");
  if (resample == RTV_CONV_DOWN2X && (kt != 1 || kh != 3)) return set_error(-1, "conv: the stride-2 downsample is a 1x3x3 conv");
  if (resample == RTV_CONV_TIME_DOWN2X && (kt != 3 || kh != 1)) return set_error(-1, "conv: the stride-2 time conv is 3x1x1");
  if (resample != RTV_CONV_NONE && resample != RTV_CONV_UPSAMPLE2X && n_split) return set_error(-1, "conv: n_split with a strided conv");
  ConvParams p;
  p.in = (const uint16_t*)in;
  p.w = (const uint16_t*)w;
  p.out = (uint16_t*)out;
  p.bias = (const uint16_t*)bias;
  p.residual = (const uint16_t*)residual;
  p.zeros = (const uint16_t*)zeros;
  p.out_ld = out_ld;
  p.res_ld = res_ld;
  p.T = T;
  p.H = H;
  p.W = W;
  p.inH = ups ? H / 2 : (resample == RTV_CONV_DOWN2X ? 2 * H : H);
  p.inW = ups ? W / 2 : (resample == RTV_CONV_DOWN2X ? 2 * W : W);
  p.sy = resample == RTV_CONV_DOWN2X ? 2 : 1;
  p.st = resample == RTV_CONV_TIME_DOWN2X ? 2 : 1;
  p.pad_h = resample == RTV_CONV_DOWN2X ? 0 : kh >> 1;
  p.pad_w = resample == RTV_CONV_DOWN2X ? 0 : kw >> 1;
  p.limH = ups ? H : p.inH;
  p.limW = ups ? W : p.inW;
  p.y_out0 = p.y_in0 = 0;
  p.in_rows = p.inH;
  if (in_rows >= 0) {  // row-window call
    if (!ups) return set_error(-1, "conv: row windows are for the upsampling stage transition");
    if (in_rows <= 0 || img_rows <= 0 || y_out0 < 0 || y_in0 < 0 || y_out0 + H > img_rows)
      return set_error(-1, "conv: bad row window");
    {  // every tap row of the window (upsampled rows y_out0 - 1 .. y_out0 + H, clipped to the image) must lie inside the
       // supplied source rows [y_in0, y_in0 + in_rows): the kernel does not clamp source rows
      const int lo_up = y_out0 > 0 ? y_out0 - (kh >> 1) : 0;
      const int hi_up = y_out0 + H - 1 + (kh >> 1) < img_rows ? y_out0 + H - 1 + (kh >> 1) : img_rows - 1;
      if ((lo_up >> 1) < y_in0 || (hi_up >> 1) >= y_in0 + in_rows)
        return set_error(-1, "conv: the row window's taps fall outside the supplied input rows");
    }
    p.y_out0 = y_out0;
    p.y_in0 = y_in0;
    p.in_rows = in_rows;
    p.inH = in_rows;      // slice stride of the input buffer
    p.limH = img_rows;
  }
  p.gather = gather;
  p.norm_gamma = nullptr;
  p.Cin = Cin;
  p.Cout = Cout;
  p.kt = kt;
  p.kh = kh;
  p.kw = kw;
  p.w_taps = kt * kh * kw;
  p.ups = ups ? 1 : 0;
  p.n_split = n_split;
  p.M = T * H * W;
  p.tiles_m = p.tiles_n = 0;
  return launch_conv(p, (hipStream_t)stream);
}
