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
  int ups;          // 1: input is read through a nearest 2x upsampling
  int sy, st;       // spatial / temporal stride of the output grid over the input (1 or 2)
  int pad_h, pad_w; // low-side zero padding (kh/2 for 'same' convs, 0 for the stride-2 downsample: its ZeroPad2d is high-side)
  int limH, limW;   // taps outside [0,limH) x [0,limW) read zeros (input dims, or output dims when ups)
  int y_out0, y_in0; // row windows (spatially sharded decode): output row py is image row y_out0 + py, input buffer row 0
                     // is image row y_in0 (in input resolution); limH is then the IMAGE height
  int in_rows;       // rows held by the input buffer
  int n_split;      // >0: output channel n -> frame 2t + n / n_split, channel n % n_split
  int M;            // T*H*W
  int tiles_m, tiles_n;
};

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(ConvParams p) {
  typedef TileCfg<BM, BN, BK, WM, WN> Cfg;
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
  // per-lane pixel coordinates of the A rows this lane stages
  int pt[Cfg::A_INST], py[Cfg::A_INST], px[Cfg::A_INST], a_chunk[Cfg::A_INST];
#pragma unroll
  for (int i = 0; i < Cfg::A_INST; ++i) {
    int row = (wave * Cfg::A_INST + i) * Cfg::RPI + rsub;
    int m = min(m0 + row, p.M - 1);
    pt[i] = m / HW;
    int rem = m - pt[i] * HW;
    py[i] = rem / p.W;
    px[i] = rem - py[i] * p.W;
    a_chunk[i] = Cfg::swz(row, cpos) * 8;
  }
  uint32_t b_off[Cfg::B_INST];
  const int Ktot = p.kt * p.kh * p.kw * p.Cin;
#pragma unroll
  for (int i = 0; i < Cfg::B_INST; ++i) {
    int row = (wave * Cfg::B_INST + i) * Cfg::RPI + rsub;
    int gn = min(n0 + row, p.Cout - 1);
    b_off[i] = (uint32_t)gn * (uint32_t)Ktot + Cfg::swz(row, cpos) * 8;
  }
  const int cpk = p.Cin / BK;  // K-steps per tap
  const int nk = p.kt * p.kh * p.kw * cpk;

  auto stage = [&](int ks, int buf) {
    char* sA = smem + buf * Cfg::STAGE_BYTES;
    char* sB = sA + Cfg::A_BYTES;
    const int tap = ks / cpk;
    const int c0 = (ks - tap * cpk) * BK;
    const int dt = tap / (p.kh * p.kw);
    const int r2 = tap - dt * (p.kh * p.kw);
    const int dy = r2 / p.kw - p.pad_h;
    const int dx = r2 - (r2 / p.kw) * p.kw - p.pad_w;
#pragma unroll
    for (int i = 0; i < Cfg::A_INST; ++i) {
      const int yy = p.y_out0 + py[i] * p.sy + dy, xx = px[i] * p.sy + dx;
      const bool ok = (yy >= 0) & (yy < p.limH) & (xx >= 0) & (xx < p.limW);
      const int ti = pt[i] * p.st + dt;
      const int sy_ = min(max((yy >> p.ups) - p.y_in0, 0), p.in_rows - 1);
      const size_t off = ((size_t)(ti * p.inH + sy_) * p.inW + (xx >> p.ups)) * p.Cin + c0 + a_chunk[i];
      const uint16_t* src = ok ? p.in + off : p.zeros;
      dma16(src, sA + (wave * Cfg::A_INST + i) * 1024);
    }
    const uint16_t* Wk = p.w + (size_t)ks * BK;
#pragma unroll
    for (int i = 0; i < Cfg::B_INST; ++i) dma16(Wk + b_off[i], sB + (wave * Cfg::B_INST + i) * 1024);
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

  stage(0, 0);
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    __syncthreads();
    if (ks + 1 < nk) stage(ks + 1, buf ^ 1);
    const char* sA = smem + buf * Cfg::STAGE_BYTES;
    mma_stage<true, Cfg, BK>(sA, sA + Cfg::A_BYTES, a_row0, b_row0, lane, acc);
  }

  // ---- epilogue: bias (+ residual), optional channel-half -> frame scatter
  const int l31 = lane & 31, g = lane >> 5;
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

template <int BM, int BN, int BK, int WM, int WN>
static int launch_conv_cfg(ConvParams p, hipStream_t stream) {
  typedef TileCfg<BM, BN, BK, WM, WN> Cfg;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.Cout + BN - 1) / BN;
  const int lds = 2 * Cfg::STAGE_BYTES;
  auto kern = conv_igemm_kernel<BM, BN, BK, WM, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(e, "conv: hipFuncSetAttribute");
    attr_set = true;
  }
  ProfScope prof(PROF_CONV, stream, 2.0 * p.M * (double)p.Cout * p.kt * p.kh * p.kw * p.Cin);
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n), dim3(Cfg::NT), lds, stream, p);
  return check_launch("conv");
}

int launch_conv(const ConvParams& p, hipStream_t stream) {
  if (p.M <= 0) return 0;
  if (p.Cin % 32) return set_error(-1, "conv: Cin must be a multiple of 32 (pad channels)");
  if (p.Cout % 8) return set_error(-1, "conv: Cout must be a multiple of 8 (pad filters)");
  if (p.out_ld % 4 || (p.residual && p.res_ld % 4)) return set_error(-1, "conv: channel strides must be multiples of 4");
  if (p.n_split && (p.n_split % 4 || p.Cout != 2 * p.n_split)) return set_error(-1, "conv: bad n_split");
  if (p.Cout % 96 == 0 && p.Cout % 128 != 0) return launch_conv_cfg<128, 96, 32, 2, 1>(p, stream);
  if (p.Cout <= 32) return launch_conv_cfg<128, 32, 32, 2, 1>(p, stream);
  return launch_conv_cfg<128, 128, 32, 2, 2>(p, stream);
}

}  // namespace rtv

using namespace rtv;

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

extern "C" int rtv_conv_cl_win(const void* in, const void* w, const void* bias, const void* residual, int res_ld,
                               void* out, int out_ld, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                               int resample, int n_split, const void* zeros, int y_out0, int y_in0, int in_rows,
                               int img_rows, rtv_stream_t stream) {
  if (!in || !w || !out || !zeros) return set_error(-1, "conv: null pointer");
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
    p.y_out0 = y_out0;
    p.y_in0 = y_in0;
    p.in_rows = in_rows;
    p.inH = in_rows;      // slice stride of the input buffer
    p.limH = img_rows;
  }
  p.Cin = Cin;
  p.Cout = Cout;
  p.kt = kt;
  p.kh = kh;
  p.kw = kw;
  p.ups = ups ? 1 : 0;
  p.n_split = n_split;
  p.M = T * H * W;
  p.tiles_m = p.tiles_n = 0;
  return launch_conv(p, (hipStream_t)stream);
}
