// Streaming VAE decoder orchestrator + its HBM-bound kernels (fp16, channels-last).
//
// One call = VAEDecoderWrapper.forward (demo_utils/vae_block3.py:195-230): per latent frame run
// VAEDecoder3d.forward (:386-443) = conv1 -> middle(Res, Attn, Res) -> 4 up stages (3 Res + Resample)
// -> RMS_norm, SiLU, conv(96->3), with the 32 two-slice feature caches kept in a caller-owned arena.
// Convolutions: vae_conv.hip.  1x1 convs / attention projections: the MFMA GEMM (gemm.hip, fp16).
#include "gemm_core.h"
#include "rtv_internal.h"

namespace rtv {

struct ConvParams;
int launch_conv(const ConvParams& p, hipStream_t stream);

// ------------------------------------------------------------------ RMS_norm (+SiLU), channels-last
// F.normalize(x, dim=C) * sqrt(C) * gamma  (wan/modules/vae.py:39-54) followed by nn.SiLU.
// 256 threads x 3 chunks of 8 channels = 768 chunks per block = 768/(C/8) whole pixels; every global access stays a
// coalesced 16 B.  The per-pixel sum of squares is reduced through LDS in a FIXED order (one thread per pixel adds the
// pixel's chunk partials in sequence), so the result does not depend on scheduling: the spatially sharded decode must
// reproduce the unsharded one bit for bit.
constexpr int RN_CHUNKS = 3;
__global__ __launch_bounds__(256) void rmsnorm_silu_cl_kernel(const f16_t* __restrict__ x, f16_t* __restrict__ out,
                                                             const f16_t* __restrict__ gamma, int C,
                                                             int64_t npix, int apply_silu) {
  __shared__ float ssq[64];
  __shared__ float part[256 * RN_CHUNKS];
  const int G = C >> 3;               // chunks per pixel (12 / 24 / 48)
  const int ppb = (256 * RN_CHUNKS) / G;  // pixels per block
  const int64_t pix0 = (int64_t)blockIdx.x * ppb;
  float v[RN_CHUNKS][8];
  int lp[RN_CHUNKS];
#pragma unroll
  for (int i = 0; i < RN_CHUNKS; ++i) {
    const int c = threadIdx.x + i * 256;  // chunk id inside the block, < 768
    lp[i] = c / G;
    const int64_t pix = pix0 + lp[i];
    float s = 0.f;
    if (pix < npix) {
      u32x4 raw = *(const u32x4*)(x + pix * C + (c - lp[i] * G) * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t u = raw[j];
        unpack_f16x2(u, v[i][2 * j], v[i][2 * j + 1]);
        s += v[i][2 * j] * v[i][2 * j] + v[i][2 * j + 1] * v[i][2 * j + 1];
      }
    }
    part[c] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < ppb) {
    float t = 0.f;
    for (int k = 0; k < G; ++k) t += part[threadIdx.x * G + k];
    ssq[threadIdx.x] = t;
  }
  __syncthreads();
  const float sqrtC = sqrtf((float)C);
#pragma unroll
  for (int i = 0; i < RN_CHUNKS; ++i) {
    const int c = threadIdx.x + i * 256;
    const int64_t pix = pix0 + lp[i];
    if (pix < npix) {
      const int ch = (c - lp[i] * G) * 8;
      const float inv = sqrtC / fmaxf(sqrtf(ssq[lp[i]]), 1e-12f);
      u32x4 graw = *(const u32x4*)(gamma + ch);
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t gu = graw[j];
        float g0, g1;
        unpack_f16x2(gu, g0, g1);
        float a = v[i][2 * j] * inv * g0;
        float b = v[i][2 * j + 1] * inv * g1;
        if (apply_silu) {
          a = silu(a);
          b = silu(b);
        }
        o[j] = pack_f16x2(a, b);
      }
      *(u32x4*)(out + pix * C + ch) = o;
    }
  }
}

// ------------------------------------------------------------------ row softmax (mid-block attention)
// P[r][0..n) = softmax(S[r][0..n)) in f32, written f16 with zero padding up to ldp columns.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const f16_t* __restrict__ s, int lds_, f16_t* __restrict__ p,
                                                          int ldp, int n) {
  __shared__ float red[8];
  const int row = blockIdx.x;
  const f16_t* sr = s + (size_t)row * lds_;
  f16_t* pr = p + (size_t)row * ldp;
  float mx = -INFINITY;
  for (int c = threadIdx.x * 8; c < n; c += 256 * 8) {
    u32x4 raw = *(const u32x4*)(sr + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t u = raw[j];
      float a, b;
      unpack_f16x2(u, a, b);
      mx = fmaxf(mx, fmaxf(a, b));
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int c = threadIdx.x * 8; c < n; c += 256 * 8) {
    u32x4 raw = *(const u32x4*)(sr + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t u = raw[j];
      float a, b;
      unpack_f16x2(u, a, b);
      sum += __expf(a - mx) + __expf(b - mx);
    }
  }
  sum = wave_sum(sum);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = threadIdx.x * 8; c < ldp; c += 256 * 8) {
    u32x4 o = {0u, 0u, 0u, 0u};
    if (c < n) {
      u32x4 raw = *(const u32x4*)(sr + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t u = raw[j];
        float a, b;
        unpack_f16x2(u, a, b);
        o[j] = pack_f16x2(__expf(a - mx) * inv, __expf(b - mx) * inv);
      }
    }
    *(u32x4*)(pr + c) = o;
  }
}

// ------------------------------------------------------------------ small glue kernels
// Temporal-upsampling cache update for a single new slice (vae_block3.py:56-62):
//   cache <- [ where(old_cache_last == 0, 0, x), x ]      buf = [c0 | c1 | x] (3 slices)
__global__ void upsample_cache_t1_kernel(f16_t* buf, int64_t slice) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slice; i += (int64_t)gridDim.x * blockDim.x) {
    f16_t c1 = buf[slice + i], x = buf[2 * slice + i];
    buf[i] = ((c1 & 0x7fff) == 0) ? (f16_t)0 : x;
    buf[slice + i] = x;
  }
}

// z[T][16][h][w] (one latent frame t) -> conv2(z * std + mean) -> channels-last [h][w][32] (16 real + 16 zero)
__global__ void vae_prep_kernel(const f16_t* __restrict__ z, int t, int hw, const float* __restrict__ mean,
                                const float* __restrict__ stdv, const float* __restrict__ w2 /*[16][16]*/,
                                const float* __restrict__ b2, f16_t* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= hw) return;
  float zin[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    // reference order in fp16: z / (1/std) + mean   (vae_block3.py:206-211)
    float zz = f16_to_f32(z[((size_t)t * 16 + c) * hw + p]);
    float inv = round_f16(1.0f / round_f16(stdv[c]));
    zin[c] = round_f16(round_f16(zz / inv) + round_f16(mean[c]));
  }
  u32x4 o[4];
#pragma unroll
  for (int co = 0; co < 16; co += 2) {
    float a = b2[co], b = b2[co + 1];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      a += w2[co * 16 + c] * zin[c];
      b += w2[(co + 1) * 16 + c] * zin[c];
    }
    o[co >> 3][(co & 7) >> 1] = pack_f16x2(a, b);
  }
  o[2] = u32x4{0u, 0u, 0u, 0u};
  o[3] = u32x4{0u, 0u, 0u, 0u};
  u32x4* dst = (u32x4*)(out + (size_t)p * 32);
  dst[0] = o[0];
  dst[1] = o[1];
  dst[2] = o[2];
  dst[3] = o[3];
}

// head output [T][Hin][W][8] f16 (3 real channels; rows skip .. skip + rows of it) -> pixels f32 [T][3][rows][W] in [-1, 1]
__global__ void vae_final_kernel(const f16_t* __restrict__ in, float* __restrict__ out, int T, int64_t hw, int64_t in_hw,
                                 int64_t skip_px) {
  const int64_t total = (int64_t)T * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / hw, p = i - t * hw;
    u32x2 raw = *(const u32x2*)(in + (t * in_hw + skip_px + p) * 8);
    const uint32_t u0 = raw[0], u1 = raw[1];
    float c0, c1, c2, c3;
    unpack_f16x2(u0, c0, c1);
    unpack_f16x2(u1, c2, c3);
    float r = fminf(fmaxf(c0, -1.f), 1.f), g = fminf(fmaxf(c1, -1.f), 1.f), bl = fminf(fmaxf(c2, -1.f), 1.f);
    out[(t * 3 + 0) * hw + p] = r;
    out[(t * 3 + 1) * hw + p] = g;
    out[(t * 3 + 2) * hw + p] = bl;
  }
}

}  // namespace rtv

using namespace rtv;

// ConvParams is defined in vae_conv.hip; the orchestrator goes through the C entry rtv_conv_cl.
extern "C" int rtv_conv_cl(const void* in, const void* w, const void* bias, const void* residual, int res_ld,
                           void* out, int out_ld, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                           int resample, int n_split, const void* zeros, rtv_stream_t stream);

#define RTV_TRY(expr)       \
  do {                      \
    int _s = (expr);        \
    if (_s != 0) return _s; \
  } while (0)

extern "C" int rtv_rmsnorm_silu_cl(const void* x, void* out, const void* gamma, int C, int64_t npix, int apply_silu,
                                   rtv_stream_t stream) {
  if (npix <= 0) return 0;
  if (C % 8 || 768 % (C / 8) || (768 / (C / 8)) > 64) return set_error(-1, "rmsnorm_silu_cl: C must be 96, 192, 384 (or another divisor layout of 768 chunks)");
  const int ppb = 768 / (C / 8);
  ProfScope prof(PROF_LN, (hipStream_t)stream, 2.0 * npix * C * 2);
  hipLaunchKernelGGL(rmsnorm_silu_cl_kernel, dim3((unsigned)((npix + ppb - 1) / ppb)), dim3(256), 0,
                     (hipStream_t)stream, (const f16_t*)x, (f16_t*)out, (const f16_t*)gamma, C, npix, apply_silu);
  return check_launch("rmsnorm_silu_cl");
}

extern "C" int rtv_softmax_rows(const void* s, int lds_, void* p, int ldp, int rows, int n, rtv_stream_t stream) {
  if (rows <= 0) return 0;
  if (n % 8 || lds_ % 8 || ldp % 8 || ldp < n) return set_error(-1, "softmax_rows: n / strides must be multiples of 8");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, (const f16_t*)s, lds_,
                     (f16_t*)p, ldp, n);
  return check_launch("softmax_rows");
}

// ------------------------------------------------------------------ arena layout
namespace {

struct Stage {
  int H, W;
};

// Row windows of a spatially sharded decode (one rank produces pixel rows [r0, r1)): stage s (resolution h << s) only
// holds image rows [a[s], b[s]).  Every conv treats its window as the image (zero padding at the window edges), which
// is wrong by one more row per 3x3 conv at edges that are not image borders, so the windows carry that many halo rows:
// 7 at full resolution (6 ResidualBlock convs + head), 6 per earlier stage plus the one row the stage-transition conv
// reads beyond its output.  Stage 0 (conv1, the global mid-block attention, 4 % of the FLOPs) is never windowed.
struct RowPlan {
  int a[4], b[4];
  int r0, r1;
};

static int make_plan(int h, int r0, int r1, RowPlan* P) {
  const int H3 = h << 3;
  if (r0 < 0 || r1 > H3 || r0 >= r1) return set_error(-1, "vae_decode: bad pixel row range");
  P->r0 = r0;
  P->r1 = r1;
  P->a[3] = r0 - 7 > 0 ? r0 - 7 : 0;
  P->b[3] = r1 + 7 < H3 ? r1 + 7 : H3;
  for (int s = 2; s >= 1; --s) {
    const int Hs = h << s;
    int lo = (P->a[s + 1] - 1) >> 1, hi = (P->b[s + 1] >> 1) + 1;  // source rows of the upsampled taps a-1 .. b
    if (lo < 0) lo = 0;
    if (hi > Hs) hi = Hs;
    P->a[s] = lo - 6 > 0 ? lo - 6 : 0;
    P->b[s] = hi + 6 < Hs ? hi + 6 : Hs;
  }
  P->a[0] = 0;
  P->b[0] = h;
  return 0;
}

struct VaeLayout {
  // persistent: 32 concat buffers (2 cache slices + Tmax new slices each)
  size_t cat_off[32];
  int cat_C[32], cat_stage[32], cat_T[32];
  // r06: a concat buffer may hold MORE than 2 + Tmax slices (cat_cap): a call then slides its [cache | new] window forward by the
  // frame's T slices per latent frame instead of copying the last two slices to the front after every frame (roll_cache), and
  // copies once when the window reaches the end / at the end of the call.  cat_slice = bytes per [H][W][C] slice.
  int cat_cap[32];
  size_t cat_slice[32];
  size_t act_off[4];   // rotating activation buffers (max activation size)
  size_t s_off, p_off, q_off, k_off, vt_off, o_off, xn_off;  // mid-block attention scratch
  size_t head_off;     // [T][H][W][8]
  size_t zeros_off;
  size_t total;
  int ldp;
};

constexpr int SLIDE_FRAMES = 3;   // latent frames a decoder concat buffer takes before its cache slices are copied to the front

// concat-buffer table in execution order: (channels, stage, Tmax)
static void build_layout(int h, int w, const RowPlan& plan, VaeLayout* L) {
  const int C[32] = {32, 384, 384, 384, 384,            // conv1, mid0.a, mid0.b, mid2.a, mid2.b
                     384, 384, 384, 384, 384, 384, 384,   // up0 x3 (a,b), time_conv0
                     192, 384, 384, 384, 384, 384, 384,   // up1: (192->384) a, b, then 384 x4, time_conv1
                     192, 192, 192, 192, 192, 192,        // up2 x3
                     96, 96, 96, 96, 96, 96, 96};         // up3 x3, head
  const int S[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3};
  const int Tm[4] = {1, 2, 4, 4};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    return a;
  };
  for (int i = 0; i < 32; ++i) {
    const int s = S[i];
    const size_t hw = (size_t)(plan.b[s] - plan.a[s]) * (w << s);
    L->cat_C[i] = C[i];
    L->cat_stage[i] = s;
    L->cat_T[i] = Tm[s];
    // three latent frames (one block of the session loop) per copy; the two time_conv buffers (their cache rules are not a plain
    // "last two slices": vae_block3.py:56-62) keep the one-frame form
    const bool time_conv_buf = i == 11 || i == 18;
    L->cat_cap[i] = 2 + (time_conv_buf ? 1 : SLIDE_FRAMES) * Tm[s];
    L->cat_slice[i] = hw * C[i] * 2;
    L->cat_off[i] = take((size_t)L->cat_cap[i] * L->cat_slice[i]);
  }
  // largest activation: max over stages of T*H*W*C
  size_t amax = 0;
  const int actC[4] = {768, 768, 384, 192};  // widest tensor living at each stage (time_conv output counted at source res)
  for (int s = 0; s < 4; ++s) {
    size_t hw = (size_t)(plan.b[s] - plan.a[s]) * (w << s);
    size_t b = (size_t)Tm[s] * hw * actC[s] * 2;
    if (b > amax) amax = b;
    if (s < 3) {  // the stage-transition conv writes [T][rows of stage s+1][C/2]
      size_t hw1 = (size_t)(plan.b[s + 1] - plan.a[s + 1]) * (w << (s + 1));
      size_t b1 = (size_t)Tm[s + 1] * hw1 * (actC[s] / 4) * 2;
      if (b1 > amax) amax = b1;
    }
  }
  for (int i = 0; i < 4; ++i) L->act_off[i] = take(amax);
  const size_t P = (size_t)h * w;
  L->ldp = (int)((P + 63) / 64 * 64);
  L->s_off = take(P * P * 2);
  L->p_off = take(P * (size_t)L->ldp * 2);
  L->q_off = take(P * 384 * 2);
  L->k_off = take(P * 384 * 2);
  L->vt_off = take((size_t)384 * L->ldp * 2);
  L->o_off = take(P * 384 * 2);
  L->xn_off = take(P * 384 * 2);
  L->head_off = take((size_t)4 * (plan.b[3] - plan.a[3]) * (w << 3) * 8 * 2);
  L->zeros_off = take(256);
  L->total = off + 256;
}

}  // namespace

extern "C" size_t rtv_vae_arena_bytes_rows(int h, int w, int row0, int row1) {
  if (h <= 0 || w <= 0) return 0;
  RowPlan P;
  if (make_plan(h, row0, row1, &P)) return 0;
  VaeLayout L;
  build_layout(h, w, P, &L);
  return L.total;
}

extern "C" size_t rtv_vae_arena_bytes(int h, int w) { return rtv_vae_arena_bytes_rows(h, w, 0, h << 3); }

/* byte offset + slice geometry of feature-cache slot i (0..31) inside the arena: the two cached slices
 * are the first two [H][W][C] slices of the conv's concat buffer. */
extern "C" int rtv_vae_cache_slot_rows(int h, int w, int row0, int row1, int slot, size_t* offset, int* C, int* H,
                                       int* W, int* first_row) {
  if (slot < 0 || slot >= 32) return set_error(-1, "vae_cache_slot: slot out of range");
  RowPlan P;
  RTV_TRY(make_plan(h, row0, row1, &P));
  VaeLayout L;
  build_layout(h, w, P, &L);
  const int s = L.cat_stage[slot];
  *offset = L.cat_off[slot];
  *C = L.cat_C[slot];
  *H = P.b[s] - P.a[s];
  *W = w << s;
  *first_row = P.a[s];
  return 0;
}

extern "C" int rtv_vae_cache_slot(int h, int w, int slot, size_t* offset, int* C, int* H, int* W) {
  int first_row = 0;
  return rtv_vae_cache_slot_rows(h, w, 0, h << 3, slot, offset, C, H, W, &first_row);
}

namespace {

static std::atomic<bool> g_fresh_tap_skip{true};   // rtv_vae_set_fresh_tap_skip (include/rtv_hip_lab.h)

struct Ctx {
  char* arena;
  const VaeLayout* L;
  int h, wd;
  hipStream_t stream;
  int act_next;
  // encoder, first chunk of a stream (ONE frame over zero caches): the causal 3x3x3 convolutions run their last time tap only
  // (conv3_last_tap: taps 0-17 would multiply the zero slices; bit-identical, a third of the matrix work)
  bool fresh = false;
  int pos[32] = {};   // first slice of concat buffer i's current [2 cached | T new] window (slides inside a call, 0 between calls)
  uint16_t* act(int i) { return (uint16_t*)(arena + L->act_off[i & 3]); }
  uint16_t* cat(int i) { return (uint16_t*)(arena + L->cat_off[i] + (size_t)pos[i] * L->cat_slice[i]); }
  const void* zeros() { return arena + L->zeros_off; }
};

// conv over concat buffer `ci` whose slices [2, 2+T) already hold the new input; then roll the cache.
static int roll_cache(Ctx& c, int ci, int T, int H, int W, int Cin);
static int cached_conv3(Ctx& c, int ci, int T, int H, int W, int Cin, const rtv_vae_conv& cw, int Cout,
                        const void* residual, void* out, int out_ld) {
  uint16_t* buf = c.cat(ci);
  // latent-resolution layers (mid block, first stage: W == the latent width in sharded and unsharded decodes alike) stay on
  // the gather kernel - 16 halo tiles per frame cannot fill the chip (scripts/conv_bench.py: 315 vs 399 TF/s at 60 x 104)
  const int flags = W == c.wd ? RTV_CONV_GATHER : RTV_CONV_NONE;
  if (c.fresh && T == 1) {
    RTV_TRY(conv3_last_tap(buf + 2 * (size_t)H * W * Cin, cw.w, cw.b, nullptr, residual, Cout, out, out_ld, H, W, Cin, Cout, flags,
                           c.zeros(), c.stream));
  } else {
    RTV_TRY(rtv_conv_cl(buf, cw.w, cw.b, residual, Cout, out, out_ld, T, H, W, Cin, Cout, 3, 3, 3, flags, 0, c.zeros(), c.stream));
  }
  return roll_cache(c, ci, T, H, W, Cin);
}

// the causal conv's feature cache = the last two input slices (vae.py:17-36).  Where the concat buffer has room for another frame
// behind the current window, the window just SLIDES forward by T slices - the cache is where it is, nothing moves (r06: the copies
// of this function were the ~150 D2D copyBuffer launches of a block, 2.6 ms); otherwise the two slices are copied to the front.
static int cache_to_front(Ctx& c, int ci, int from) {
  char* b = c.arena + c.L->cat_off[ci];
  const size_t slice = c.L->cat_slice[ci];
  c.pos[ci] = 0;
  if (from == 0) return 0;
  if (from == 1) {   // overlapping: slice 1 -> 0, then 2 -> 1
    if (hipMemcpyAsync(b, b + slice, slice, hipMemcpyDeviceToDevice, c.stream) != hipSuccess ||
        hipMemcpyAsync(b + slice, b + 2 * slice, slice, hipMemcpyDeviceToDevice, c.stream) != hipSuccess)
      return set_error(-1, "vae: cache roll memcpy failed");
    return 0;
  }
  if (hipMemcpyAsync(b, b + (size_t)from * slice, 2 * slice, hipMemcpyDeviceToDevice, c.stream) != hipSuccess)
    return set_error(-1, "vae: cache roll memcpy failed");
  return 0;
}
static int roll_cache(Ctx& c, int ci, int T, int H, int W, int Cin) {
  if ((size_t)H * W * Cin * 2 != c.L->cat_slice[ci]) return set_error(-1, "vae: concat buffer geometry mismatch");
  const int np = c.pos[ci] + T;                                  // the cache is slices [np, np + 2) now
  if (np + 2 + c.L->cat_T[ci] <= c.L->cat_cap[ci]) {             // room for the largest next frame: slide
    c.pos[ci] = np;
    return 0;
  }
  return cache_to_front(c, ci, np);
}
// end of a call: every window back to the front (the cache slots the caller sees are the first two slices of each buffer)
static int flush_caches(Ctx& c, int n) {
  for (int i = 0; i < n; ++i)
    if (c.pos[i]) RTV_TRY(cache_to_front(c, i, c.pos[i]));
  return 0;
}

// ResidualBlock (wan/modules/vae.py:175-209): x [T][H][W][cin] -> y [T][H][W][cout]
static int res_block(Ctx& c, int ci, int T, int H, int W, int cin, int cout, const rtv_vae_res& rw,
                     const uint16_t* x, uint16_t* tmp, uint16_t* sc_buf, uint16_t* y) {
  const int64_t npix = (int64_t)T * H * W;
  const size_t sl_in = (size_t)H * W * cin, sl_out = (size_t)H * W * cout;
  RTV_TRY(rtv_rmsnorm_silu_cl(x, c.cat(ci) + 2 * sl_in, rw.gamma0, cin, npix, 1, c.stream));
  // conv_a -> RMS_norm -> SiLU: one launch where the conv kernel holds all channels of a pixel (96 filters, halo kernel), with
  // the normalised activation written straight into conv_b's concat buffer; else conv into `tmp` + the separate pass
  {
    const int flags = W == c.wd ? RTV_CONV_GATHER : RTV_CONV_NONE;
    int st = cout != 96 ? 1
             : (c.fresh && T == 1)
                 ? conv3_last_tap(c.cat(ci) + 2 * sl_in, rw.conv_a.w, rw.conv_a.b, rw.gamma3, nullptr, 0, c.cat(ci + 1) + 2 * sl_out, cout,
                                  H, W, cin, cout, flags, c.zeros(), c.stream)
                 : rtv_conv3_norm_silu_cl(c.cat(ci), rw.conv_a.w, rw.conv_a.b, rw.gamma3, c.cat(ci + 1) + 2 * sl_out, cout, T, H, W, cin,
                                          cout, flags, c.zeros(), c.stream);
    if (st < 0) return st;
    if (st == 0) {
      RTV_TRY(roll_cache(c, ci, T, H, W, cin));
    } else {
      RTV_TRY(cached_conv3(c, ci, T, H, W, cin, rw.conv_a, cout, nullptr, tmp, cout));
      RTV_TRY(rtv_rmsnorm_silu_cl(tmp, c.cat(ci + 1) + 2 * sl_out, rw.gamma3, cout, npix, 1, c.stream));
    }
  }
  const uint16_t* hres = x;
  if (rw.shortcut.w) {  // 1x1x1 conv = plain GEMM over pixels (K = 96 is not a multiple of the GEMM's 64: conv kernel)
    if (cin % 64 == 0)
      RTV_TRY(rtv_gemm(x, cin, rw.shortcut.w, cin, sc_buf, cout, (int)npix, cout, cin, rw.shortcut.b, 0, nullptr, 0, 0, 0,
                       nullptr, 0, RTV_DTYPE_F16, 0, c.stream));
    else
      RTV_TRY(rtv_conv_cl(x, rw.shortcut.w, rw.shortcut.b, nullptr, 0, sc_buf, cout, T, H, W, cin, cout, 1, 1, 1, 0, 0,
                          c.zeros(), c.stream));
    hres = sc_buf;
  }
  RTV_TRY(cached_conv3(c, ci + 1, T, H, W, cout, rw.conv_b, cout, hres, y, cout));
  return 0;
}

// AttentionBlock (wan/modules/vae.py:212-251) on one frame: x [P][384] -> y [P][384]
static int mid_attention(Ctx& c, const rtv_vae_attn& a, const uint16_t* x, uint16_t* y) {
  const int P = c.h * c.wd, C = 384, ldp = c.L->ldp;
  char* A = c.arena;
  uint16_t *S = (uint16_t*)(A + c.L->s_off), *Pm = (uint16_t*)(A + c.L->p_off), *q = (uint16_t*)(A + c.L->q_off),
           *k = (uint16_t*)(A + c.L->k_off), *vt = (uint16_t*)(A + c.L->vt_off), *o = (uint16_t*)(A + c.L->o_off),
           *xn = (uint16_t*)(A + c.L->xn_off);
  auto G = [&](const void* a_, int lda, const void* w_, int ldw, void* out, int ldc, int M, int N, int K,
               const void* bias, const void* res, int ldr) {
    return rtv_gemm(a_, lda, w_, ldw, out, ldc, M, N, K, bias, 0, nullptr, 0, 0, 0, res, ldr, RTV_DTYPE_F16, 0, c.stream);
  };
  RTV_TRY(rtv_rmsnorm_silu_cl(x, xn, a.gamma, C, P, 0, c.stream));
  RTV_TRY(G(xn, C, a.wq, C, q, C, P, C, C, a.bq, nullptr, 0));      // wq/bq carry the 1/sqrt(C) softmax scale
  RTV_TRY(G(xn, C, a.wk, C, k, C, P, C, C, a.bk, nullptr, 0));
  RTV_TRY(G(a.wv, C, xn, C, vt, ldp, C, P, C, nullptr, nullptr, 0)); // V^T [C][P] (bias folded below: rows of P sum to 1)
  RTV_TRY(G(q, C, k, C, S, P, P, P, C, nullptr, nullptr, 0));        // S = q k^T
  RTV_TRY(rtv_softmax_rows(S, P, Pm, ldp, P, P, c.stream));
  RTV_TRY(G(Pm, ldp, vt, ldp, o, C, P, C, ldp, a.bv, nullptr, 0));   // O = P V + bv
  RTV_TRY(G(o, C, a.wproj, C, y, C, P, C, C, a.bproj, x, C));        // proj + identity
  return 0;
}

}  // namespace

extern "C" int rtv_conv_cl_win(const void* in, const void* w, const void* bias, const void* residual, int res_ld,
                               void* out, int out_ld, int T, int H, int W, int Cin, int Cout, int kt, int kh, int kw,
                               int resample, int n_split, const void* zeros, int y_out0, int y_in0, int in_rows,
                               int img_rows, rtv_stream_t stream);

/* Decode producing only pixel rows [row0, row1) (a horizontal stripe of every frame): the spatially sharded decode of
 * the context-parallel path.  Stage 0 runs on the whole latent image; stages 1-3 on the row windows of make_plan().
 * pixels: float32 [T'][3][row1-row0][8w].  row0 = 0, row1 = 8h is the plain decode. */
// `single`: the graph of VAEDecoderWrapperSingle (demo_utils/vae.py:150-314, the reference's one-latent-frame / TensorRT-export
// form) instead of VAEDecoderWrapper's (demo_utils/vae_block3.py).  The two differ in the temporal upsampling only
// (demo_utils/vae.py:72-123 vs vae_block3.py:46-67): on the first frame the single form does not skip time_conv's doubling but
// interleaves a ZERO frame in front of every frame (cat([zeros, x], dim=1) + the same reshape / stack), leaving the time_conv cache
// as it was, so a first call yields 4 pixel frames; and its time_conv cache after a one-frame call is [zeros, x], not
// [where(old == 0, 0, x), x].
static int vae_decode_impl(const rtv_vae_weights* w, const void* z, int T, int h, int wd, int first, int row0, int row1,
                           void* arena, size_t arena_bytes, void* pixels, rtv_stream_t stream_, bool single) {
  if (!w || !z || !arena || !pixels) return set_error(-1, "vae_decode: null argument");
  if (T <= 0) return 0;
  if ((h * wd) % 8) return set_error(-1, "vae_decode: h*w must be a multiple of 8");
  if (((uintptr_t)arena) & 255) return set_error(-1, "vae_decode: arena must be 256-byte aligned");
  RowPlan P;
  RTV_TRY(make_plan(h, row0, row1, &P));
  VaeLayout L;
  build_layout(h, wd, P, &L);
  if (arena_bytes < L.total) return set_error(-1, "vae_decode: arena too small (see rtv_vae_arena_bytes)");
  hipStream_t stream = (hipStream_t)stream_;
  Ctx c{(char*)arena, &L, h, wd, stream, 0};
  const int W8 = wd * 8;
  const int64_t out_hw = (int64_t)(row1 - row0) * W8;
  int out_frame = 0;
  // V^T pad columns (K padding of the P.V GEMM) must be zero
  if (L.ldp != h * wd)
    if (hipMemsetAsync((char*)arena + L.vt_off, 0, (size_t)384 * L.ldp * 2, stream) != hipSuccess)
      return set_error(-1, "vae_decode: memset failed");

  for (int f = 0; f < T; ++f) {
    const bool is_first = first && f == 0;
    int H = h, W = wd, Tn = 1;  // H = rows of the current stage's window
    uint16_t *a0 = c.act(0), *a1 = c.act(1), *a2 = c.act(2), *a3 = c.act(3);
    // conv2 (1x1x1, 16->16) + de-normalisation, written into conv1's concat buffer (32-channel padded)
    {
      const size_t sl = (size_t)H * W * 32;
      hipLaunchKernelGGL(vae_prep_kernel, dim3((H * W + 127) / 128), dim3(128), 0, stream, (const f16_t*)z, f, H * W,
                         (const float*)w->mean, (const float*)w->std, (const float*)w->conv2_w,
                         (const float*)w->conv2_b, (f16_t*)(c.cat(0) + 2 * sl));
      RTV_TRY(check_launch("vae_prep"));
    }
    RTV_TRY(cached_conv3(c, 0, 1, H, W, 32, w->conv1, 384, nullptr, a0, 384));
    RTV_TRY(res_block(c, 1, 1, H, W, 384, 384, w->mid0, a0, a1, a2, a3));
    RTV_TRY(mid_attention(c, w->attn, a3, a0));
    RTV_TRY(res_block(c, 3, 1, H, W, 384, 384, w->mid2, a0, a1, a2, a3));
    uint16_t* x = a3;  // current activation; the other three buffers are free
    int ci = 5;
    for (int s = 0; s < 4; ++s) {
      const int cout = (s == 0 || s == 1) ? 384 : (s == 2 ? 192 : 96);
      int cin = (s == 0) ? 384 : (s == 1 ? 192 : (s == 2 ? 192 : 96));
      for (int r = 0; r < 3; ++r) {
        uint16_t* bufs[3];
        int nb = 0;
        for (int i = 0; i < 4; ++i)
          if (c.act(i) != x) bufs[nb++] = c.act(i);
        RTV_TRY(res_block(c, ci, Tn, H, W, cin, cout, w->up[s * 3 + r], x, bufs[0], bufs[1], bufs[2]));
        x = bufs[2];
        ci += 2;
        cin = cout;
      }
      if (s == 3) break;
      uint16_t* free_[3];
      int nb = 0;
      for (int i = 0; i < 4; ++i)
        if (c.act(i) != x) free_[nb++] = c.act(i);
      if (s < 2) {  // upsample3d: temporal doubling through time_conv, skipped for the very first frame
        const size_t sl = (size_t)H * W * cout;
        uint16_t* buf = c.cat(ci);
        if (is_first && single) {   // demo_utils/vae.py:112-123 with is_first_frame: frames [0, x_0, 0, x_1, ...], cache untouched
          for (int t = 0; t < Tn; ++t) {
            if (hipMemsetAsync(free_[0] + (size_t)(2 * t) * sl, 0, sl * 2, stream) != hipSuccess ||
                hipMemcpyAsync(free_[0] + (size_t)(2 * t + 1) * sl, x + (size_t)t * sl, sl * 2, hipMemcpyDeviceToDevice, stream) !=
                    hipSuccess)
              return set_error(-1, "vae_decode: memset / memcpy failed");
          }
          x = free_[0];
          Tn *= 2;
          nb = 0;
          for (int i = 0; i < 4; ++i)
            if (c.act(i) != x) free_[nb++] = c.act(i);
        } else if (!is_first) {
          if (hipMemcpyAsync(buf + 2 * sl, x, (size_t)Tn * sl * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return set_error(-1, "vae_decode: memcpy failed");
          RTV_TRY(rtv_conv_cl(buf, w->time_conv[s].w, w->time_conv[s].b, nullptr, 0, free_[0], cout, Tn, H, W, cout,
                              2 * cout, 3, 1, 1, 0, cout, c.zeros(), stream));
          if (Tn == 1 && single) {   // demo_utils/vae.py:106-111: cache <- [zeros, x]
            if (hipMemsetAsync(buf, 0, sl * 2, stream) != hipSuccess ||
                hipMemcpyAsync(buf + sl, buf + 2 * sl, sl * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess)
              return set_error(-1, "vae_decode: memset / memcpy failed");
          } else if (Tn == 1) {
            hipLaunchKernelGGL(upsample_cache_t1_kernel, dim3(1024), dim3(256), 0, stream, (f16_t*)buf, (int64_t)sl);
            RTV_TRY(check_launch("upsample_cache_t1"));
          } else {
            if (hipMemcpyAsync(buf, buf + (size_t)Tn * sl, 2 * sl * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess)
              return set_error(-1, "vae_decode: memcpy failed");
          }
          x = free_[0];
          Tn *= 2;
          nb = 0;
          for (int i = 0; i < 4; ++i)
            if (c.act(i) != x) free_[nb++] = c.act(i);
        }
        ci += 1;
      }
      // nearest 2x + Conv2d 3x3 (cout -> cout/2), per frame: produces the row window of the next stage from this one's
      const int Hn = P.b[s + 1] - P.a[s + 1];
      RTV_TRY(rtv_conv_cl_win(x, w->resample[s].w, w->resample[s].b, nullptr, 0, free_[0], cout / 2, Tn, Hn, 2 * W, cout,
                              cout / 2, 1, 3, 3, RTV_CONV_UPSAMPLE2X, 0, c.zeros(), P.a[s + 1], P.a[s], H, h << (s + 1),
                              stream));
      x = free_[0];
      H = Hn;
      W *= 2;
    }
    // head: RMS_norm, SiLU, conv 96 -> 3 (filters padded to 8)
    {
      const size_t sl = (size_t)H * W * 96;
      RTV_TRY(rtv_rmsnorm_silu_cl(x, c.cat(31) + 2 * sl, w->head_gamma, 96, (int64_t)Tn * H * W, 1, stream));
      uint16_t* ho = (uint16_t*)((char*)arena + L.head_off);
      RTV_TRY(cached_conv3(c, 31, Tn, H, W, 96, w->head, 8, nullptr, ho, 8));
      hipLaunchKernelGGL(vae_final_kernel, dim3(2048), dim3(256), 0, stream, (const f16_t*)ho,
                         (float*)pixels + (size_t)out_frame * 3 * out_hw, Tn, out_hw, (int64_t)H * W,
                         (int64_t)(row0 - P.a[3]) * W);
      RTV_TRY(check_launch("vae_final"));
      out_frame += Tn;
    }
  }
  return flush_caches(c, 32);
}

extern "C" int rtv_vae_decode_rows(const rtv_vae_weights* w, const void* z, int T, int h, int wd, int first, int row0,
                                   int row1, void* arena, size_t arena_bytes, void* pixels, rtv_stream_t stream) {
  return vae_decode_impl(w, z, T, h, wd, first, row0, row1, arena, arena_bytes, pixels, stream, false);
}

extern "C" int rtv_vae_decode(const rtv_vae_weights* w, const void* z, int T, int h, int wd, int first, void* arena,
                              size_t arena_bytes, void* pixels, rtv_stream_t stream) {
  return vae_decode_impl(w, z, T, h, wd, first, 0, h << 3, arena, arena_bytes, pixels, stream, false);
}

extern "C" int rtv_vae_decode_single(const rtv_vae_weights* w, const void* z, int h, int wd, int is_first_frame, void* arena,
                                     size_t arena_bytes, void* pixels, rtv_stream_t stream) {
  return vae_decode_impl(w, z, 1, h, wd, is_first_frame ? 1 : 0, 0, h << 3, arena, arena_bytes, pixels, stream, true);
}

// ====================================================================================== streaming encoder
// One call = one time chunk of VAEEncoderWrapper.forward (demo_utils/vae_block3.py:138-175): Encoder3d.forward
// (wan/modules/vae.py:307-345) over 1 frame (first chunk of a stream, fresh caches) or 4 frames, then the wrapper's
// 1x1x1 conv1, chunk(2) -> mu and the latent normalisation.  24 feature caches: 22 two-slice conv caches and the two
// single-frame caches of the downsample3d time convs (vae.py:139-158), all kept as the leading slices of the convs'
// concat buffers in a caller-owned arena.
namespace rtv {

// frames f16 [3][Ttot][H][W] (frames t0..t0+T) -> channels-last [T][H][W][32] (3 real channels + zero padding)
__global__ void vae_enc_prep_kernel(const f16_t* __restrict__ frames, int Ttot, int t0, int T, int64_t hw,
                                    f16_t* __restrict__ out) {
  const int64_t total = (int64_t)T * hw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = i / hw, p = i - t * hw;
    const f16_t r = frames[((int64_t)0 * Ttot + t0 + t) * hw + p];
    const f16_t g = frames[((int64_t)1 * Ttot + t0 + t) * hw + p];
    const f16_t b = frames[((int64_t)2 * Ttot + t0 + t) * hw + p];
    u32x4* dst = (u32x4*)(out + i * 32);
    dst[0] = u32x4{(uint32_t)r | ((uint32_t)g << 16), (uint32_t)b, 0u, 0u};
    dst[1] = u32x4{0u, 0u, 0u, 0u};
    dst[2] = u32x4{0u, 0u, 0u, 0u};
    dst[3] = u32x4{0u, 0u, 0u, 0u};
  }
}

// head output [T][hw][32] f16 -> conv1 (1x1x1, first 16 of 32 output channels = mu) -> (mu - mean) * (1/std), with the
// reference's fp16 rounding points (vae_block3.py:168-172) -> mu f16 [16][Tout_tot][hw] at frames tout..tout+T
__global__ void vae_enc_final_kernel(const f16_t* __restrict__ in, int T, int64_t hw, const float* __restrict__ w1 /*[32][32]*/,
                                     const float* __restrict__ b1, const float* __restrict__ mean,
                                     const float* __restrict__ stdv, f16_t* __restrict__ mu, int Tout_tot, int tout) {
  const int64_t total = (int64_t)T * hw;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int64_t t = i / hw, p = i - t * hw;
  float x[32];
  const u32x4* src = (const u32x4*)(in + i * 32);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    u32x4 raw = src[q];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t u = raw[j];
      unpack_f16x2(u, x[q * 8 + 2 * j], x[q * 8 + 2 * j + 1]);
    }
  }
#pragma unroll 4
  for (int co = 0; co < 16; ++co) {
    float a = b1[co];
#pragma unroll
    for (int c = 0; c < 32; ++c) a += w1[co * 32 + c] * x[c];
    const float inv = round_f16(1.0f / round_f16(stdv[co]));
    const float v = round_f16(round_f16(round_f16(a) - round_f16(mean[co])) * inv);
    mu[((int64_t)co * Tout_tot + tout + t) * hw + p] = f32_to_f16(v);
  }
}

}  // namespace rtv

namespace {

// concat buffers of the encoder in execution order: (input channels, stage = log2 of the spatial reduction, max new slices)
static void build_enc_layout(int H, int W, VaeLayout* L) {
  const int C[24] = {32, 96, 96, 96, 96,        // conv1, downsamples.0 (a,b), .1 (a,b)
                     96, 192, 192, 192, 192,      // .3 (96->192: a,b), .4 (a,b), .5 time_conv
                     192, 384, 384, 384, 384,     // .6 (192->384: a,b), .7 (a,b), .8 time_conv
                     384, 384, 384, 384,          // .9, .10
                     384, 384, 384, 384, 384};    // middle.0, middle.2, head
  const int S[24] = {0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3};
  const int Tm[24] = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 2, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t a = (off + 255) & ~(size_t)255;
    off = a + bytes;
    return a;
  };
  for (int i = 0; i < 32; ++i) {
    L->cat_off[i] = 0;
    L->cat_C[i] = L->cat_stage[i] = L->cat_T[i] = 0;
  }
  for (int i = 0; i < 24; ++i) {
    const size_t hw = (size_t)(H >> S[i]) * (W >> S[i]);
    L->cat_C[i] = C[i];
    L->cat_stage[i] = S[i];
    L->cat_T[i] = Tm[i];
    L->cat_cap[i] = 2 + Tm[i];          // the encoder rolls after every chunk (one chunk per call)
    L->cat_slice[i] = hw * C[i] * 2;
    L->cat_off[i] = take((size_t)(2 + Tm[i]) * hw * C[i] * 2);
  }
  // widest activation: 4 x H x W x 96 (stage 0); 4 x H/2 x W/2 x 192 is half of it
  const size_t amax = (size_t)4 * H * W * 96 * 2;
  for (int i = 0; i < 4; ++i) L->act_off[i] = take(amax);
  const size_t P = (size_t)(H >> 3) * (W >> 3);
  L->ldp = (int)((P + 63) / 64 * 64);
  L->s_off = take(P * P * 2);
  L->p_off = take(P * (size_t)L->ldp * 2);
  L->q_off = take(P * 384 * 2);
  L->k_off = take(P * 384 * 2);
  L->vt_off = take((size_t)384 * L->ldp * 2);
  L->o_off = take(P * 384 * 2);
  L->xn_off = take(P * 384 * 2);
  L->head_off = take(P * 32 * 2);
  L->zeros_off = take(256);
  L->total = off + 256;
}

static int enc_dims_ok(int H, int W) {
  if (H <= 0 || W <= 0 || (H & 7) || (W & 7)) return set_error(-1, "vae_encode: H and W must be positive multiples of 8");
  if (((H >> 3) * (W >> 3)) % 8) return set_error(-1, "vae_encode: (H/8)*(W/8) must be a multiple of 8");
  return 0;
}

}  // namespace

extern "C" size_t rtv_vae_enc_arena_bytes(int H, int W) {
  if (enc_dims_ok(H, W)) return 0;
  VaeLayout L;
  build_enc_layout(H, W, &L);
  return L.total;
}

/* byte offset + geometry of encoder feature-cache slot i (0..23): the cached slices are the `nslices` slices that end
 * at slice 2 of the conv's concat buffer (2 for the 3x3x3 convs, 1 for the downsample3d frame caches, slots 9 and 14:
 * `offset` already points at that slice). */
extern "C" int rtv_vae_enc_cache_slot(int H, int W, int slot, size_t* offset, int* C, int* h, int* w, int* nslices) {
  if (slot < 0 || slot >= 24) return set_error(-1, "vae_enc_cache_slot: slot out of range");
  RTV_TRY(enc_dims_ok(H, W));
  VaeLayout L;
  build_enc_layout(H, W, &L);
  const int s = L.cat_stage[slot];
  const bool frame_cache = slot == 9 || slot == 14;
  *C = L.cat_C[slot];
  *h = H >> s;
  *w = W >> s;
  *nslices = frame_cache ? 1 : 2;
  *offset = L.cat_off[slot] + (frame_cache ? (size_t)(H >> s) * (W >> s) * L.cat_C[slot] * 2 : 0);
  return 0;
}

extern "C" int rtv_vae_encode(const rtv_vae_enc_weights* w, const void* frames, int Ttot, int t0, int tn, int H, int W,
                              int first, void* arena, size_t arena_bytes, void* mu, int Tout_tot, int tout,
                              rtv_stream_t stream_) {
  if (!w || !frames || !arena || !mu) return set_error(-1, "vae_encode: null argument");
  RTV_TRY(enc_dims_ok(H, W));
  if (first ? tn != 1 : tn != 4)
    return set_error(-1, "vae_encode: a chunk is 1 frame on fresh caches (first=1) or 4 frames (vae_block3.py:146-166)");
  if (t0 < 0 || t0 + tn > Ttot) return set_error(-1, "vae_encode: frame range outside the clip");
  if (tout < 0 || tout + 1 > Tout_tot) return set_error(-1, "vae_encode: latent frame index outside the output");
  if (((uintptr_t)arena) & 255) return set_error(-1, "vae_encode: arena must be 256-byte aligned");
  VaeLayout L;
  build_enc_layout(H, W, &L);
  if (arena_bytes < L.total) return set_error(-1, "vae_encode: arena too small (see rtv_vae_enc_arena_bytes)");
  hipStream_t stream = (hipStream_t)stream_;
  const int h = H >> 3, wd = W >> 3;
  Ctx c{(char*)arena, &L, h, wd, stream, 0};
  c.fresh = first && g_fresh_tap_skip.load(std::memory_order_relaxed);
  if (L.ldp != h * wd)
    if (hipMemsetAsync((char*)arena + L.vt_off, 0, (size_t)384 * L.ldp * 2, stream) != hipSuccess)
      return set_error(-1, "vae_encode: memset failed");

  if (first) {
    // a fresh stream reads zeros wherever the reference has no cache yet (vae.py:17-36: zero padding in front of the first
    // chunk): the two leading slices of every concat buffer + the padding page.  Cleared here so that an arena may be REUSED
    // for a new stream without the caller zero-filling all of it (vae_encoder.py: recycled one-shot arenas).
    for (int i = 0; i < 24; ++i) {
      const size_t sl = (size_t)(H >> L.cat_stage[i]) * (W >> L.cat_stage[i]) * L.cat_C[i] * 2;
      if (hipMemsetAsync((char*)arena + L.cat_off[i], 0, 2 * sl, stream) != hipSuccess)
        return set_error(-1, "vae_encode: memset failed");
    }
    if (hipMemsetAsync((char*)arena + L.zeros_off, 0, 256, stream) != hipSuccess)
      return set_error(-1, "vae_encode: memset failed");
  }
  int T = tn, Hs = H, Ws = W;
  // pixels -> conv1's concat buffer (channels-last, 32-channel padded)
  hipLaunchKernelGGL(vae_enc_prep_kernel, dim3(2048), dim3(256), 0, stream, (const f16_t*)frames, Ttot, t0, T,
                     (int64_t)H * W, (f16_t*)(c.cat(0) + 2 * (size_t)H * W * 32));
  RTV_TRY(check_launch("vae_enc_prep"));
  uint16_t* x = c.act(0);
  RTV_TRY(cached_conv3(c, 0, T, Hs, Ws, 32, w->conv1, 96, nullptr, x, 96));
  auto others = [&](uint16_t** out3) {  // the three activation buffers that do not hold x
    int nb = 0;
    for (int i = 0; i < 4 && nb < 3; ++i)
      if (c.act(i) != x) out3[nb++] = c.act(i);
  };
  const int dims[5] = {96, 96, 192, 384, 384};
  int ci = 1;
  for (int s = 0; s < 4; ++s) {
    int cin = dims[s];
    const int cout = dims[s + 1];
    for (int r = 0; r < 2; ++r) {
      uint16_t* b[3];
      others(b);
      RTV_TRY(res_block(c, ci, T, Hs, Ws, cin, cout, w->down[s * 2 + r], x, b[0], b[1], b[2]));
      x = b[2];
      ci += 2;
      cin = cout;
    }
    if (s == 3) break;
    uint16_t* b[3];
    others(b);
    if (s == 0) {  // downsample2d: ZeroPad2d((0,1,0,1)) + Conv2d(3, stride 2), per frame
      RTV_TRY(rtv_conv_cl(x, w->resample[s].w, w->resample[s].b, nullptr, 0, b[0], cout, T, Hs / 2, Ws / 2, cout, cout, 1,
                          3, 3, RTV_CONV_DOWN2X, 0, c.zeros(), stream));
      x = b[0];
    } else {       // downsample3d: the spatial conv writes the new slices of the time_conv's concat buffer
      const size_t sl = (size_t)(Hs / 2) * (Ws / 2) * cout;  // elements per slice at the reduced resolution
      uint16_t* buf = c.cat(ci);
      RTV_TRY(rtv_conv_cl(x, w->resample[s].w, w->resample[s].b, nullptr, 0, buf + 2 * sl, cout, T, Hs / 2, Ws / 2, cout,
                          cout, 1, 3, 3, RTV_CONV_DOWN2X, 0, c.zeros(), stream));
      if (first) {  // vae.py:142-144: the first frame is only stored
        if (hipMemcpyAsync(b[0], buf + 2 * sl, sl * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess)
          return set_error(-1, "vae_encode: memcpy failed");
      } else {      // time_conv (3,1,1) / stride (2,1,1) over [last cached frame | T new frames]
        RTV_TRY(rtv_conv_cl(buf + sl, w->time_conv[s - 1].w, w->time_conv[s - 1].b, nullptr, 0, b[0], cout, T / 2, Hs / 2,
                            Ws / 2, cout, cout, 3, 1, 1, RTV_CONV_TIME_DOWN2X, 0, c.zeros(), stream));
      }
      // cache <- last new frame
      if (hipMemcpyAsync(buf + sl, buf + (size_t)(1 + T) * sl, sl * 2, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return set_error(-1, "vae_encode: memcpy failed");
      x = b[0];
      if (!first) T /= 2;
      ci += 1;
    }
    Hs /= 2;
    Ws /= 2;
  }
  {
    uint16_t* b[3];
    others(b);
    RTV_TRY(res_block(c, ci, T, Hs, Ws, 384, 384, w->mid0, x, b[0], b[1], b[2]));
    RTV_TRY(mid_attention(c, w->attn, b[2], b[0]));
    RTV_TRY(res_block(c, ci + 2, T, Hs, Ws, 384, 384, w->mid2, b[0], b[1], b[2], x));
    ci += 4;
  }
  {
    const size_t sl = (size_t)Hs * Ws * 384;
    RTV_TRY(rtv_rmsnorm_silu_cl(x, c.cat(ci) + 2 * sl, w->head_gamma, 384, (int64_t)T * Hs * Ws, 1, stream));
    uint16_t* ho = (uint16_t*)((char*)arena + L.head_off);
    RTV_TRY(cached_conv3(c, ci, T, Hs, Ws, 384, w->head, 32, nullptr, ho, 32));
    const int64_t hw = (int64_t)Hs * Ws;
    hipLaunchKernelGGL(vae_enc_final_kernel, dim3((unsigned)((T * hw + 127) / 128)), dim3(128), 0, stream, (const f16_t*)ho,
                       T, hw, (const float*)w->conv1x1_w, (const float*)w->conv1x1_b, (const float*)w->mean,
                       (const float*)w->std, (f16_t*)mu, Tout_tot, tout);
    RTV_TRY(check_launch("vae_enc_final"));
  }
  return flush_caches(c, 24);   // (a no-op today: the encoder's buffers hold one chunk, every roll copies)
}

extern "C" int rtv_vae_set_fresh_tap_skip(int on) {   // include/rtv_hip_lab.h
  g_fresh_tap_skip.store(on != 0, std::memory_order_relaxed);
  return 0;
}
