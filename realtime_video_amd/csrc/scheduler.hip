// Flow-matching scheduler arithmetic of one denoising step in ONE launch: the nearest-timestep lookup, the flow -> x0
// conversion in float64 (utils/wan_wrapper.py:181-205) and the re-noising of x0 to the next step's level
// (utils/scheduler.py:159-176).  The latents of a block are small (3 x 16 x 60 x 104 elements), so this is launch-count work:
// the eager chain is ~25 tiny launches per denoising step, here it is one.  Every rounding point of the eager chain is kept
// (no FMA contraction: products and sums are rounded separately, as separate torch kernels round them).
#include "rtv_common.h"
#include "rtv_internal.h"

// A product and the sum that follows it are two roundings everywhere in this file (plain operators, not the __*_rn
// intrinsics: those are header inlines compiled under the default contraction mode and still fuse).
#pragma clang fp contract(off)

namespace rtv {

struct SchedArgs {
  const bf16_t* flow;   // [F][C][hw] through strides (elements); null = x0 is read from `x0` (add_noise alone)
  const bf16_t* xt;
  int64_t flow_sf, flow_sc, xt_sf, xt_sc;
  const void* t;        // [F] timestep of this step (dtype t_kind)
  const void* t_next;   // [F] timestep of the next step, null = no re-noising
  int t_kind;           // 0 float32, 1 float64, 2 int64
  const float* timesteps;
  const float* sigmas;
  int n_table;
  bf16_t* x0;           // [F][C][hw] contiguous
  const bf16_t* noise;  // [F][C][hw] contiguous
  bf16_t* noisy;        // [F][C][hw] contiguous
  int C, hw;
};

// float -> bf16 (RNE) on the bit pattern: written out so that the compiler cannot merge it with the preceding
// double -> float truncation into ONE rounding (it does for `(__bf16)(float)d`; torch rounds twice and ties differ).
__device__ __forceinline__ bf16_t f32_bits_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

__device__ __forceinline__ double load_t(const void* p, int kind, int f) {
  if (kind == 0) return (double)((const float*)p)[f];
  if (kind == 1) return ((const double*)p)[f];
  return (double)((const int64_t*)p)[f];
}

// index of the smallest |table[i] - t| (first one on ties, like torch.argmin), in float64 (F64) or float32 arithmetic
template <bool F64>
__device__ int nearest_timestep(const float* table, int n, double t, int* red_i, double* red_v) {
  double best = 1e300;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    double d;
    if (F64) {
      d = fabs((double)table[i] - t);
    } else {
      d = (double)fabsf(table[i] - (float)t);
    }
    if (d < best) {  // i ascends per thread: strict '<' keeps the first
      best = d;
      bi = i;
    }
  }
  red_v[threadIdx.x] = best;
  red_i[threadIdx.x] = bi;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      double ov = red_v[threadIdx.x + s];
      int oi = red_i[threadIdx.x + s];
      if (ov < red_v[threadIdx.x] || (ov == red_v[threadIdx.x] && oi < red_i[threadIdx.x])) {
        red_v[threadIdx.x] = ov;
        red_i[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  int r = red_i[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(256) void scheduler_step_kernel(SchedArgs a) {
  __shared__ int red_i[256];
  __shared__ double red_v[256];
  const int f = blockIdx.y;
  double sigma64 = 0.0;
  float sigma_next = 0.f;
  if (a.flow) {
    // wan_wrapper.py:196-200: timesteps/sigmas as float64, argmin of |timesteps - t|
    int idx = nearest_timestep<true>(a.timesteps, a.n_table, load_t(a.t, a.t_kind, f), red_i, red_v);
    sigma64 = (double)a.sigmas[idx];
  }
  if (a.t_next) {
    // scheduler.py:168-172: float32 table minus the timestep (int64 / float32 promote to float32)
    // (a float64 timestep promotes the difference to float64)
    const double tn = load_t(a.t_next, a.t_kind, f);
    int idx = a.t_kind == 1 ? nearest_timestep<true>(a.timesteps, a.n_table, tn, red_i, red_v)
                            : nearest_timestep<false>(a.timesteps, a.n_table, tn, red_i, red_v);
    sigma_next = a.sigmas[idx];
  }
  const float one_minus = 1.0f - sigma_next;
  const int per_frame = a.C * a.hw;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < per_frame; e += gridDim.x * blockDim.x) {
    const int c = e / a.hw, p = e - c * a.hw;
    const size_t o = (size_t)f * per_frame + e;
    bf16_t x0b;
    if (a.flow) {
      double fp = (double)bf16_to_f32(a.flow[f * a.flow_sf + c * a.flow_sc + p]);
      double x = (double)bf16_to_f32(a.xt[f * a.xt_sf + c * a.xt_sc + p]);
      const double prod = sigma64 * fp;
      const double x0 = x - prod;                          // wan_wrapper.py:201
      x0b = f32_bits_to_bf16(__double2float_rn(x0));       // .to(bf16): double -> float -> bf16, two roundings, as torch casts
      a.x0[o] = x0b;
    } else {
      x0b = a.x0[o];
    }
    if (a.t_next) {
      // scheduler.py:174-175: (1 - sigma) * x0 + sigma * noise in float32, then .type_as(noise)
      const float keep = one_minus * bf16_to_f32(x0b), add = sigma_next * bf16_to_f32(a.noise[o]);
      const float s = keep + add;
      a.noisy[o] = f32_to_bf16(s);
    }
  }
}

}  // namespace rtv

using namespace rtv;

extern "C" int rtv_scheduler_step(const void* flow, int64_t flow_frame_stride, int64_t flow_channel_stride,
                                  const void* xt, int64_t xt_frame_stride, int64_t xt_channel_stride,
                                  const void* t, const void* t_next, int t_kind,
                                  const void* timesteps, const void* sigmas, int n_table,
                                  void* x0, const void* noise, void* noisy,
                                  int F, int C, int hw, rtv_stream_t stream) {
  if (t_kind < 0 || t_kind > 2) return set_error(-1, "scheduler_step: t_kind must be 0 (f32), 1 (f64) or 2 (i64)");
  if (n_table <= 0 || !timesteps || !sigmas) return set_error(-1, "scheduler_step: empty timestep table");
  if (!x0) return set_error(-1, "scheduler_step: x0 buffer missing");
  if (flow && (!xt || !t)) return set_error(-1, "scheduler_step: flow needs xt and t");
  if (t_next && (!noise || !noisy)) return set_error(-1, "scheduler_step: t_next needs noise and an output buffer");
  if (!flow && !t_next) return set_error(-1, "scheduler_step: nothing to do (no flow, no t_next)");
  if (F <= 0 || C <= 0 || hw <= 0) return 0;
  SchedArgs a;
  a.flow = (const bf16_t*)flow;
  a.xt = (const bf16_t*)xt;
  a.flow_sf = flow_frame_stride;
  a.flow_sc = flow_channel_stride;
  a.xt_sf = xt_frame_stride;
  a.xt_sc = xt_channel_stride;
  a.t = t;
  a.t_next = t_next;
  a.t_kind = t_kind;
  a.timesteps = (const float*)timesteps;
  a.sigmas = (const float*)sigmas;
  a.n_table = n_table;
  a.x0 = (bf16_t*)x0;
  a.noise = (const bf16_t*)noise;
  a.noisy = (bf16_t*)noisy;
  a.C = C;
  a.hw = hw;
  int bx = (C * hw + 1023) / 1024;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(scheduler_step_kernel, dim3(bx, F), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("scheduler_step");
}
