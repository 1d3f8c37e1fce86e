// Does VALU work of one wave hide under the MFMAs of the OTHER wave of its SIMD?  512-thread workgroups (two waves per SIMD):
// waves 0-3 issue 16 independent v_mfma_f32_32x32x16_bf16 per iteration, waves 4-7 a VALU loop (mode: 1 = 32 v_exp_f32,
// 2 = 64 v_fma_f32, 3 = 32 v_exp + 32 v_pk_fma + 16 v_max3 - roughly a softmax step).  Each role is also run alone.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu_coissue.hip -o mfma_valu_coissue ; run: mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int MODE>
__global__ __launch_bounds__(512, 1) void coissue(float* out, int iters, int run_mfma, int run_valu, long long* clk) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  long long c0 = 0, c1 = 0;
  if (wave < 4) {
    if (run_mfma) {
      bf16x8 a[4], b[4];
      unsigned h = threadIdx.x * 2654435761u + 12345u;
      for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) {
          h = h * 1664525u + 1013904223u; a[q][i] = (__bf16)(((h >> 8) & 0xffff) / 65536.f - 0.5f);
          h = h * 1664525u + 1013904223u; b[q][i] = (__bf16)(((h >> 8) & 0xffff) / 65536.f - 0.5f);
        }
      f32x16 acc[16];
      for (int n = 0; n < 16; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
      c0 = clock64();
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 16; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n >> 2], b[n & 3], acc[n], 0, 0, 0);
      }
      c1 = clock64();
      for (int n = 0; n < 16; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
      if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
    }
  } else if (run_valu) {
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = (threadIdx.x * 0.001f + i) * 1e-3f;
    c0 = clock64();
    for (int it = 0; it < iters; ++it) {
      if (MODE == 1) {
#pragma unroll
        for (int i = 0; i < 32; ++i) x[i] = __builtin_amdgcn_exp2f(x[i]) * 0.5f;   // (the mul keeps values bounded; 32 exp + 32 mul)
      } else if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 32; ++i) { x[i] = __builtin_fmaf(x[i], 0.999f, 0.001f); x[i] = __builtin_fmaf(x[i], 1.001f, -0.001f); }
      } else {
        float m = x[0];
#pragma unroll
        for (int i = 1; i < 32; i += 2) m = fmaxf(fmaxf(m, x[i]), x[(i + 1) & 31]);
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
          f32x2 v = {x[i], x[i + 1]};
          v = __builtin_elementwise_fma(v, f32x2{0.5f, 0.5f}, f32x2{-m * 1e-3f, -m * 1e-3f});
          x[i] = __builtin_amdgcn_exp2f(v[0]) * 0.25f;
          x[i + 1] = __builtin_amdgcn_exp2f(v[1]) * 0.25f;
        }
      }
      asm volatile("" ::: "memory");
    }
    c1 = clock64();
    for (int i = 0; i < 32; ++i) s += x[i];
    if (blockIdx.x == 0 && threadIdx.x == 256) clk[1] = c1 - c0;
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name) {
  int iters = 20000, blocks = 256;
  float* out; long long* clk;
  (void)hipMalloc(&out, (size_t)blocks * 512 * 4); (void)hipMalloc(&clk, 16);
  for (int cfg = 0; cfg < 3; ++cfg) {
    int rm = cfg != 1, rv = cfg != 0;
    (void)hipMemset(clk, 0, 16);
    for (int rep = 0; rep < 2; ++rep) coissue<MODE><<<blocks, 512>>>(out, iters, rm, rv, clk);
    (void)hipDeviceSynchronize();
    long long hh[2]; (void)hipMemcpy(hh, clk, 16, hipMemcpyDeviceToHost);
    printf("%-34s %-10s cycles per MFMA %6.2f   cycles per VALU iteration %8.1f\n", name, cfg == 0 ? "MFMA only" : cfg == 1 ? "VALU only" : "both",
           rm ? (double)hh[0] / (iters * 16.0) : 0.0, rv ? (double)hh[1] / iters : 0.0);
  }
}

int main() {
  run<1>("32 v_exp + 32 v_mul");
  run<2>("64 v_fma");
  run<3>("softmax-like (16 max3, 16 pk_fma, 32 exp)");
  return 0;
}
