// As mfma_peak.hip but with v_mfma_f32_16x16x32_bf16 (4 passes, f32x4 accumulators): same FLOPs per cycle on paper;
// does it sustain a different clock / rate than 32x32x16 on random operands?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* clk) {
  bf16x8 a[4], b[4];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; float x = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      h = h * 1664525u + 1013904223u; float y = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      a[s][i] = (__bf16)(RANDOM ? x * 3.f : 1.0f); b[s][i] = (__bf16)(RANDOM ? y * 3.f : 0.5f);
    }
  f32x4 acc[16];
  for (int n = 0; n < 16; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int n = 0; n < 16; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(u + n) & 3], b[(u * 3 + n) & 3], acc[n], 0, 0, 0);
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < 16; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

int main(int argc, char** argv) {
  int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 200000, random = argc > 3 ? atoi(argv[3]) : 1;
  int blocks = 256 * wps;
  float* out; long long* clk;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    if (random) mfma_loop<1><<<blocks, 256>>>(out, iters, clk); else mfma_loop<0><<<blocks, 256>>>(out, iters, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long hh[2]; (void)hipMemcpy(hh, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 32.0 * 2.0 * 16 * 16 * 32;
    printf("16x16x32 random=%d waves/SIMD=%d  %.2f ms  %.1f TF/s  cycles/MFMA(wave0)=%.2f  shader clock %.0f MHz\n", random, wps, ms,
           flops / ms / 1e9, (double)hh[0] / (iters * 32.0), (double)hh[0] / ((double)hh[1] / 100.0));
  }
  return 0;
}
