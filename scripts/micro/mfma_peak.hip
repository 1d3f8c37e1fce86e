// Sustained MFMA rate of the chip (no memory traffic): W waves per SIMD, each looping over independent
// 32x32x16 bf16 MFMAs.  Gives the practical ceiling (power/clock-limited) that the GEMM / attention rates
// are compared with in DESIGN.md.   build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, int RANDOM>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* clk) {
  // operands: RANDOM != 0 -> 4 register sets of pseudo-random normal-ish bf16 values cycled per MFMA (realistic
  // operand toggling -> realistic power); 0 -> one constant set (best case for the clock governor)
  bf16x8 a[4], b[4];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; float x = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      h = h * 1664525u + 1013904223u; float y = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      a[s][i] = (__bf16)(RANDOM ? x * 3.f : 1.0f); b[s][i] = (__bf16)(RANDOM ? y * 3.f : 0.5f);
    }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + n) & 3], b[(u * 3 + n) & 3], acc[n], 0, 0, 0);
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

int main(int argc, char** argv) {
  int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
  int iters = argc > 2 ? atoi(argv[2]) : 200000;
  int random = argc > 3 ? atoi(argv[3]) : 1;
  int blocks = 256 * waves_per_simd;  // 4 waves per block = one per SIMD
  float* out; long long* clk;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    if (random) mfma_loop<4, 1><<<blocks, 256>>>(out, iters, clk);
    else mfma_loop<4, 0><<<blocks, 256>>>(out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 16.0 * 2.0 * 32 * 32 * 16;
    printf("random=%d waves/SIMD=%d  %.2f ms  %.1f TF/s   shader clocks/MFMA(wave0)=%.2f  wall(100MHz) ticks=%lld -> shader clock %.0f MHz\n",
           random, waves_per_simd, ms, flops / ms / 1e9, (double)h[0] / (iters * 16.0), h[1], (double)h[0] / ((double)h[1] / 100.0));
  }
  return 0;
}
