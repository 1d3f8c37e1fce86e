// How much of a removed bubble comes back as a lower clock?  MFMA-only loop (v_mfma_f32_32x32x16_bf16, random operands, 2 waves per
// SIMD, as mfma_peak.hip) with an idle gap of SLEEP x 64 cycles (s_sleep) behind every 16 MFMAs: the duty cycle of the matrix pipe goes
// from 100 % down, the power governor raises the clock in return.  Prints TF/s, shader clock, duty.  If throughput falls much less
// than the duty cycle, bubbles in the real kernels are "partly free" under the 1400 W cap and schedule tightening pays little.
//   build: hipcc --offload-arch=gfx950 -O3 mfma_duty.hip -o mfma_duty ;  run: ./mfma_duty [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SLEEP>
__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters, long long* clk) {
  bf16x8 a[4], b[4];
  unsigned h = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; float x = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      h = h * 1664525u + 1013904223u; float y = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      a[s][i] = (__bf16)(x * 3.f); b[s][i] = (__bf16)(y * 3.f);
    }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + n) & 3], b[(u * 3 + n) & 3], acc[n], 0, 0, 0);
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

template <int SLEEP>
void run(float* out, long long* clk, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 0, ms = 0; long long h[2] = {0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    mfma_loop<SLEEP><<<256, 512>>>(out, iters, clk);     // one 8-wave workgroup per CU: 2 waves per SIMD
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    best = ms;
  }
  double flops = 256.0 * 8 * iters * 16.0 * 2.0 * 32 * 32 * 16;
  double cyc_per_iter = (double)h[0] / iters;           // per wave: 16 MFMAs (2 waves share the pipe: 1024 pipe cycles per iteration pair)
  printf("sleep %2d x64: %8.2f ms  %7.1f TF/s  clock64/iter %7.1f  (clock64 rate %.0f MHz)  matrix-pipe duty ~%.2f\n", SLEEP, best,
         flops / best / 1e9, cyc_per_iter, (double)h[0] / ((double)h[1] / 100.0), 1024.0 / cyc_per_iter);
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 1000000;
  float* out; long long* clk;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 16);
  run<0>(out, clk, iters); run<1>(out, clk, iters); run<2>(out, clk, iters); run<4>(out, clk, iters);
  run<6>(out, clk, iters); run<8>(out, clk, iters); run<12>(out, clk, iters); run<16>(out, clk, iters); run<0>(out, clk, iters);
  return 0;
}
