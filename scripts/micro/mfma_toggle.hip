// Does the ORDER in which an MFMA loop cycles its operands change the power-capped rate?  Same 16 random bf16 fragment
// registers (8 A, 8 B), 16 independent accumulators, three issue orders:
//   0: both operands change with every MFMA (the pattern of mfma_peak.hip)
//   1: the A operand stays for 4 consecutive MFMAs, B cycles (a 128x128 register tile walked row by row)
//   2: A stays for 2, B alternates between 2 (gemm8's order: ks, nq, mb with mb innermost -> B shared by pairs)
//   3: both stay for 4 consecutive MFMAs (no toggling inside a group of 4: lower bound on operand switching)
// build: hipcc --offload-arch=gfx950 -O3 mfma_toggle.hip -o mfma_toggle ; run: mfma_toggle <pattern> [waves/SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int PAT>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* clk) {
  bf16x8 a[8], b[8];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int s = 0; s < 8; ++s)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; float x = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      h = h * 1664525u + 1013904223u; float y = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      a[s][i] = (__bf16)(x * 3.f); b[s][i] = (__bf16)(y * 3.f);
    }
  f32x16 acc[16];
  for (int n = 0; n < 16; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      int ia, ib;
      if (PAT == 0) { ia = n & 7; ib = (n * 3 + 1) & 7; }
      else if (PAT == 1) { ia = n >> 2; ib = n & 3; }
      else if (PAT == 2) { ia = ((n >> 2) << 1) | (n & 1); ib = (n >> 1) & 1 | ((n >> 3) << 1); }
      else { ia = n >> 2; ib = n >> 2; }
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ia], b[ib], acc[n], 0, 0, 0);
    }
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < 16; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

int main(int argc, char** argv) {
  int pat = argc > 1 ? atoi(argv[1]) : 0, wps = argc > 2 ? atoi(argv[2]) : 2, iters = 100000;
  int blocks = 256 * wps;
  float* out; long long* clk;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&clk, 16);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    if (pat == 0) mfma_loop<0><<<blocks, 256>>>(out, iters, clk);
    else if (pat == 1) mfma_loop<1><<<blocks, 256>>>(out, iters, clk);
    else if (pat == 2) mfma_loop<2><<<blocks, 256>>>(out, iters, clk);
    else mfma_loop<3><<<blocks, 256>>>(out, iters, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long hh[2]; (void)hipMemcpy(hh, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 16.0 * 2.0 * 32 * 32 * 16;
    if (rep == 2)
      printf("pattern %d waves/SIMD=%d  %.2f ms  %.1f TF/s  cycles/MFMA(wave0)=%.2f  shader clock %.0f MHz\n", pat, wps, ms, flops / ms / 1e9,
             (double)hh[0] / (iters * 16.0), (double)hh[0] / ((double)hh[1] / 100.0));
  }
  return 0;
}
