// Sustained MFMA-only rate under the package power cap per INPUT TYPE, random operands (no memory traffic): the same loop as
// mfma_peak.hip with v_mfma_f32_32x32x16_bf16 / _f16 and v_mfma_f32_32x32x64_f8f6f4 (fp8 e4m3).  The DiT runs bf16 MFMAs, the
// VAE (the reference's decoder / encoder run in fp16) f16 ones: does the f16 multiplier array (11-bit significands against 8)
// sustain a lower rate at 1400 W?   build: hipcc --offload-arch=gfx950 -O3 mfma_dtype.hip -o bin/mfma_dtype;  run: bin/mfma_dtype [waves/SIMD] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int KIND>   // 0 bf16, 1 f16, 2 fp8 (e4m3, K = 64)
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* clk) {
  bf16x8 ab[4], bb[4];
  f16x8 ah[4], bh[4];
  i32x8 a8[4], b8[4];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; float x = (((h >> 8) & 0xffff) / 65536.f - 0.5f) * 3.f;
      h = h * 1664525u + 1013904223u; float y = (((h >> 8) & 0xffff) / 65536.f - 0.5f) * 3.f;
      ab[s][i] = (__bf16)x; bb[s][i] = (__bf16)y;
      ah[s][i] = (_Float16)x; bh[s][i] = (_Float16)y;
      h = h * 1664525u + 1013904223u; a8[s][i] = (int)(h & 0x77777777u);    // four e4m3 values per int, exponents kept finite
      h = h * 1664525u + 1013904223u; b8[s][i] = (int)(h & 0x77777777u);
    }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if constexpr (KIND == 0) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[(u + n) & 3], bb[(u * 3 + n) & 3], acc[n], 0, 0, 0);
        else if constexpr (KIND == 1) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(u + n) & 3], bh[(u * 3 + n) & 3], acc[n], 0, 0, 0);
        else acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[(u + n) & 3], b8[(u * 3 + n) & 3], acc[n], 0, 0, 0, 0, 0, 0);
      }
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

int main(int argc, char** argv) {
  int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 100000;
  int blocks = 256 * wps;
  float* out; long long* clk;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"bf16 32x32x16", "f16  32x32x16", "fp8  32x32x64"};
  for (int round = 0; round < 3; ++round)
    for (int kind = 0; kind < 3; ++kind) {
      hipEventRecord(e0);
      if (kind == 0) mfma_loop<0><<<blocks, 256>>>(out, iters, clk);
      else if (kind == 1) mfma_loop<1><<<blocks, 256>>>(out, iters, clk);
      else mfma_loop<2><<<blocks, 256>>>(out, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long hc[2]; hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
      double k = kind == 2 ? 64 : 16;
      double flops = (double)blocks * 4 * iters * 16.0 * 2.0 * 32 * 32 * k;
      printf("%s random operands, %d waves/SIMD: %8.2f ms  %7.1f TF/s  %5.2f shader clocks per MFMA (wave 0)  shader clock %4.0f MHz\n",
             names[kind], wps, ms, flops / ms / 1e9, (double)hc[0] / (iters * 16.0), (double)hc[0] / ((double)hc[1] / 100.0));
    }
  return 0;
}
