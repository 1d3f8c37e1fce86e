// What does an LDS read cost a wave that issues MFMAs back to back?  16 independent MFMAs (32x32x16 bf16) per iteration with R
// ds_read_b128 (inline asm, never waited for inside the loop; all four waves of a workgroup read, conflict-free addresses) and
// D buffer_load ... lds pieces (1 KiB each from an L2-resident 64 KiB buffer) placed one behind an MFMA each.
// build: hipcc --offload-arch=gfx950 -O3 mfma_lds_mix.hip -o mfma_lds_mix ; run: mfma_lds_mix <reads 0..16> <dma 0..4> [waves/SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDSP __attribute__((address_space(3)))

template <int R, int D>
__global__ __launch_bounds__(256) void mix_loop(float* out, const unsigned* src, int iters, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8 a[4], b[4];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; float x = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      h = h * 1664525u + 1013904223u; float y = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
      a[s][i] = (__bf16)(x * 3.f); b[s][i] = (__bf16)(y * 3.f);
    }
  f32x16 acc[16];
  for (int n = 0; n < 16; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned lds_addr = (unsigned)(uintptr_t)(LDSP char*)smem + wave * 16384 + (lane & 31) * 128 + ((((lane >> 5)) ^ ((lane >> 1) & 7)) << 4);
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  const unsigned voff = (blockIdx.x & 7) * 8192 + lane * 16;
  u32x4 junk[16];
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n >> 2], b[n & 3], acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (n < R) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(junk[n]) : "v"(lds_addr), "n"(0));
      if (n >= 12 && n - 12 < D)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDSP void*)(smem + 65536 + wave * 4096 + (n - 12) * 1024), 16, voff, (unsigned)(n - 12) * 1024, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (R > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (D > 0) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // three iterations (~1500 cycles) of slack: L2-hit latency must not show
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < 16; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  if (R > 0) for (int n = 0; n < R; ++n) s += __uint_as_float(junk[n][0]) * 1e-30f;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = t1 - t0; }
}

template <int R, int D>
void run(int wps) {
  int iters = 50000, blocks = 256 * wps;
  float* out; long long* clk; unsigned* src;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&clk, 16); (void)hipMalloc(&src, 1 << 20);
  (void)hipMemset(src, 0, 1 << 20);
  (void)hipFuncSetAttribute((const void*)mix_loop<R, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    mix_loop<R, D><<<blocks, 256, 81920>>>(out, src, iters, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long hh[2]; (void)hipMemcpy(hh, clk, 16, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 16.0 * 2.0 * 32 * 32 * 16;
    if (rep == 2)
      printf("reads/16 MFMA %2d  dma %d  waves/SIMD %d:  %.1f TF/s  cycles/MFMA(wave0) %.2f  clock %.0f MHz\n", R, D, wps, flops / ms / 1e9,
             (double)hh[0] / (iters * 16.0), (double)hh[0] / ((double)hh[1] / 100.0));
  }
}

int main(int argc, char** argv) {
  int r = argc > 1 ? atoi(argv[1]) : 8, d = argc > 2 ? atoi(argv[2]) : 0, wps = argc > 3 ? atoi(argv[3]) : 1;
  if (r == 0 && d == 0) run<0, 0>(wps);
  else if (r == 4 && d == 0) run<4, 0>(wps);
  else if (r == 8 && d == 0) run<8, 0>(wps);
  else if (r == 12 && d == 0) run<12, 0>(wps);
  else if (r == 16 && d == 0) run<16, 0>(wps);
  else if (r == 0 && d == 4) run<0, 4>(wps);
  else if (r == 8 && d == 4) run<8, 4>(wps);
  else printf("unsupported combination\n");
  return 0;
}
