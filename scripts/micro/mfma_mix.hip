// How much MFMA issue time does one LDS read / one LDS-DMA displace?  Each wave loops over 16 independent
// v_mfma_f32_32x32x16_bf16 with R ds_read_b128 and D global_load_lds (16 B/lane, L2-resident source) issued behind
// them (one per MFMA slot).  Reports SIMD cycles per 16-MFMA group vs the MFMA-only loop.
// build: hipcc --offload-arch=gfx950 -O3 mfma_mix.hip -o mfma_mix ; run: mfma_mix <waves/SIMD 1|2>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int V>
struct IC {
  static constexpr int value = V;
};

template <int R, int D, int G>
__global__ __launch_bounds__(512) void k(const char* src, size_t span, int iters, float* out, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  bf16x8 a[2], b[2];
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 8; ++i) { a[s][i] = (__bf16)(0.01f * (lane + i + s)); b[s][i] = (__bf16)(0.02f * (lane - i) + s); }
  f32x16 acc[8];
  for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  u32x4 frag[16];
  for (int i = 0; i < 16; ++i) frag[i] = u32x4{0, 0, 0, 0};
  const char* s = src + lane * 16;
  const size_t blk_off = (size_t)blockIdx.x * 262144;  // every block walks its own region of the source span
  u32x4 stg[8];
  for (int i = 0; i < 8; ++i) stg[i] = u32x4{0, 0, 0, 0};
  char* dst = smem + wave * 8192;
  const unsigned rp = 65536 + wave * 4096 + lane * 16;
  __syncthreads();
  long long c0 = clock64();
  auto body = [&](int it, auto parc) {
    constexpr int PAR = decltype(parc)::value;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      acc[n & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n & 1], b[(n >> 1) & 1], acc[n & 7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (n < R) asm volatile("ds_read_b128 %0, %1" : "=v"(frag[n]) : "v"(rp + (unsigned)((n & 3) * 1024)));
      else if (n < R + D)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + ((blk_off + (size_t)(it * 16 + n) * 1024) & (span - 1))),
                                         (__attribute__((address_space(3))) void*)(dst + (n & 7) * 1024), 16, 0, 0);
      else if (n < R + D + G) {   // reload the staging set that was written to LDS in the previous iteration
        const int q = n - R - D;
        // plain load: the compiler tracks the outstanding load and places the counted vmcnt before the register is used
        stg[(PAR ^ 1) * 4 + (q & 3)] = *(const u32x4*)(s + ((blk_off + (size_t)(it * 16 + n) * 1024) & (span - 1)));
      } else if (n < R + D + 2 * G) {   // LDS write of the set loaded ONE iteration ago: retire it, this iteration's G loads stay in flight
        const int q = n - R - D - G;
        *(u32x4*)(dst + q * 1024 + lane * 16) = stg[PAR * 4 + (q & 3)];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (D) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  };
  for (int it = 0; it < iters; it += 2) {
    body(it, IC<0>{});
    body(it + 1, IC<1>{});

  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long c1 = clock64();
  float t = 0.f;
  for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) t += acc[n][r];
  for (int i = 0; i < 16; ++i) t += (float)(frag[i][0] & 1);
  for (int i = 0; i < 8; ++i) t += (float)(stg[i][1] & 1);
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
  if (blockIdx.x == 0 && lane == 0 && wave == 0) clk[0] = c1 - c0;
}

template <int R, int D, int G>
static void run(const char* src, size_t span, float* out, long long* clk, int wps, int iters) {
  hipFuncSetAttribute((const void*)k<R, D, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  k<R, D, G><<<256, 256 * wps, 131072>>>(src, span, iters, out, clk);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
  printf("span %5zu MiB waves/SIMD %d  reads %2d  dma %d  reg-staged %d : %7.1f cycles per 16 MFMA per wave (MFMA-only = %d)\n", span >> 20, wps, R, D, G,
         (double)h / iters, 512 * wps);
  fflush(stdout);
}

int main(int argc, char** argv) {
  int wps = 1, iters = 4000;
  size_t total = (size_t)1 << 30;
  char* src; float* out; long long* clk;
  hipMalloc(&src, total + 65536); hipMemset(src, 1, total + 65536);
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 8);
  for (size_t span : {(size_t)1 << 16, (size_t)1 << 30}) {   // L1/L2-resident source vs a 1 GiB walk (misses)
    run<0, 0, 0>(src, span, out, clk, wps, iters);
    run<12, 0, 0>(src, span, out, clk, wps, iters);
    run<0, 4, 0>(src, span, out, clk, wps, iters);
    run<0, 0, 4>(src, span, out, clk, wps, iters);
    run<12, 4, 0>(src, span, out, clk, wps, iters);
    run<8, 0, 4>(src, span, out, clk, wps, iters);
    run<8, 4, 0>(src, span, out, clk, wps, iters);
  }
  return 0;
}
