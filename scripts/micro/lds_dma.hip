// LDS-side throughput of the global_load_lds DMA (gfx950) and its interference with ds_read_b128:
//   mode 0: 8 waves per CU only issue DMA (16 B/lane, source L2-resident)       -> DMA bytes/clk/CU
//   mode 1: 8 waves only issue ds_read_b128 (conflict-free)                      -> read bytes/clk/CU
//   mode 2: waves 0-3 DMA, waves 4-7 ds_read_b128                                -> both, concurrently
// build: hipcc --offload-arch=gfx950 -O3 lds_dma.hip -o lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ __launch_bounds__(512) void k(const char* src, int iters, int mode, unsigned* sink, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool do_dma = mode == 0 || (mode == 2 && wave < 4);
  const bool do_read = mode == 1 || (mode == 2 && wave >= 4);
  const char* s = src + (size_t)(blockIdx.x & 7) * 65536 + lane * 16;
  char* dst = smem + wave * 8192;
  u32x4 acc = {0, 0, 0, 0};
  __syncthreads();
  long long c0 = clock64();
  if (do_dma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + ((it * 8 + j) & 63) * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (do_read) {
    const char* rp = smem + 65536 + wave * 8192 + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(v) : "v"((unsigned)(size_t)(rp + j * 1024 - smem) + (unsigned)((it & 1) * 16384)));
        asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        acc ^= v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  long long c1 = clock64();
  if (acc[0] == 0x12345678u) sink[0] = acc[1];
  if (blockIdx.x == 0 && lane == 0) clk[wave] = c1 - c0;
}

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 20000;
  char* src; unsigned* sink; long long* clk;
  hipMalloc(&src, 8 * 65536 + 65536); hipMemset(src, 1, 8 * 65536 + 65536);
  hipMalloc(&sink, 16); hipMalloc(&clk, 64);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (int mode = 0; mode < 3; ++mode) {
    k<<<256, 512, 131072>>>(src, iters, mode, sink, clk);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, clk, 64, hipMemcpyDeviceToHost);
    double cyc_dma = (double)h[0], cyc_rd = (double)h[7];
    int nd = mode == 0 ? 8 : (mode == 2 ? 4 : 0), nr = mode == 1 ? 8 : (mode == 2 ? 4 : 0);
    double dma_bytes = (double)nd * iters * 8 * 1024, rd_bytes = (double)nr * iters * 8 * 1024;
    printf("mode %d: ", mode);
    if (nd) printf("DMA %.1f B/clk/CU (%d waves, %.0f cycles)  ", dma_bytes / cyc_dma, nd, cyc_dma);
    if (nr) printf("ds_read_b128 %.1f B/clk/CU (%d waves, %.0f cycles)", rd_bytes / cyc_rd, nr, cyc_rd);
    printf("\n");
  }
  return 0;
}
