// Hardware-layout probes (debug / test support): they exercise, in isolation, the two gfx950 layout
// assumptions the attention and GEMM kernels are built on — the MFMA 32x32x16 operand/result lane
// maps and the ds_read_b64_tr_b16 transpose gather — so a parity failure can be localised.
#include "rtv_common.h"
#include "rtv_internal.h"

namespace rtv {

// D[32][32] = A[32][16] . B[16][32]  (row-major bf16 inputs, f32 output), one wave.
__global__ void probe_mfma_kernel(const uint16_t* A, const uint16_t* B, float* D) {
  const int l = threadIdx.x, l31 = l & 31, g = l >> 5;
  u32x4 a, b;
  uint16_t av[8], bv[8];
  for (int j = 0; j < 8; ++j) {
    av[j] = A[l31 * 16 + g * 8 + j];    // A operand: row l31, k = g*8 + j
    bv[j] = B[(g * 8 + j) * 32 + l31];  // B operand: k = g*8 + j, col l31
  }
  for (int t = 0; t < 4; ++t) {
    a[t] = av[2 * t] | ((uint32_t)av[2 * t + 1] << 16);
    b[t] = bv[2 * t] | ((uint32_t)bv[2 * t + 1] << 16);
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b),
                                                acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * g;
    D[row * 32 + l31] = acc[r];
  }
}

// V: [64][128] bf16 row-major in global -> staged to LDS with the attention kernel's V swizzle ->
// out[db][lane][8] = the V^T fragment the attention kernel would feed to the MFMA for (kbk=0,s=0).
__global__ void probe_tr_kernel(const uint16_t* V, uint16_t* out, int kbk, int s) {
  __shared__ __attribute__((aligned(16))) char sV[64 * 256];
  const int lane = threadIdx.x;
  for (int id = lane; id < 64 * 16; id += 64) {
    int r = id >> 4, c = id & 15;
    *(u32x4*)(sV + r * 256 + ((c << 4) ^ ((r & 3) << 6))) = *(const u32x4*)(V + r * 128 + c * 8);
  }
  __syncthreads();
  const int g = lane >> 5, i16 = lane & 15, h16 = (lane >> 4) & 1;
  const int v_lane_off = (4 * g + (i16 >> 2)) * 256 + h16 * 32 + (i16 & 3) * 8;
  const int v_swz = i16 >> 2;
  const char* vrow = sV + (kbk * 32 + s * 16) * 256 + v_lane_off;
  for (int db = 0; db < 4; ++db) {
    const int col = ((db ^ v_swz) << 6);
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTV_LDS s16x4*)(vrow + col));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((RTV_LDS s16x4*)(vrow + 8 * 256 + col));
    for (int j = 0; j < 4; ++j) {
      out[(db * 64 + lane) * 8 + j] = (uint16_t)lo[j];
      out[(db * 64 + lane) * 8 + 4 + j] = (uint16_t)hi[j];
    }
  }
}

}  // namespace rtv

using namespace rtv;

extern "C" int rtv_probe_mfma(const void* A, const void* B, void* D, rtv_stream_t stream) {
  hipLaunchKernelGGL(probe_mfma_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)A,
                     (const uint16_t*)B, (float*)D);
  return check_launch("probe_mfma");
}

extern "C" int rtv_probe_tr(const void* V, void* out, int kbk, int s, rtv_stream_t stream) {
  hipLaunchKernelGGL(probe_tr_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)V,
                     (uint16_t*)out, kbk, s);
  return check_launch("probe_tr");
}
