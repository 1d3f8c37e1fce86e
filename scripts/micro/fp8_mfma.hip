// fp8 (e4m3) MFMA on gfx950: operand layout probe + sustained rate of v_mfma_f32_32x32x64_f8f6f4 and
// v_mfma_f32_32x32x16_fp8_fp8.   build: hipcc --offload-arch=gfx950 -O3 fp8_mfma.hip -o fp8_mfma
#include <hip/hip_runtime.h>
#include <hip/hip_fp8.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// layout hypothesis: lane l holds row (l & 31), K elements [32 * (l >> 5), +32) contiguous (32 bytes)
__global__ void probe64(const unsigned char* A /*[32][64]*/, const unsigned char* B /*[32][64] (N x K)*/, float* D /*[32][32]*/) {
  const int l = threadIdx.x, r = l & 31, g = l >> 5;
  v8i a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = *(const int*)(A + r * 64 + g * 32 + i * 4);
    b[i] = *(const int*)(B + r * 64 + g * 32 + i * 4);
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  // C/D layout of 32x32 MFMAs: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + r] = c[i];
}
__global__ void probe16(const unsigned char* A /*[32][16]*/, const unsigned char* B, float* D) {
  const int l = threadIdx.x, r = l & 31, g = l >> 5;
  long a = *(const long*)(A + r * 16 + g * 8), b = *(const long*)(B + r * 16 + g * 8);
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + r] = c[i];
}

template <int WHICH>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  v8i a[2], b[2];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int s = 0; s < 2; ++s)
    for (int i = 0; i < 8; ++i) {
      h = h * 1664525u + 1013904223u; a[s][i] = (int)(h & 0x77777777u);   // random e4m3 bytes without NaN patterns
      h = h * 1664525u + 1013904223u; b[s][i] = (int)(h & 0x77777777u);
    }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (WHICH == 64) acc[n] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[(u + n) & 1], b[u & 1], acc[n], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(((long*)&a[(u + n) & 1])[n & 3], ((long*)&b[u & 1])[u & 3], acc[n], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float e4m3_to_float(unsigned char v) {
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), e - 10);
  return s ? -f : f;
}

int main() {
  // ---- layout probes
  for (int K : {64, 16}) {
    std::vector<unsigned char> A(32 * K), B(32 * K);
    srand(3);
    for (auto& x : A) x = (unsigned char)(rand() & 0x77);
    for (auto& x : B) x = (unsigned char)((rand() & 0x77) | (rand() & 0x80));
    unsigned char *dA, *dB; float* dD;
    (void)hipMalloc(&dA, 32 * K); (void)hipMalloc(&dB, 32 * K); (void)hipMalloc(&dD, 4096);
    (void)hipMemcpy(dA, A.data(), 32 * K, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 32 * K, hipMemcpyHostToDevice);
    if (K == 64) probe64<<<1, 64>>>(dA, dB, dD); else probe16<<<1, 64>>>(dA, dB, dD);
    std::vector<float> D(1024);
    (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    // hypotheses: D[m][n] = sum_k A[m][k] B[n][k]  (a = first operand rows -> D rows)  or transposed
    double e1 = 0, e2 = 0, ref_max = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double r = 0;
        for (int k = 0; k < K; ++k) r += (double)e4m3_to_float(A[m * K + k]) * e4m3_to_float(B[n * K + k]);
        e1 = fmax(e1, fabs(D[m * 32 + n] - r));
        e2 = fmax(e2, fabs(D[n * 32 + m] - r));
        ref_max = fmax(ref_max, fabs(r));
      }
    printf("probe K=%d: max|D[m][n]-ref| = %.3g, max|D[n][m]-ref| = %.3g (ref max %.3g)\n", K, e1, e2, ref_max);
  }
  // ---- rates
  float* out; (void)hipMalloc(&out, 512 * 256 * 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int which : {64, 16}) {
    int iters = 100000;
    for (int rep = 0; rep < 2; ++rep) {
      (void)hipEventRecord(e0);
      if (which == 64) rate<64><<<512, 256>>>(out, iters); else rate<16><<<512, 256>>>(out, iters);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      double flops = 512.0 * 4 * iters * 16 * 2.0 * 32 * 32 * which;
      printf("mfma 32x32x%d fp8: %.2f ms  %.1f TF/s (random operands, 2 waves/SIMD)\n", which, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
