// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of the
// real-time video hot path.  wave = 64 lanes, MFMA 32x32x16 (bf16 / f16), LDS 160 KiB/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

typedef uint16_t bf16_t;  // raw storage type for bf16 in global memory
typedef uint16_t f16_t;   // raw storage type for fp16 in global memory

#define RTV_LDS __attribute__((address_space(3)))
#define RTV_GLOBAL __attribute__((address_space(1)))

// ---------------------------------------------------------------- bf16 <-> f32
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even, same as torch's float->bfloat16 cast (NaN not special-cased:
// inputs on this path are finite).
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32 on gfx950 (RNE)
  bf16_t r;
  __builtin_memcpy(&r, &h, 2);
  return r;
}
// round a float through bf16 (used to reproduce the reference's bf16 rounding points)
__device__ __forceinline__ float round_bf16(float f) { return (float)(__bf16)f; }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  f16x2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
// NOTE: never __builtin_bit_cast a vector *element* expression (vec[j]) directly: hipcc (ROCm 7.2)
// reinterprets the whole vector's first bytes instead.  Go through a scalar copy, as here.
__device__ __forceinline__ void unpack_f16x2(uint32_t u, float& lo, float& hi) {
  f16x2 h = __builtin_bit_cast(f16x2, u);
  lo = (float)h[0];
  hi = (float)h[1];
}
__device__ __forceinline__ float f16_to_f32(f16_t v) {
  _Float16 h;
  __builtin_memcpy(&h, &v, 2);
  return (float)h;
}
__device__ __forceinline__ f16_t f32_to_f16(float f) {
  _Float16 h = (_Float16)f;  // RNE
  f16_t r;
  __builtin_memcpy(&r, &h, 2);
  return r;
}
__device__ __forceinline__ float round_f16(float f) { return (float)(_Float16)f; }

// unpack 8 bf16 (one 16-byte vector) to 8 floats
__device__ __forceinline__ void unpack_bf16x8(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack_bf16x8(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
  return v;
}

// ---------------------------------------------------------------- wave / block reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---------------------------------------------------------------- activations
__device__ __forceinline__ float gelu_tanh(float x) {
  // nn.GELU(approximate='tanh'): 0.5 x (1 + tanh( sqrt(2/pi) (x + 0.044715 x^3) ))
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float u = k0 * (x + k1 * x * x * x);
  // tanh(u) = 1 - 2 / (1 + exp(2u));   exp overflow -> inf -> tanh = 1 (fine)
  float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
  return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }

// XCD-aware bijective remap of a linear workgroup id (8 XCDs, block b runs on XCD b % 8):
// give each XCD a contiguous chunk of the logical id space.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int xcd = bid % nx, slot = bid / nx;
  int q = nwg / nx, r = nwg % nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

}  // namespace rtv
