// Shared device helpers for the CtrLoRA gfx950 kernels.
//
// Storage type T is either bf16 (raw uint16_t bits, "perf mode") or float
// ("parity mode": the f32-input MFMA is an exact fmaf chain on CDNA4, so the
// whole engine can be checked against the fp32 oracle to ~1e-5).  All
// accumulation, normalisation statistics and softmax run in fp32 regardless.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cl {

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

enum { CL_OK = 0, CL_EINVAL = 1, CL_ELAUNCH = 2 };
enum { CL_BF16 = 0, CL_F32 = 1 };

__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// fp32 -> bf16, round-to-nearest-even: the __bf16 casts lower to ONE v_cvt_pk_bf16_f32 per pair on
// gfx950 (a hand-rolled integer RNE costs ~5 VALU per element, which made the attention backward
// and every epilogue VALU-bound).
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f2bf(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  bf16x2_t v; v.x = (__bf16)lo; v.y = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }

// 8 consecutive elements <-> float[8]; p must be 16-byte aligned (bf16) /
// 16-byte aligned (float, two float4).
__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
  uint4 r = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
__device__ __forceinline__ void load8(const float* p, float v[8]) {
  float4 a = reinterpret_cast<const float4*>(p)[0];
  float4 b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
  uint4 r;
  r.x = pack2bf(v[0], v[1]); r.y = pack2bf(v[2], v[3]);
  r.z = pack2bf(v[4], v[5]); r.w = pack2bf(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = r;
}
__device__ __forceinline__ void store8(float* p, const float v[8]) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void load4(const bf16_t* p, float v[4]) {
  uint2 r = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
__device__ __forceinline__ void load4(const float* p, float v[4]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}
__device__ __forceinline__ void store4(bf16_t* p, const float v[4]) {
  uint2 r; r.x = pack2bf(v[0], v[1]); r.y = pack2bf(v[2], v[3]);
  *reinterpret_cast<uint2*>(p) = r;
}
__device__ __forceinline__ void store4(float* p, const float v[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + __expf(-z)); }
// d silu(z) / dz
__device__ __forceinline__ float dsilu_f(float z) {
  float s = 1.0f / (1.0f + __expf(-z));
  return s * (1.0f + z * (1.0f - s));
}
// exact (erf) GELU, as torch F.gelu default (ldm/modules/attention.py:56)
__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float dgelu_f(float x) {
  float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// 64-lane butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// async global -> LDS, 16 bytes per lane.  LDS destination is
// (wave-uniform base) + lane*16 (hardware adds the lane offset); the global
// source address is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the hipError_t behind the most recent CL_ELAUNCH (diagnostics; read through cl_last_hip_error())
inline int g_last_hip_error = 0;
#define CL_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) { cl::g_last_hip_error = (int)e__; return cl::CL_ELAUNCH; } \
  } while (0)

}  // namespace cl
