// MFMA wrappers + raw LDS reads shared by the GEMM and attention kernels.
#pragma once
#include "common.h"

namespace cl {

// One "K step" = one 16-byte fragment per lane: 8 bf16 or 4 floats.
//   A operand: lane l holds row (l & 15), k-group (l >> 4)
//   B operand: lane l holds col (l & 15), k-group (l >> 4)
//   C/D      : lane l holds col (l & 15), rows 4*(l >> 4) + r, r = 0..3
// Any k partition that is consistent between A and B is valid, which is what
// lets the f32 path reuse the bf16 LDS image byte for byte.
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static constexpr int K = 32;  // contraction length per 16-byte fragment pair
  static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static constexpr int K = 16;
  static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
  }
};

// Raw LDS reads.  They are inline asm on purpose: while an LDS-DMA
// (global_load_lds) is in flight hipcc puts s_waitcnt vmcnt(0) in front of
// every ds_read it can see, which would serialise prefetch and MFMA.  The
// caller must issue lds_wait() before consuming the results.
__device__ __forceinline__ u32x4_t lds_read_b128(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ u32x2_t lds_read_b64(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

}  // namespace cl
