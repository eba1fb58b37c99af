// Helpers shared by the attention forward and backward kernels.
#pragma once
#include "attention.h"
#include "mma.h"

namespace cl {

template <typename T> struct AttnTraits;
template <> struct AttnTraits<bf16_t> { static constexpr int EB = 2; };
template <> struct AttnTraits<float> { static constexpr int EB = 4; };

// P^T fragment (B operand of the PV product) from one / two S^T accumulator fragments
template <typename T> struct PFrag;
template <> struct PFrag<bf16_t> {
  static constexpr int FRAGS = 2;  // kv fragments (16 keys each) per MFMA K step
  static __device__ __forceinline__ u32x4_t make(const f32x4_t* p) {
    u32x4_t r;
    r.x = pack2bf(p[0][0], p[0][1]); r.y = pack2bf(p[0][2], p[0][3]);
    r.z = pack2bf(p[1][0], p[1][1]); r.w = pack2bf(p[1][2], p[1][3]);
    return r;
  }
  // A operand from a kv-contiguous LDS row: keys {4g..4g+3} of both fragments
  static __device__ __forceinline__ u32x4_t read_a(uint32_t row_addr, int step, int g) {
    const u32x2_t lo = lds_read_b64(row_addr + (step * 32 + 4 * g) * 2);
    const u32x2_t hi = lds_read_b64(row_addr + (step * 32 + 16 + 4 * g) * 2);
    return u32x4_t{lo.x, lo.y, hi.x, hi.y};
  }
};
template <> struct PFrag<float> {
  static constexpr int FRAGS = 1;
  static __device__ __forceinline__ u32x4_t make(const f32x4_t* p) {
    return u32x4_t{__float_as_uint(p[0][0]), __float_as_uint(p[0][1]), __float_as_uint(p[0][2]),
                   __float_as_uint(p[0][3])};
  }
  static __device__ __forceinline__ u32x4_t read_a(uint32_t row_addr, int step, int g) {
    return lds_read_b128(row_addr + (step * 16 + 4 * g) * 4);
  }
};

}  // namespace cl
