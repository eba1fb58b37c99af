// Helpers shared by the attention forward and backward kernels.
#pragma once
#include "attention.h"
#include "mma.h"

namespace cl {

// Transposed tiles ([d rows][keys or queries contiguous]) are staged by global_load_lds, whose LDS
// image is lane-linear, i.e. plain row-major with 128- or 256-byte rows: read column-wise by the
// MFMA A-operand fragments that is an 8- to 16-way bank conflict.  Fix on the SOURCE side: the lane
// filling 16-byte slot q of row d fetches logical chunk q ^ tile_swz(d); readers look up chunk c of
// row d in slot c ^ tile_swz(d).  (d >> 1) & 7 makes rows {2j, 2j+1} share a slot but sit in opposite
// bank halves (128-byte rows): conflict-free for ds_read_b64 / b128 over 16 consecutive rows.
template <int ROW_BYTES> __device__ __forceinline__ int tile_swz(int d) {
  static_assert(ROW_BYTES % 128 == 0, "transposed tile rows must be whole 128-byte lines");
  return ROW_BYTES == 128 ? ((d >> 1) & 7) : (d & 7);
}

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
  // A operand from a kv-contiguous LDS row: keys {4g..4g+3} of both fragments.  `sw` is the row's
  // chunk swizzle (tile_swz): logical 16-byte chunk c of the row lives in slot c ^ sw.
  static __device__ __forceinline__ u32x4_t read_a(uint32_t row_addr, int step, int g, int sw) {
    const int c = step * 4 + (g >> 1), sub = (g & 1) * 8;
    const u32x2_t lo = lds_read_b64(row_addr + ((c ^ sw) * 16) + sub);
    const u32x2_t hi = lds_read_b64(row_addr + (((c + 2) ^ sw) * 16) + sub);
    return u32x4_t{lo.x, lo.y, hi.x, hi.y};
  }
};
template <> struct PFrag<float> {
  static constexpr int FRAGS = 1;
  static __device__ __forceinline__ u32x4_t make(const f32x4_t* p) {
    return u32x4_t{__float_as_uint(p[0][0]), __float_as_uint(p[0][1]), __float_as_uint(p[0][2]),
                   __float_as_uint(p[0][3])};
  }
  static __device__ __forceinline__ u32x4_t read_a(uint32_t row_addr, int step, int g, int sw) {
    return lds_read_b128(row_addr + (((step * 4 + g) ^ sw) * 16));
  }
};

}  // namespace cl
