// Building blocks shared by the transpose-free bf16 attention kernels (attention_tr.hip, attention_fwd40.hip):
// LDS tile geometry, pad-chunk initialisation, LDS transpose reads, counted waits.
#pragma once
#include <type_traits>
#include "attn_common.h"

namespace cl {
namespace {

template <int DH> struct Geo {
  static constexpr int CPR = DH / 8;             // 16-byte chunks of DATA per row
  static constexpr int KSTEPS = (CPR + 3) / 4;   // 32-deep MFMA steps over the head dim
  static constexpr int DN = (DH + 15) / 16;      // 16-wide output fragments over the head dim
  // LDS row pitch.  d_head 40 (the 64x64 level, where the attention time is): 80-byte rows put the ds_read_b128 row
  // fragments AND the ds_read_b64_tr_b16 column fragments 2-way on the banks (PMC: SQ_LDS_BANK_CONFLICT = 50 % of
  // SQ_LDS_IDX_ACTIVE, and LDS bandwidth is what bounds these kernels: ~20 fragment reads per 28 MFMAs per wave); a
  // 96-byte pitch (one pad chunk per row) makes both patterns conflict-free: chunk (6 r + g) mod 16 is a permutation over
  // a b128 lane group, and rows r = 0..7 start 24 banks apart -> eight disjoint 8-bank windows for the transpose reads.
  // The pad chunk is written once per kernel (zeros; ones for V in the ping-pong forward: it IS the softmax denominator
  // row) and masked out of the tile DMA.  160- and 320-byte pitches (d_head 80, 160) are left as they are.
  static constexpr int ROWB = DH == 40 ? 96 : DH * 2;   // bytes per LDS tile row (pitch)
  static constexpr int CPRP = ROWB / 16;         // chunks per LDS row incl. padding
  static constexpr int TILE = 64 * ROWB;         // a 64-row operand tile
  static constexpr int TI = CPRP;                // DMA instructions (64 lanes x 16 B) per tile
};

// write the pad chunk (bytes [16 CPR, ROWB) of every row) of `ntile` consecutive tiles; odd tiles get `odd` instead of
// `even` (K / V or Q / dO pairs).  No-op when the pitch has no padding.  Callers synchronise before the first read.
template <int DH> __device__ __forceinline__ void init_pads(char* tiles, int ntile, uint32_t even, uint32_t odd, int tid,
                                                            int nthreads) {
  using G = Geo<DH>;
  if constexpr (G::CPRP > G::CPR) {
    for (int i = tid; i < ntile * 64; i += nthreads) {
      const uint32_t v = ((i >> 6) & 1) ? odd : even;
      *reinterpret_cast<uint4*>(tiles + (long)i * G::ROWB + G::CPR * 16) = make_uint4(v, v, v, v);
    }
  }
}

// Plain fp32 VALU instructions issue at 4 cycles per wave64 on gfx950 (measured: the softmax / dS arithmetic, not
// the matrix pipe, bounds these kernels); v_pk_{fma,mul,add}_f32 do two lanes' worth per issue slot.
typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <int IMM> __device__ __forceinline__ u32x2_t tr_read(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM) : "memory");
  return v;
}
// one 8-deep A-operand fragment = transpose reads of rows r0+4g+j and r0+16+4g+j (r0 = 32 * STEP)
template <int ROWB, int STEP> __device__ __forceinline__ u32x4_t tr_frag(uint32_t addr) {
  const u32x2_t lo = tr_read<STEP * 32 * ROWB>(addr), hi = tr_read<STEP * 32 * ROWB + 16 * ROWB>(addr);
  return u32x4_t{lo.x, lo.y, hi.x, hi.y};
}

// the same with a compile-time byte offset folded into the instruction immediates (one address register for many fragments)
template <int ROWB, int STEP, int OFF> __device__ __forceinline__ u32x4_t tr_frag_off(uint32_t addr) {
  const u32x2_t lo = tr_read<STEP * 32 * ROWB + OFF>(addr), hi = tr_read<STEP * 32 * ROWB + 16 * ROWB + OFF>(addr);
  return u32x4_t{lo.x, lo.y, hi.x, hi.y};
}

typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
template <int OFF> __device__ __forceinline__ u32x4_t lds_read_b128_off(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// a value produced by an asynchronous LDS read: every use must follow the wait this is placed after
__device__ __forceinline__ void pin(u32x4_t& v) { asm volatile("" : "+v"(v)); }
template <int N> __device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}


}  // namespace
}  // namespace cl
