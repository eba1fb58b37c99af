// bf16 fused attention for gfx950 without materialised transposes.
//
// CrossAttention.forward (ldm/modules/attention.py:163-194) and its data gradient, flash style
// (fp32 scores / softmax as the reference forces at :171-179, log-sum-exp saved for the backward).
// The round-0 kernels (attention_fwd.hip / attention_bwd.hip, still used for the fp32 parity mode)
// needed V^T, Q^T, dO^T and K^T copies in HBM for the second MFMA of every stage, because an MFMA
// operand wants the contraction index contiguous.  Here every tile goes HBM->LDS row-major exactly
// as the projections left it (global_load_lds), and the "transposed" operands are built by
// ds_read_b64_tr_b16: within a 16-lane group, lane i receives element (i & 3) of the 8 bytes addressed
// by lane 4j + (i >> 2), j = 0..3.  With lanes 4j..4j+3 pointing at 16 consecutive head-dim columns of
// key/query row r0 + j, lane i gets column i of rows r0..r0+3, i.e. one half of an A-operand
// fragment whose contraction index runs over keys/queries.  So each tile is read twice from LDS
// (row-wise with ds_read_b128 for Q.K^T-type products, column-wise with the transpose read for
// P.V-type products) and only once from HBM.
//
// k-index bookkeeping: the B operand of the second product comes straight out of the first product's
// accumulators (C layout: lane = column, rows 4g+r), so a 32-deep contraction step made of two
// 16-row fragments f0, f1 gives lane group g the rows {16 f0 + 4g + r} U {16 f1 + 4g + r}; the
// transpose reads fetch exactly those rows (row 4g + j of each fragment), any consistent
// permutation of the contraction index being valid for an MFMA.
//
// Backward: dK/dV kernel = 64*KF keys per workgroup (KF key fragments per wave, so every LDS
// fragment feeds KF MFMAs), loop over 64-query tiles {Q, dO, lse, delta} double buffered;
// dQ kernel = 64*QF queries per workgroup, loop over 64-key tiles {K, V}.  log-sum-exp and delta
// ride in the tile (LDS broadcast reads) instead of per-fragment global loads.
#include <type_traits>
#include "attn_common.h"
#include "attn_tr_util.h"

// Probe builds only (tools/build_probes.sh attn_bwd_abl): -DATTN_BWD_ABL=n removes one ingredient of the two d_head-40 fold
// backward kernels so that its share of the time can be measured (results are wrong by construction):
//   1 no exp2   2 no MFMA (operands kept alive)   3 no LDS fragment reads   4 no DMA wait / barrier   5 no cvt_pk (P, dS not packed)
#ifndef ATTN_BWD_ABL
#define ATTN_BWD_ABL 0
#endif
#if ATTN_BWD_ABL == 2
#define BWD_MMA(a, b, c) asm volatile("" ::"v"(a), "v"(b))
#else
#define BWD_MMA(a, b, c) Mma<bf16_t>::run(a, b, c)
#endif
#if ATTN_BWD_ABL == 3
#define BWD_RD128 abl_rd128
#define BWD_TRF abl_trf
#else
#define BWD_RD128 lds_read_b128_off
#define BWD_TRF tr_frag_off
#endif
#if ATTN_BWD_ABL == 5
#define BWD_PACK(p) abl_pack(p)
#else
#define BWD_PACK(p) PFrag<bf16_t>::make(p)
#endif
#if ATTN_BWD_ABL == 1
#define BWD_EXP2(x) (x)
#else
#define BWD_EXP2(x) __builtin_amdgcn_exp2f(x)
#endif

namespace cl {
#if ATTN_BWD_ABL == 3
template <int OFF> __device__ __forceinline__ u32x4_t abl_rd128(uint32_t a) { u32x4_t v = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; asm volatile("" : "+v"(v) : "v"(a)); return v; }
template <int ROWB, int STEP, int OFF> __device__ __forceinline__ u32x4_t abl_trf(uint32_t a) { u32x4_t v = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; asm volatile("" : "+v"(v) : "v"(a)); return v; }
#endif
#if ATTN_BWD_ABL == 5
__device__ __forceinline__ u32x4_t abl_pack(const f32x4_t* p) {
  return u32x4_t{__float_as_uint(p[0][0]), __float_as_uint(p[0][2]), __float_as_uint(p[1][0]), __float_as_uint(p[1][2])};
}
#endif

namespace {

// Staging of 64-row tiles of a [rows, ld] bf16 matrix (head slice already applied to `base`); wave w issues
// DMA instructions w, w+4, ...  PMC showed the forward VALU-bound (SQ_ACTIVE_INST_VALU ~ 90 % of the kernel)
// with ~30 % of the vector instructions spent on per-tile DMA address generation (divide by chunks-per-row,
// clamp, 64-bit multiply), so the per-lane (row, chunk) decomposition is done ONCE and a full tile costs one
// 64-bit add per instruction; only the ragged last tile clamps rows.
template <int DH> struct TileDma {
  static constexpr int TI = Geo<DH>::TI, NJ = (TI + 3) / 4, CPR = Geo<DH>::CPR, CPRP = Geo<DH>::CPRP;
  int r[NJ], cc16[NJ];
  bool real[NJ];                                   // false: this lane's LDS slot is a pad chunk (left alone)
  __device__ __forceinline__ void init(int wave, int lane) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = (wave + 4 * j) * 64 + lane;
      r[j] = c / CPRP;
      const int col = c - r[j] * CPRP;
      real[j] = col < CPR;
      cc16[j] = (real[j] ? col : 0) * 16;
    }
  }
  __device__ __forceinline__ void offsets(long ld_bytes, int (&off)[NJ]) const {
#pragma unroll
    for (int j = 0; j < NJ; ++j) off[j] = r[j] * (int)ld_bytes + cc16[j];
  }
  // Whole 64-row tile whose pad chunk (16 bytes per row) comes from a second array `pad` with `pad_stride` bytes per row:
  // ONE instruction per 1 KiB piece, the source address selected per lane
  __device__ __forceinline__ void issue_with_pad(const char* base, long ld_bytes, const int (&off)[NJ], int row0, char* dst,
                                                 int wave, const char* pad, int pad_stride) const {
    const char* tb = base + (long)row0 * ld_bytes;      // wave-uniform
    const char* pb = pad + (long)row0 * pad_stride;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (wave + 4 * j < TI) {
        const char* src = real[j] ? tb + off[j] : pb + r[j] * pad_stride;
        glds16(src, dst + (wave + 4 * j) * 1024);
      }
  }
  __device__ __forceinline__ void issue(const char* base, long ld_bytes, const int (&off)[NJ], int row0, int nrows,
                                        char* dst, int wave) const {
    const char* tb = base + (long)row0 * ld_bytes;      // wave-uniform
    if (row0 + 64 <= nrows) {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (wave + 4 * j < TI && real[j]) glds16(tb + off[j], dst + (wave + 4 * j) * 1024);
    } else {
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        if (wave + 4 * j < TI && real[j]) {
          const int rr = min(row0 + r[j], nrows - 1);
          glds16(base + (long)rr * ld_bytes + cc16[j], dst + (wave + 4 * j) * 1024);
        }
    }
  }
};

// x as three bf16 pieces (hi + mid + lo = x to ~2^-24 relative): words {hi | mid << 16, lo}
__device__ __forceinline__ void split3_bf16(float x, uint32_t& w0, uint32_t& w1) {
  const uint32_t hi = pack2bf(x, 0.f) & 0xffffu;
  const float r1 = x - __uint_as_float(hi << 16);
  const uint32_t mid = pack2bf(r1, 0.f) & 0xffffu;
  const float r2 = r1 - __uint_as_float(mid << 16);
  const uint32_t lo = pack2bf(r2, 0.f) & 0xffffu;
  w0 = hi | (mid << 16); w1 = lo;
}
// pad chunk (1, 1, 1, 0, 0, 0, 0, 0): meets the three pieces above in the contraction
__device__ __forceinline__ u32x4_t ones3_chunk() { return u32x4_t{0x3F803F80u, 0x00003F80u, 0u, 0u}; }
template <int DH> __device__ __forceinline__ void init_pads_ones3(char* tiles, int ntile, int tid, int nthreads) {
  using G = Geo<DH>;
  for (int i = tid; i < ntile * 64; i += nthreads)
    *reinterpret_cast<uint4*>(tiles + (long)i * G::ROWB + G::CPR * 16) = make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u);
}

}  // namespace

// =============================================================================== forward
// TAIL: Nkv is not a multiple of 64 (cross-attention's 77 keys): only that instantiation carries key masking
template <int DH, int QW, bool TAIL>
__global__ __launch_bounds__(256, 2) void attn_fwd_tr_kernel(AttnFwdArgs p, const void* __restrict__ V, long ldv) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, KSTEPS = G::KSTEPS, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE;
  constexpr int STAGE = 2 * TILE;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages + 64 bytes of slack

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lq = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (64 * QW) + wave * (16 * QW);
  const float sl2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

  u32x4_t qf[QW][KSTEPS];
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    const int row = min(q0 + f * 16 + lq, p.N - 1);
    const char* qp = (const char*)p.Q + (((long)b * p.N + row) * p.ldq + (long)h * DH) * 2;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = 4 * ks + g;
      qf[f][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(qp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
  }
  const char* kbase = (const char*)p.K + ((long)b * p.Nkv * p.ldk + (long)h * DH) * 2;
  const char* vbase = (const char*)V + ((long)b * p.Nkv * ldv + (long)h * DH) * 2;

  f32x4_t ot[DN][QW];
#pragma unroll
  for (int i = 0; i < DN; ++i)
#pragma unroll
    for (int f = 0; f < QW; ++f) ot[i][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run[QW], l_run[QW];
#pragma unroll
  for (int f = 0; f < QW; ++f) { m_run[f] = -1e30f; l_run[f] = 0.f; }   // finite: exp2(m_run - m_new) must not see inf - inf

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // b128: row lq, chunk 4 ks + g; a lane past the head dim re-reads the last data chunk against the zero slots of its Q
  // fragment (+-0 added: bit-identical) instead of being masked off -- see the dK / dV kernel
  uint32_t krow[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) krow[ks] = lq * ROWB + min(4 * ks + g, CPR - 1) * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;  // transpose read: row 4g+j, cols 4q
  const int ntiles = (p.Nkv + 63) / 64;
  TileDma<DH> dma; dma.init(wave, lane);
  int koff[TileDma<DH>::NJ], voff[TileDma<DH>::NJ];
  dma.offsets(p.ldk * 2, koff); dma.offsets(ldv * 2, voff);
  init_pads<DH>(smem, 4, 0u, 0u, tid, 256);           // 2 stages x {K, V}
  dma.issue(kbase, p.ldk * 2, koff, 0, p.Nkv, smem, wave);
  dma.issue(vbase, ldv * 2, voff, 0, p.Nkv, smem + TILE, wave);
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles) {
      dma.issue(kbase, p.ldk * 2, koff, (t + 1) * 64, p.Nkv, smem + (buf ^ 1) * STAGE, wave);
      dma.issue(vbase, ldv * 2, voff, (t + 1) * 64, p.Nkv, smem + (buf ^ 1) * STAGE + TILE, wave);
    }
    const uint32_t kt = lds0 + buf * STAGE, vt = kt + TILE;

    // ---- S^T = K . Q^T   (rows = keys 16 kf + 4g + r, col = query lq)
    // (software-pipelining the LDS reads one fragment ahead was measured: no gain -- the kernel is VALU-bound
    // and the other resident waves already cover LDS latency -- and it cost an occupancy step in registers)
    f32x4_t st[4][QW];
    uint32_t kr[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) kr[ks] = kt + krow[ks];
    static_for<0, 4>([&](auto KF_) {
      constexpr int kf = decltype(KF_)::value;
      u32x4_t ka[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) ka[ks] = lds_read_b128_off<kf * 16 * ROWB>(kr[ks]);
      lds_wait();
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        st[kf][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) Mma<bf16_t>::run(ka[ks], qf[f][ks], st[kf][f]);
      }
    });
    // ---- online softmax: lane owns query lq of each q fragment; max on the raw scores, one fma + exp2 each
    const int kv0 = t * 64;
    {
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        float mx = -INFINITY;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (TAIL) { if (kv0 + kf * 16 + 4 * g + r >= p.Nkv) st[kf][f][r] = -INFINITY; }
            mx = fmaxf(mx, st[kf][f][r]);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[f], mx * sl2);      // sl2 > 0
        const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
        m_run[f] = m_new;
        float ls = 0.f;
#pragma unroll
        for (int kf = 0; kf < 4; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(st[kf][f][r], sl2, -m_new));
            st[kf][f][r] = e;
            ls += e;
          }
        l_run[f] = l_run[f] * alpha + ls;
#pragma unroll
        for (int i = 0; i < DN; ++i) ot[i][f] *= alpha;
      }
    }
    // ---- O^T += V^T . P^T   (A = V^T via transpose reads of the row-major V tile)
    u32x4_t pb[2][QW];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        f32x4_t tmp[2] = {st[2 * s][f], st[2 * s + 1][f]};
        pb[s][f] = PFrag<bf16_t>::make(tmp);
      }
    const uint32_t tb = vt + troff;
    static_for<0, DN>([&](auto I_) {
      constexpr int i = decltype(I_)::value;
      u32x4_t va[2];
      va[0] = tr_frag_off<ROWB, 0, i * 32>(tb);
      va[1] = tr_frag_off<ROWB, 1, i * 32>(tb);
      lds_wait();
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[s], pb[s][f], ot[i][f]);
    });
  }

  // ---- epilogue: normalise, store O rows, log-sum-exp
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    float l = l_run[f];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int row = q0 + f * 16 + lq;
    if (row < p.N) {
      bf16_t* op = reinterpret_cast<bf16_t*>(p.O) + ((long)b * p.N + row) * p.ldo + (long)h * DH;
#pragma unroll
      for (int i = 0; i < DN; ++i) {
        const int d0 = i * 16 + 4 * g;
        if (d0 < DH) {
          float v[4] = {ot[i][f][0] * inv, ot[i][f][1] * inv, ot[i][f][2] * inv, ot[i][f][3] * inv};
          store4(op + d0, v);
        }
      }
      if (p.LSE && g == 0) p.LSE[((long)b * p.H + h) * p.lse_stride + row] = m_run[f] + __builtin_amdgcn_logf(l);
    }
  }
}

// =============================================================================== forward, ping-pong schedule (description)
// Self-attention at the 64x64 / 32x32 levels (N = 4096 / 1024 keys, d_head 40 / 80) is where the attention time
// goes, and there the kernel above is bound by neither pipe: per 32-query x 64-key step a wave issues 28 MFMAs
// (448 matrix-pipe cycles) and ~175 VALU instructions (softmax), but the tile barrier keeps every wave of the
// workgroup in the same phase, so the matrix pipe idles while all waves do softmax and vice versa (measured:
// ~1400 cycles per wave-step).  This kernel runs 8 waves per workgroup (256 queries) as two groups of four that
// are ONE phase apart -- waves w and w + 4 share a SIMD -- so that on every SIMD one wave is in its matrix phase
//     M(u) = O^T += V^T P^T of tile u-1, then S^T = K Q^T of tile u     (28 MFMAs, LDS fragment reads pipelined)
// while its partner is in its vector phase
//     V(u) = online softmax of tile u                                     (VALU only, no LDS, no memory)
// and a workgroup barrier separates the phases.  Schedule in barrier intervals (group A = waves 0-3, B = 4-7):
//     interval 2u   : A: M(u)   B: V(u-1)      interval 2u+1 : A: V(u)   B: M(u)
// K/V tiles ride a 3-stage LDS ring: tile u is read in intervals 2u .. 2u+3 (K by the two M(u), V by the two
// M(u+1)), so its stage is refilled with tile u+3 at the start of interval 2u+4 (every wave issues its share of
// the DMA there) and every wave drains its own DMA (vmcnt 0) before the barrier that ends interval 2u+5.
// Softmax VALU diet: (a) the running maximum moves only when some score exceeds it by more than RESCALE_THR
// (in log2 units) -- the common step has no cross-lane traffic and no rescale of O; (b) for d_head 40 the
// softmax denominator is produced by the matrix pipe: the V^T operand has spare rows (40..47), row 40 is forced
// to ones, so O^T[40, q] = sum_k P[q, k] of exactly the bf16 P that multiplies V.
// (The 16x16x32-only kernel this schedule was first built as -- round 2 -- is gone: the 32x32x16 form below replaced it in
// round 3, and round 4's pre-scaled-Q forward, attention_fwd40.hip, takes the d_head-40 self-attentions.)

// =============================================================================== forward, ping-pong + 32x32 Q.K^T
// Same schedule as attn_fwd_pp_kernel (two wave groups one phase apart, 3-stage K/V ring, lazy maximum, softmax
// denominator out of the matrix pipe for d_head 40), with S^T = K Q^T formed by v_mfma_f32_32x32x16_bf16:
//   * d_head 40 costs a 48-deep walk (3 instructions of 32 cycles per 32 keys x 32 queries) instead of the 64-deep walk of
//     the 16x16x32 form (8 instructions of 16 cycles): 192 instead of 256 matrix cycles per 64-key step, and 6 instead of
//     8 ds_read_b128 of K per wave and step (one fragment feeds a 32 x 32 tile);
//   * a lane then holds 32 scores of ONE query (lane & 31; the two half-waves split the keys), so the running maximum
//     is per lane and the row maximum needs one cross-lane step, in the rare rescale branch only;
//   * P^T changes hands to the 16x16x32 P.V product (d_head 40 -> 48 output rows: the 32x32 form would pad to 64) by
//     v_permlane16_swap: C registers (2k, 2k+1) are packed to bf16 pairs P_k; swapping lane rows 1 / 3 of P_a with rows
//     0 / 2 of P_b puts the two query halves of the 32-wide tile into two 16-query B operands whose four 16-lane groups hold
//     MFMA rows {b, .., b+3, b+16, .., b+19} with b = 0, 8, 4, 12; the K fragment rows are permuted (bits 2 and 3 exchanged) so that
//     these are the LDS key rows {4g .. 4g+3, 16+4g ..} the V^T transpose reads address.

template <int DH, int LA = 3>
__global__ __launch_bounds__(512, 2) void attn_fwd_hyb_kernel(AttnFwdArgs p, const void* __restrict__ V, long ldv,
                                                              int nqb, int remap) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE;
  constexpr int STAGE = 2 * TILE, QW = 2;
  constexpr int NK = (DH + 15) / 16;           // 16-deep steps of the 32x32x16 product over the head dim
  constexpr bool ONES = (DH % 16) != 0;
  constexpr bool PADONES = ONES && G::CPRP > G::CPR;
  static_assert(!ONES || PADONES, "spare V^T rows come from the padded LDS pitch");
  static_assert(2 * NK <= G::CPRP, "K fragment reads stay inside the LDS row");
  constexpr int LROW = DH % 16;
  constexpr float RESCALE_THR = 6.0f;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int g = lane >> 4, lq = lane & 15, l31 = lane & 31, hi = lane >> 5;
  int bh, qb;
  {
    const int id = blockIdx.x;
    if (remap) { const int xcd = id & 7, slot = id >> 3; bh = xcd + 8 * (slot / nqb); qb = slot - (slot / nqb) * nqb; }
    else { bh = id / nqb; qb = id - bh * nqb; }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qb * 256 + wave * 32;
  const float sl2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

  for (int i = tid; i < (16 * ROWB + 64) / 4; i += 512) reinterpret_cast<uint32_t*>(smem + 3 * STAGE)[i] = 0u;
  init_pads<DH>(smem, 6, 0u, 0x3F803F80u, tid, 512);   // K pad = 0 (meets Q zeros); V pad = 1.0: rows DH.. of V^T

  // Q as the B operand of the 32x32x16 product: col = query l31, k = 16 j + 8 hi .. +7
  u32x4_t qh[NK];
  {
    const char* qp = (const char*)p.Q + (((long)b * p.N + q0 + l31) * p.ldq + (long)h * DH) * 2;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int c = 2 * j + hi;
      qh[j] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(qp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
  }
  const char* kbase = (const char*)p.K + ((long)b * p.Nkv * p.ldk + (long)h * DH) * 2;
  const char* vbase = (const char*)V + ((long)b * p.Nkv * ldv + (long)h * DH) * 2;

  f32x4_t ot[DN][QW];
#pragma unroll
  for (int i = 0; i < DN; ++i)
#pragma unroll
    for (int f = 0; f < QW; ++f) ot[i][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;           // of query l31 (both half-waves keep the same maximum)
  f32x16_t st[2];                              // S^T tiles: keys 32 s + (r & 3) + 8 (r >> 2) + 4 hi, query l31
  u32x4_t pb[2][QW];

  constexpr int CPRP = G::CPRP, NJ = (CPRP + 7) / 8;
  int koff[NJ], voff[NJ];
  bool real[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + 8 * j) * 64 + lane, r = c / CPRP, col = c - r * CPRP;
    real[j] = col < CPR;
    const int cc = (real[j] ? col : 0) * 16;
    koff[j] = r * (int)(p.ldk * 2) + cc;
    voff[j] = r * (int)(ldv * 2) + cc;
  }
  auto issue = [&](int t, int stage) {
    const char* kb = kbase + (long)t * 64 * p.ldk * 2;
    const char* vb = vbase + (long)t * 64 * ldv * 2;
    char* dst = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (wave + 8 * j < CPRP && real[j]) {
        glds16(kb + koff[j], dst + (wave + 8 * j) * 1024);
        glds16(vb + voff[j], dst + TILE + (wave + 8 * j) * 1024);
      }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // MFMA row i of the S^T tile takes LDS key row pi(i) = i with bits 2 and 3 exchanged: lane group g of the P operand then
  // holds LDS rows {4g .. 4g+3, 16+4g .. 16+4g+3}, i.e. the V^T transpose reads address the same CONSECUTIVE rows as in the
  // 16x16 kernel (8 rows of a half-wave 24 banks apart: conflict-free at the 96-byte pitch; with pi = identity the rows
  // {0-3, 8-11} collided pairwise: SQ_LDS_BANK_CONFLICT was 50 % of SQ_LDS_IDX_ACTIVE, profiles/r03_final/pmc_attn_fwd_hyb.txt)
  const int kr = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const uint32_t krow = kr * ROWB + hi * 16;                                   // b128: key row pi(l31) of a 32-key tile, chunk 2 j + hi
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;  // key rows 4g + j (+16)
  const int nt = p.Nkv / 64;

  // ---- matrix phase: PV of the previous tile (PREV), then the two 32-key S^T tiles of this one
  auto phaseM = [&](auto PREVc, uint32_t kt, uint32_t vt) {
    constexpr bool PREV = decltype(PREVc)::value;
    constexpr int NV = PREV ? DN : 0, NG = NV + 2;
    u32x4_t va[LA][2], ka[LA][NK];
    auto cnt_of = [](int j) constexpr { return j < NV ? 4 : NK; };
    auto read_group = [&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      if constexpr (J < NV) {
        va[J % LA][0] = tr_frag<ROWB, 0>(vt + troff + J * 32);
        va[J % LA][1] = tr_frag<ROWB, 1>(vt + troff + J * 32);
      } else if constexpr (J < NG) {
        constexpr int s = J - NV;
        static_for<0, NK>([&](auto Kc) {
          constexpr int j = decltype(Kc)::value;
          ka[s % LA][j] = lds_read_b128_off<s * 32 * ROWB + j * 32>(kt + krow);
        });
      }
    };
    static_for<0, LA - 1>([&](auto Jc) { read_group(Jc); });
    __builtin_amdgcn_s_setprio(1);
    static_for<0, NG>([&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      read_group(std::integral_constant<int, J + LA - 1>{});
      constexpr int pending = [&]() constexpr { int n = 0; for (int k = J + 1; k < J + LA && k < NG; ++k) n += cnt_of(k); return n; }();
      lgkm_wait<(pending > 15 ? 15 : pending)>();
      if constexpr (J < NV) {
        pin(va[J % LA][0]); pin(va[J % LA][1]);
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[J % LA][0], pb[0][f], ot[J][f]);
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[J % LA][1], pb[1][f], ot[J][f]);
      } else {
        constexpr int s = J - NV;
#pragma unroll
        for (int j = 0; j < NK; ++j) pin(ka[s % LA][j]);
        f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NK; ++j)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka[s % LA][j]),
                                                        __builtin_bit_cast(bf16x8_t, qh[j]), acc, 0, 0, 0);
        st[s] = acc;
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __builtin_amdgcn_s_setprio(0);
  };

  // ---- vector phase: online softmax of st -> pb (registers only)
  auto phaseV = [&]() {
    float mx = st[0][0];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[s][r]);
    const bool need = mx * sl2 > m_run + RESCALE_THR;
    if (__any(need)) {   // wave-uniform, rare after the first tiles
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));                 // the other half-wave holds the other keys of this query
      const float m_new = fmaxf(m_run, mx * sl2);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        const float af = __shfl(alpha, 16 * f + lq, 64);      // O^T fragment f, column lq = query 16 f + lq
#pragma unroll
        for (int i = 0; i < DN; ++i) ot[i][f] *= af;
      }
    }
    float ls = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint32_t pk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const f32x2_t x = f32x2_t{st[s][2 * k], st[s][2 * k + 1]} * sl2 - m_run;
        const float e0 = __builtin_amdgcn_exp2f(x.x), e1 = __builtin_amdgcn_exp2f(x.y);
        if constexpr (!ONES) ls += e0 + e1;
        pk[k] = pack2bf(e0, e1);
      }
      // (P0,P2) (P1,P3) (P4,P6) (P5,P7): rows 1 / 3 of the first <-> rows 0 / 2 of the second
      const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0], pk[2], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(pk[1], pk[3], false, false);
      const auto s2 = __builtin_amdgcn_permlane16_swap(pk[4], pk[6], false, false);
      const auto s3 = __builtin_amdgcn_permlane16_swap(pk[5], pk[7], false, false);
      pb[s][0] = u32x4_t{s0[0], s1[0], s2[0], s3[0]};
      pb[s][1] = u32x4_t{s0[1], s1[1], s2[1], s3[1]};
    }
    if constexpr (!ONES) l_run += ls;
  };

  issue(0, 0);
  issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int sk = 0;
  if (grp == 1) __builtin_amdgcn_s_setprio(1);         // (phaseM ends with setprio(0): the static form is re-armed below)
  auto pv_tail = [&](uint32_t vt) {
    u32x4_t va[2];
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      va[0] = tr_frag<ROWB, 0>(vt + troff + i * 32); va[1] = tr_frag<ROWB, 1>(vt + troff + i * 32);
      lds_wait();
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[s], pb[s][f], ot[i][f]);
    }
  };
  if (grp == 0) {
    int sp = 2;
    for (int u = 0; u < nt; ++u) {
      const int sn = (sk == 2) ? 0 : sk + 1;
      if (u >= 1 && u + 1 < nt) issue(u + 1, sn);
      if (u == 0) phaseM(std::false_type{}, lds0 + sk * STAGE, 0u);
      else phaseM(std::true_type{}, lds0 + sk * STAGE, lds0 + sp * STAGE + TILE);
      __builtin_amdgcn_s_barrier();
      phaseV();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      sp = sk; sk = sn;
    }
    pv_tail(lds0 + sp * STAGE + TILE);
    __builtin_amdgcn_s_barrier();
  } else {
    int sp = 2;
    __builtin_amdgcn_s_barrier();
    for (int u = 0; u < nt; ++u) {
      const int sn = (sk == 2) ? 0 : sk + 1, sn2 = (sn == 2) ? 0 : sn + 1;
      if (u == 0) phaseM(std::false_type{}, lds0 + sk * STAGE, 0u);
      else phaseM(std::true_type{}, lds0 + sk * STAGE, lds0 + sp * STAGE + TILE);
      __builtin_amdgcn_s_setprio(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (u + 2 < nt) issue(u + 2, sn2);
      phaseV();
      __builtin_amdgcn_s_barrier();
      sp = sk; sk = sn;
    }
    pv_tail(lds0 + sp * STAGE + TILE);
  }

  // ---- epilogue: normalise, store O rows, log-sum-exp (log2 domain)
  float l_full = l_run;
  if constexpr (!ONES) l_full += __shfl_xor(l_full, 32, 64);
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    float l;
    if constexpr (ONES) l = __shfl(ot[DN - 1][f][LROW & 3], lq + 16 * (LROW >> 2), 64);
    else l = __shfl(l_full, 16 * f + lq, 64);
    const float m = __shfl(m_run, 16 * f + lq, 64);
    const float inv = 1.0f / l;
    const int row = q0 + f * 16 + lq;
    bf16_t* op = reinterpret_cast<bf16_t*>(p.O) + ((long)b * p.N + row) * p.ldo + (long)h * DH;
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      const int d0 = i * 16 + 4 * g;
      if (d0 < DH) {
        float v[4] = {ot[i][f][0] * inv, ot[i][f][1] * inv, ot[i][f][2] * inv, ot[i][f][3] * inv};
        store4(op + d0, v);
      }
    }
    if (p.LSE && g == 0) p.LSE[((long)b * p.H + h) * p.lse_stride + row] = m + __builtin_amdgcn_logf(l);
  }
}

// =============================================================================== dK / dV
// TAIL: N is not a multiple of 64 (query masking)
// FOLD (d_head 40, pre-scaled Q, row_ws given): -lse and -delta ride through the matrix products.  The 64-deep walk of
// d_head 40 has 24 spare contraction slots; columns 40..42 of every Q row hold -lse as three bf16 pieces (hi + mid + lo),
// columns 40..42 of K hold 1.0 -- so S arrives as s - lse (log2 domain: Q is pre-scaled) and P = exp2(S); likewise dO / V
// carry -delta / 1.0 and dP arrives as dP - delta.  Per score pair that leaves exp2, exp2, one packed multiply (5 -> 3 VALU).
// The pieces come from row_ws (written by the dQ kernel, which runs first) as the pad chunk of the Q / dO tiles.
template <int DH, int KF, bool TAIL, bool PRIO = false, bool FOLD = false>
__global__ __launch_bounds__(256, (DH <= 80 && KF <= 2 ? 2 : 1)) void attn_bwd_dkv_tr_kernel(AttnBwdArgs p) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, KSTEPS = G::KSTEPS, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE, TI = G::TI;
  static_assert(!FOLD || (DH == 40 && !TAIL), "the fold uses the pad chunk of the 96-byte pitch, whole tiles only");
  constexpr int CIN = FOLD ? G::CPRP : CPR;    // chunks of a row that enter the S / dP contraction
  constexpr bool PIPE = FOLD;                  // software-pipelined fragment reads (see the tile loop)
  constexpr int STAGE = 2 * TILE + 512;     // Q tile, dO tile, lse[64], delta[64]
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lq = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kv_w = blockIdx.x * (64 * KF) + wave * (16 * KF);
  const float sl2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;
  const float kscale = p.q_prescaled ? 0.6931471805599453f : p.scale;   // dK = scale dS^T q = ln 2 dS^T q' for q' = q scale log2 e

  // K and V fragments (B operands: col = key lq, k = d chunk) stay in registers
  u32x4_t kb[KF][KSTEPS], vb[KF][KSTEPS];
#pragma unroll
  for (int kf = 0; kf < KF; ++kf) {
    const int kr = min(kv_w + kf * 16 + lq, p.Nkv - 1);
    const char* kp = (const char*)p.K + (((long)b * p.Nkv + kr) * p.ldk + (long)h * DH) * 2;
    const char* vp = (const char*)p.V + (((long)b * p.Nkv + kr) * p.ldv + (long)h * DH) * 2;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = 4 * ks + g;
      kb[kf][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(kp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
      vb[kf][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(vp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
      if constexpr (FOLD) {
        if (c == CPR) { kb[kf][ks] = ones3_chunk(); vb[kf][ks] = ones3_chunk(); }    // columns 40..42 = 1.0
      }
    }
  }
  const char* qbase = (const char*)p.Q + ((long)b * p.N * p.ldq + (long)h * DH) * 2;
  const char* dobase = (const char*)p.dO + ((long)b * p.N * p.lddo + (long)h * DH) * 2;
  const float* lse = p.LSE + ((long)b * p.H + h) * p.lse_stride;
  const float* dlt = p.Delta + ((long)b * p.H + h) * p.lse_stride;

  f32x4_t dvt[KF][DN], dkt[KF][DN];
#pragma unroll
  for (int kf = 0; kf < KF; ++kf)
#pragma unroll
    for (int i = 0; i < DN; ++i) { dvt[kf][i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dkt[kf][i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  TileDma<DH> dma; dma.init(wave, lane);
  int qoff[TileDma<DH>::NJ], dooff[TileDma<DH>::NJ];
  dma.offsets(p.ldq * 2, qoff); dma.offsets(p.lddo * 2, dooff);
  const char* rowpad = FOLD ? (const char*)p.row_ws + ((long)b * p.H + h) * p.lse_stride * 32 : nullptr;
  auto issue = [&](int t, int buf) {
    char* base = smem + buf * STAGE;
    if constexpr (FOLD) {                    // pad chunks = (-lse pieces) / (-delta pieces) of the rows: no lse / delta block
      dma.issue_with_pad(qbase, p.ldq * 2, qoff, t * 64, base, wave, rowpad, 32);
      dma.issue_with_pad(dobase, p.lddo * 2, dooff, t * 64, base + TILE, wave, rowpad + 16, 32);
      return;
    }
    dma.issue(qbase, p.ldq * 2, qoff, t * 64, p.N, base, wave);
    dma.issue(dobase, p.lddo * 2, dooff, t * 64, p.N, base + TILE, wave);
    if (wave == (TI & 3) && lane < 32) {   // lse (lanes 0-15) and delta (16-31), 64 floats each; lse_stride % 64 == 0
      const float* src = lane < 16 ? lse + t * 64 + lane * 4 : dlt + t * 64 + (lane - 16) * 4;
      glds16(src, base + 2 * TILE);
    }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // Row-fragment addresses, one per 32-deep step.  A lane whose chunk 4 ks + g lies past the contraction (d_head 40: chunks 6, 7
  // of the second step) reads the LAST contracted chunk again instead of being masked off: its B-operand slots (K / V
  // fragments above) are exact zeros, so the duplicate contributes +-0 and the result is bit-identical -- and the reads need
  // no exec masking, no zero-filled registers and no per-lane select (that was 36 v_mov + 10 branches per tile).
  uint32_t rrow[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) rrow[ks] = lq * ROWB + min(4 * ks + g, CIN - 1) * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const int ntiles = (p.N + 63) / 64;
  if constexpr (!FOLD) {                    // (FOLD: the pad chunks arrive with the tile DMA)
    init_pads<DH>(smem, 2, 0u, 0u, tid, 256);
    init_pads<DH>(smem + STAGE, 2, 0u, 0u, tid, 256);
  }
  // PIPE: tiles are requested TWO ahead into a ring of three stages, and the wait at the top of a tile counts this wave's
  // own requests of the tile after it (loads retire in order), so a tile's DMA has two tile times to land instead of one --
  // at ~0.9 us per tile one was not enough to cover an L2 / HBM round trip under load.  A stage is refilled at the top of
  // tile t with tile t + 2: its last readers (tile t - 1) are behind the barrier every wave has just passed.
  const int dma_per_tile = 2 * ((TI - wave + 3) / 4);      // this wave's DMA instructions per tile (Q + dO pieces)
  issue(0, 0);
  if constexpr (PIPE) { if (ntiles > 1) issue(1, 1); }
  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
#if ATTN_BWD_ABL == 4
    if constexpr (!PIPE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#else
    if constexpr (PIPE) {
      if (t + 1 >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (dma_per_tile == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#endif
    const uint32_t aQ = lds0 + buf * STAGE, adO = aQ + TILE, aL = adO + TILE;
    const int q0 = t * 64;
    // PIPE (= FOLD): the row fragments of the NEXT query fragment and the column fragments of the NEXT d block are requested
    // before the products of the current one are issued, and each wait counts only the requests that must have landed
    // (lgkmcnt retires LDS reads in order).  The masked form below waits for lgkmcnt(0) seven times per tile with nothing
    // of its own in flight: at two waves per SIMD that is seven exposed LDS round trips per wave and tile.
    uint32_t rb[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) rb[ks] = aQ + rrow[ks];
    const uint32_t tb = aQ + troff;
    u32x4_t qa[2][KSTEPS], da[2][KSTEPS];
    if constexpr (PIPE) {
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) { qa[0][ks] = BWD_RD128<0>(rb[ks]); da[0][ks] = BWD_RD128<TILE>(rb[ks]); }
    }
    if constexpr (PIPE) {
      if (t + 2 < ntiles) issue(t + 2, buf == 0 ? 2 : buf - 1);
    } else {
      if (t + 1 < ntiles) issue(t + 1, buf ^ 1);
    }

    // ---- S = Q K^T, dP = dO V^T  (rows = queries 16 qf + 4g + r, col = key lq)
    // (every fragment offset below is an instruction immediate on three address registers: rb[ks], tb)
    f32x4_t ps[KF][4], ds[KF][4];
    u32x4_t oa[2][2], qt[2][2];
    static_for<0, 4>([&](auto QF_) {
      constexpr int qf = decltype(QF_)::value;
      constexpr int cur = PIPE ? (qf & 1) : 0;
      u32x4_t l4 = {0u, 0u, 0u, 0u}, d4 = {0u, 0u, 0u, 0u};
      if constexpr (PIPE) {
        if constexpr (qf + 1 < 4) {
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            qa[cur ^ 1][ks] = BWD_RD128<(qf + 1) * 16 * ROWB>(rb[ks]);
            da[cur ^ 1][ks] = BWD_RD128<TILE + (qf + 1) * 16 * ROWB>(rb[ks]);
          }
          lgkm_wait<2 * KSTEPS>();
        } else {           // last query fragment: the first d block's column fragments go out under its exp2 / pack work
          oa[0][0] = BWD_TRF<ROWB, 0, TILE>(tb); oa[0][1] = BWD_TRF<ROWB, 1, TILE>(tb);
          qt[0][0] = BWD_TRF<ROWB, 0, 0>(tb); qt[0][1] = BWD_TRF<ROWB, 1, 0>(tb);
          lgkm_wait<8>();
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          qa[0][ks] = BWD_RD128<qf * 16 * ROWB>(rb[ks]);
          da[0][ks] = BWD_RD128<TILE + qf * 16 * ROWB>(rb[ks]);
        }
        if constexpr (!FOLD) {
          l4 = lds_read_b128(aL + (qf * 16 + 4 * g) * 4);
          d4 = lds_read_b128(aL + 256 + (qf * 16 + 4 * g) * 4);
        }
        lds_wait();
      }
      const float lv[4] = {__uint_as_float(l4.x), __uint_as_float(l4.y), __uint_as_float(l4.z), __uint_as_float(l4.w)};
      const float dv[4] = {__uint_as_float(d4.x), __uint_as_float(d4.y), __uint_as_float(d4.z), __uint_as_float(d4.w)};
      const int qrow = q0 + qf * 16 + 4 * g;
#pragma unroll
      for (int kf = 0; kf < KF; ++kf) {
        f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) { BWD_MMA(qa[cur][ks], kb[kf][ks], sc); BWD_MMA(da[cur][ks], vb[kf][ks], dp); }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        // P = exp2(s * sl2 - lse), dS = P (dP - delta); the d_head^-0.5 factor of dS is applied once to dK in the
        // epilogue (linear).  Two rows per packed instruction.
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          f32x2_t pr, dsv;
          if constexpr (FOLD) {            // sc = s - lse, dp = dP - delta already
            pr = f32x2_t{BWD_EXP2(sc[2 * h2]), BWD_EXP2(sc[2 * h2 + 1])};
            dsv = pr * f32x2_t{dp[2 * h2], dp[2 * h2 + 1]};
          } else {
            const f32x2_t s2 = {sc[2 * h2], sc[2 * h2 + 1]}, l2 = {lv[2 * h2], lv[2 * h2 + 1]};
            const f32x2_t x = s2 * sl2 - l2;
            pr = f32x2_t{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            const f32x2_t dd = f32x2_t{dp[2 * h2], dp[2 * h2 + 1]} - f32x2_t{dv[2 * h2], dv[2 * h2 + 1]};
            dsv = pr * dd;
          }
          if constexpr (TAIL) {   // lse / delta pads may hold NaN
            if (qrow + 2 * h2 >= p.N) { pr.x = 0.f; dsv.x = 0.f; }
            if (qrow + 2 * h2 + 1 >= p.N) { pr.y = 0.f; dsv.y = 0.f; }
          }
          ps[kf][qf][2 * h2] = pr.x; ps[kf][qf][2 * h2 + 1] = pr.y;
          ds[kf][qf][2 * h2] = dsv.x; ds[kf][qf][2 * h2 + 1] = dsv.y;
        }
      }
    });
    // ---- dV^T += dO^T . P ;  dK^T += Q^T . dS   (A operands by transpose reads of the same tiles)
    u32x4_t pb[KF][2], sb[KF][2];
#pragma unroll
    for (int kf = 0; kf < KF; ++kf)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { pb[kf][s2] = BWD_PACK(&ps[kf][2 * s2]); sb[kf][s2] = BWD_PACK(&ds[kf][2 * s2]); }
    static_for<0, DN>([&](auto I_) {
      constexpr int i = decltype(I_)::value;
      constexpr int cur = PIPE ? (i & 1) : 0;
      if constexpr (PIPE) {
        if constexpr (i + 1 < DN) {
          // (lgkmcnt is a 4-bit counter: never more than 12 reads in flight)
          oa[cur ^ 1][0] = BWD_TRF<ROWB, 0, TILE + (i + 1) * 32>(tb); oa[cur ^ 1][1] = BWD_TRF<ROWB, 1, TILE + (i + 1) * 32>(tb);
          lgkm_wait<4>();
          qt[cur ^ 1][0] = BWD_TRF<ROWB, 0, (i + 1) * 32>(tb); qt[cur ^ 1][1] = BWD_TRF<ROWB, 1, (i + 1) * 32>(tb);
        } else {
          lgkm_wait<0>();
        }
      } else {
        oa[0][0] = BWD_TRF<ROWB, 0, TILE + i * 32>(tb); oa[0][1] = BWD_TRF<ROWB, 1, TILE + i * 32>(tb);
        qt[0][0] = BWD_TRF<ROWB, 0, i * 32>(tb); qt[0][1] = BWD_TRF<ROWB, 1, i * 32>(tb);
        lds_wait();
      }
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kf = 0; kf < KF; ++kf)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { BWD_MMA(oa[cur][s2], pb[kf][s2], dvt[kf][i]); BWD_MMA(qt[cur][s2], sb[kf][s2], dkt[kf][i]); }
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    });
    if constexpr (PIPE) buf = buf == 2 ? 0 : buf + 1;
    else buf ^= 1;
  }
  // ---- store dK / dV rows (4 consecutive d per lane)
#pragma unroll
  for (int kf = 0; kf < KF; ++kf) {
    const int kr = kv_w + kf * 16 + lq;
    if (kr < p.Nkv) {
      bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dK) + ((long)b * p.Nkv + kr) * p.lddk + (long)h * DH;
      bf16_t* dvp = reinterpret_cast<bf16_t*>(p.dV) + ((long)b * p.Nkv + kr) * p.lddv + (long)h * DH;
#pragma unroll
      for (int i = 0; i < DN; ++i) {
        const int d0 = i * 16 + 4 * g;
        if (d0 < DH) {
          float a[4] = {dkt[kf][i][0] * kscale, dkt[kf][i][1] * kscale, dkt[kf][i][2] * kscale, dkt[kf][i][3] * kscale};
          float c[4] = {dvt[kf][i][0], dvt[kf][i][1], dvt[kf][i][2], dvt[kf][i][3]};
          store4(dkp + d0, a);
          store4(dvp + d0, c);
        }
      }
    }
  }
}

// =============================================================================== dQ
// TAIL: Nkv is not a multiple of 64 (key masking)
// DELTA: delta[q] = sum_d dO[q,d] O[q,d] is formed here from the dO fragments the kernel holds anyway (+ one read of the
// O rows) and stored for the dK/dV kernel, which then has to be launched AFTER this one: saves the separate
// attn_delta launch (32 per training step, ~13 us each at the 64x64 level).
// FOLD: as in the dK/dV kernel; here Q / dO are this wave's register fragments, so -lse / -delta go straight into their
// columns 40..42 and K / V tiles get (1, 1, 1, 0, ...) as their pad chunk.  The kernel also leaves the pieces in row_ws for
// the dK/dV kernel (DELTA form: it runs first).
template <int DH, int QF, bool TAIL, bool DELTA = false, bool PRIO = false, bool FOLD = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_tr_kernel(AttnBwdArgs p) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, KSTEPS = G::KSTEPS, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE;
  static_assert(!FOLD || (DH == 40 && !TAIL && DELTA), "fold: d_head 40, whole key tiles, fused delta");
  constexpr int CIN = FOLD ? G::CPRP : CPR;
  constexpr bool PIPE = FOLD;
  constexpr int STAGE = 2 * TILE;           // K tile, V tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lq = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q_w = blockIdx.x * (64 * QF) + wave * (16 * QF);
  const float sl2 = p.q_prescaled ? 1.0f : p.scale * 1.4426950408889634f;

  u32x4_t qb[QF][KSTEPS], ob[QF][KSTEPS];
  float lse_q[QF], dlt_q[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int qr = min(q_w + f * 16 + lq, p.N - 1);
    const char* qp = (const char*)p.Q + (((long)b * p.N + qr) * p.ldq + (long)h * DH) * 2;
    const char* op = (const char*)p.dO + (((long)b * p.N + qr) * p.lddo + (long)h * DH) * 2;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = 4 * ks + g;
      qb[f][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(qp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
      ob[f][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(op + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
    lse_q[f] = p.LSE[((long)b * p.H + h) * p.lse_stride + qr];
    if constexpr (DELTA) {
      // lane (lq, g) holds channels [8 (4 ks + g), +8) of query lq: partial dot product, then the four g lanes
      const char* orow = (const char*)p.O + (((long)b * p.N + qr) * p.ldo + (long)h * DH) * 2;
      float acc = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int c = 4 * ks + g;
        if (c < CPR) {
          const u32x4_t ov = *reinterpret_cast<const u32x4_t*>(orow + c * 16);
          const u32x4_t dv = ob[f][ks];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc += __uint_as_float(ov[j] << 16) * __uint_as_float(dv[j] << 16);
            acc += __uint_as_float(ov[j] & 0xffff0000u) * __uint_as_float(dv[j] & 0xffff0000u);
          }
        }
      }
      acc += __shfl_xor(acc, 16, 64);
      acc += __shfl_xor(acc, 32, 64);
      dlt_q[f] = acc;
      if (g == 0 && q_w + f * 16 + lq < p.N) p.Delta[((long)b * p.H + h) * p.lse_stride + qr] = acc;
    } else {
      dlt_q[f] = p.Delta[((long)b * p.H + h) * p.lse_stride + qr];
    }
    if constexpr (FOLD) {
      uint32_t l0, l1, d0, d1;
      split3_bf16(-lse_q[f], l0, l1);
      split3_bf16(-dlt_q[f], d0, d1);
      if (g == 1) {                          // chunk 5 = columns 40..47 of this lane's query (4 ks + g with ks = 1)
        qb[f][1] = u32x4_t{l0, l1, 0u, 0u};
        ob[f][1] = u32x4_t{d0, d1, 0u, 0u};
      }
      if (g == 0 && q_w + f * 16 + lq < p.N) {
        uint4* rp = reinterpret_cast<uint4*>((char*)p.row_ws + (((long)b * p.H + h) * p.lse_stride + qr) * 32);
        rp[0] = make_uint4(l0, l1, 0u, 0u);
        rp[1] = make_uint4(d0, d1, 0u, 0u);
      }
    }
  }
  const char* kbase = (const char*)p.K + ((long)b * p.Nkv * p.ldk + (long)h * DH) * 2;
  const char* vbase = (const char*)p.V + ((long)b * p.Nkv * p.ldv + (long)h * DH) * 2;

  f32x4_t dqt[QF][DN];
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int i = 0; i < DN; ++i) dqt[f][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t rrow[KSTEPS];     // as in the dK / dV kernel: lanes past the contraction re-read its last chunk against zero B slots
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) rrow[ks] = lq * ROWB + min(4 * ks + g, CIN - 1) * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const int ntiles = (p.Nkv + 63) / 64;
  TileDma<DH> dma; dma.init(wave, lane);
  int koff[TileDma<DH>::NJ], voff[TileDma<DH>::NJ];
  dma.offsets(p.ldk * 2, koff); dma.offsets(p.ldv * 2, voff);
  if constexpr (FOLD) init_pads_ones3<DH>(smem, 6, tid, 256);      // K / V columns 40..42 = 1.0 (three stages, see PIPE below)
  else init_pads<DH>(smem, 4, 0u, 0u, tid, 256);
  // PIPE: K / V tiles requested two ahead into three stages, counted vmcnt -- as in the dK / dV kernel.  (The row_ws / delta
  // stores above may still be in flight at the first wait: they only make it wait longer, never shorter -- loads retire
  // in order among themselves, and `count <= my requests of the next tile` implies the current tile's have all landed.)
  constexpr int TI = G::TI;
  const int dma_per_tile = 2 * ((TI - wave + 3) / 4);
  auto issue_kv = [&](int t, int st_) {
    dma.issue(kbase, p.ldk * 2, koff, t * 64, p.Nkv, smem + st_ * STAGE, wave);
    dma.issue(vbase, p.ldv * 2, voff, t * 64, p.Nkv, smem + st_ * STAGE + TILE, wave);
  };
  issue_kv(0, 0);
  if constexpr (PIPE) { if (ntiles > 1) issue_kv(1, 1); }
  int buf = 0;
  for (int t = 0; t < ntiles; ++t) {
#if ATTN_BWD_ABL == 4
    if constexpr (!PIPE) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#else
    if constexpr (PIPE) {
      if (t + 1 >= ntiles) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (dma_per_tile == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#endif
    const uint32_t aK = lds0 + buf * STAGE, aV = aK + TILE;
    const int kv0 = t * 64;
    uint32_t rb[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) rb[ks] = aK + rrow[ks];
    const uint32_t tb = aK + troff;
    u32x4_t ka[2][KSTEPS], va[2][KSTEPS], kc[2][2];
    if constexpr (PIPE) {        // software-pipelined fragment reads, as in the dK / dV kernel
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) { ka[0][ks] = BWD_RD128<0>(rb[ks]); va[0][ks] = BWD_RD128<TILE>(rb[ks]); }
    }
    if constexpr (PIPE) {
      if (t + 2 < ntiles) issue_kv(t + 2, buf == 0 ? 2 : buf - 1);
    } else {
      if (t + 1 < ntiles) issue_kv(t + 1, buf ^ 1);
    }

    // ---- S^T = K Q^T, dP^T = V dO^T  (rows = keys 16 kf + 4g + r, col = query lq)
    f32x4_t dst[QF][4];
    static_for<0, 4>([&](auto KF_) {
      constexpr int kf = decltype(KF_)::value;
      constexpr int cur = PIPE ? (kf & 1) : 0;
      if constexpr (PIPE) {
        if constexpr (kf + 1 < 4) {
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) {
            ka[cur ^ 1][ks] = BWD_RD128<(kf + 1) * 16 * ROWB>(rb[ks]);
            va[cur ^ 1][ks] = BWD_RD128<TILE + (kf + 1) * 16 * ROWB>(rb[ks]);
          }
          lgkm_wait<2 * KSTEPS>();
        } else {
          kc[0][0] = BWD_TRF<ROWB, 0, 0>(tb); kc[0][1] = BWD_TRF<ROWB, 1, 0>(tb);
          lgkm_wait<4>();
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          ka[0][ks] = BWD_RD128<kf * 16 * ROWB>(rb[ks]);
          va[0][ks] = BWD_RD128<TILE + kf * 16 * ROWB>(rb[ks]);
        }
        lds_wait();
      }
#pragma unroll
      for (int f = 0; f < QF; ++f) {
        f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) { BWD_MMA(ka[cur][ks], qb[f][ks], sc); BWD_MMA(va[cur][ks], ob[f][ks], dp); }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        // dS^T = P (dP - delta), two keys per packed instruction; d_head^-0.5 goes onto dQ in the epilogue
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          f32x2_t dsv;
          if constexpr (FOLD) {            // sc = s - lse, dp = dP - delta already
            const f32x2_t pr = {BWD_EXP2(sc[2 * h2]), BWD_EXP2(sc[2 * h2 + 1])};
            dsv = pr * f32x2_t{dp[2 * h2], dp[2 * h2 + 1]};
          } else {
            const f32x2_t x = f32x2_t{sc[2 * h2], sc[2 * h2 + 1]} * sl2 - lse_q[f];
            const f32x2_t pr = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            dsv = pr * (f32x2_t{dp[2 * h2], dp[2 * h2 + 1]} - dlt_q[f]);
          }
          if constexpr (TAIL) {
            if (kv0 + kf * 16 + 4 * g + 2 * h2 >= p.Nkv) dsv.x = 0.f;
            if (kv0 + kf * 16 + 4 * g + 2 * h2 + 1 >= p.Nkv) dsv.y = 0.f;
          }
          dst[f][kf][2 * h2] = dsv.x; dst[f][kf][2 * h2 + 1] = dsv.y;
        }
      }
    });
    // ---- dQ^T += K^T . dS^T   (A = K^T by transpose reads of the K tile)
    u32x4_t sb[QF][2];
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) sb[f][s2] = BWD_PACK(&dst[f][2 * s2]);
    static_for<0, DN>([&](auto I_) {
      constexpr int i = decltype(I_)::value;
      constexpr int cur = PIPE ? (i & 1) : 0;
      if constexpr (PIPE) {
        if constexpr (i + 1 < DN) {
          kc[cur ^ 1][0] = BWD_TRF<ROWB, 0, (i + 1) * 32>(tb); kc[cur ^ 1][1] = BWD_TRF<ROWB, 1, (i + 1) * 32>(tb);
          lgkm_wait<4>();
        } else {
          lgkm_wait<0>();
        }
      } else {
        kc[0][0] = BWD_TRF<ROWB, 0, i * 32>(tb); kc[0][1] = BWD_TRF<ROWB, 1, i * 32>(tb);
        lds_wait();
      }
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) BWD_MMA(kc[cur][s2], sb[f][s2], dqt[f][i]);
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    });
    if constexpr (PIPE) buf = buf == 2 ? 0 : buf + 1;
    else buf ^= 1;
  }
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int qrow = q_w + f * 16 + lq;
    if (qrow < p.N) {
      bf16_t* dqp = reinterpret_cast<bf16_t*>(p.dQ) + ((long)b * p.N + qrow) * p.lddq + (long)h * DH;
#pragma unroll
      for (int i = 0; i < DN; ++i) {
        const int d0 = i * 16 + 4 * g;
        if (d0 < DH) {
          float a[4] = {dqt[f][i][0] * p.scale, dqt[f][i][1] * p.scale, dqt[f][i][2] * p.scale, dqt[f][i][3] * p.scale};
          store4(dqp + d0, a);
        }
      }
    }
  }
}

// =============================================================================== host side
template <typename K>
static int set_lds(K kern, int bytes) {
  if (bytes > 65536 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return CL_ELAUNCH;
  return CL_OK;
}

int g_attn_variant_dkv4 = 0;   // probe hook: variant 21 = dK/dV with four key fragments per wave (fold kernels)
int g_attn_variant = 0;    // probe hook (csrc/debug_hooks.h): 1 = tile-synchronous kernels only, 11 = backward with s_setprio,
                           // 13 / 14 = hybrid forward with fragment lookahead 3 / 2 (also: skip the pre-scaled-Q forward)

template <int DH, int LA>
static int launch_fwd_hyb_t(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  constexpr int LDS = 3 * 2 * Geo<DH>::TILE + 16 * Geo<DH>::ROWB + 64;
  static bool done = false;
  if (!done) {
    if (set_lds(&attn_fwd_hyb_kernel<DH, LA>, LDS)) return CL_ELAUNCH;
    done = true;
  }
  const int nqb = a.N / 256;
  const long grid = (long)nqb * a.H * a.B;
  const int remap = ((a.B * a.H) % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((attn_fwd_hyb_kernel<DH, LA>), dim3((unsigned)grid), dim3(512), LDS, st, a, V, ldv, nqb, remap);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

// hybrid ping-pong forward: d_head 40 / 80, N a multiple of 256 queries, whole 64-key tiles, at least a chip of workgroups.
// INTERLEAVED A/B on one box, 7 rounds, medians (profiles/r03_attention/interleaved_*.json): B 8 x H 8, N 4096, d_head 40:
// 248.8 us (round-2 16x16x32 kernel) -> 240.4 us; d_head 80, N 1024: 38.1 -> 36.6 us.
template <int DH>
static bool launch_fwd_pp(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st, int* rc) {
  if constexpr (DH == 40 || DH == 80) {
    const long grid = (long)(a.N / 256) * a.H * a.B;
    if (g_attn_variant == 1 || a.N % 256 || a.Nkv % 64 || a.Nkv < 128 || grid < 256) return false;
    *rc = g_attn_variant == 13 ? launch_fwd_hyb_t<DH, 3>(a, V, ldv, st) : launch_fwd_hyb_t<DH, 2>(a, V, ldv, st);
    return true;
  }
  return false;
}

template <int DH, bool TAIL>
static int launch_fwd_tr(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  constexpr int LDS = 2 * 2 * Geo<DH>::TILE + 64 + 16 * Geo<DH>::ROWB;
  constexpr int QW = DH <= 80 ? 2 : 1;
  static bool done = false;
  if (!done) {
    if (set_lds(&attn_fwd_tr_kernel<DH, QW, TAIL>, LDS) || set_lds(&attn_fwd_tr_kernel<DH, 1, TAIL>, LDS)) return CL_ELAUNCH;
    done = true;
  }
  const long blocks128 = (long)((a.N + 127) / 128) * a.H * a.B;
  if (QW == 2 && blocks128 >= 512) {
    dim3 grid((a.N + 127) / 128, a.H, a.B);
    hipLaunchKernelGGL((attn_fwd_tr_kernel<DH, QW, TAIL>), grid, dim3(256), LDS, st, a, V, ldv);
  } else {
    dim3 grid((a.N + 63) / 64, a.H, a.B);
    hipLaunchKernelGGL((attn_fwd_tr_kernel<DH, 1, TAIL>), grid, dim3(256), LDS, st, a, V, ldv);
  }
  CL_CHECK_LAUNCH();
  return CL_OK;
}

template <int DH>
static int launch_fwd_tr_t(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  int rc = CL_OK;
  if (launch_fwd_pp<DH>(a, V, ldv, st, &rc)) return rc;
  return (a.Nkv % 64) ? launch_fwd_tr<DH, true>(a, V, ldv, st) : launch_fwd_tr<DH, false>(a, V, ldv, st);
}

int attn_fwd_tr(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  if ((a.ldq * 2) % 16 || (a.ldk * 2) % 16 || (ldv * 2) % 16 || (a.ldo * 2) % 16 || a.Nkv < 1 || a.N < 1) return CL_EINVAL;
  // pre-scaled Q, d_head 40, whole blocks: the VALU-lean software-pipelined forward (attention_fwd40.hip)
  if (g_attn_variant == 0 && attn_fwd40_applies(a)) return attn_fwd40(a, V, ldv, st);
  switch (a.DH) {
    case 8: return launch_fwd_tr_t<8>(a, V, ldv, st);
    case 16: return launch_fwd_tr_t<16>(a, V, ldv, st);
    case 32: return launch_fwd_tr_t<32>(a, V, ldv, st);
    case 40: return launch_fwd_tr_t<40>(a, V, ldv, st);
    case 80: return launch_fwd_tr_t<80>(a, V, ldv, st);
    case 160: return launch_fwd_tr_t<160>(a, V, ldv, st);
    default: return CL_EINVAL;
  }
}

int g_attn_fuse_delta = 1;   // A/B hook (cl_debug_attention_fuse_delta(0) clears it)

// tile-synchronous backward.  fused_delta: the dQ kernel forms delta itself and runs first (default); otherwise the separate
// attn_delta launch precedes both kernels.  (Ping-pong forms of these two kernels were built in round 2, measured correct
// and 5 % slower -- profiles/r02_attention_ab.json -- and removed in round 4.)
template <int DH, bool TQ, bool TK>
static int launch_bwd_tr(const AttnBwdArgs& a, hipStream_t st) {
  constexpr int KF = DH <= 40 ? 2 : 1;     // key / query fragments per wave (register budget: <= 256 VGPRs)
  constexpr int LDS_DKV = 2 * (2 * Geo<DH>::TILE + 512) + 64 + 16 * Geo<DH>::ROWB;
  constexpr int LDS_DQ = 2 * 2 * Geo<DH>::TILE + 64 + 16 * Geo<DH>::ROWB;
  constexpr int LDS_DKV3 = 3 * (2 * Geo<DH>::TILE + 512) + 64 + 16 * Geo<DH>::ROWB;   // fold kernels: three-stage ring
  constexpr int LDS_DQ3 = 3 * 2 * Geo<DH>::TILE + 64 + 16 * Geo<DH>::ROWB;
  static bool done = false;
  if (!done) {
    if (set_lds(&attn_bwd_dkv_tr_kernel<DH, KF, TQ>, LDS_DKV) || set_lds(&attn_bwd_dkv_tr_kernel<DH, 1, TQ>, LDS_DKV) ||
        set_lds(&attn_bwd_dq_tr_kernel<DH, KF, TK>, LDS_DQ) || set_lds(&attn_bwd_dq_tr_kernel<DH, 1, TK>, LDS_DQ) ||
        set_lds(&attn_bwd_dq_tr_kernel<DH, KF, TK, true>, LDS_DQ) || set_lds(&attn_bwd_dq_tr_kernel<DH, 1, TK, true>, LDS_DQ) ||
        set_lds(&attn_bwd_dq_tr_kernel<DH, KF, TK, true, true>, LDS_DQ) || set_lds(&attn_bwd_dkv_tr_kernel<DH, KF, TQ, true>, LDS_DKV))
      return CL_ELAUNCH;
    done = true;
  }
  const bool fused_delta = g_attn_fuse_delta != 0;
  if (!fused_delta) {
    const int rc = attn_delta(a, st);
    if (rc) return rc;
  }
  const bool prio = g_attn_variant == 11;
  // -lse / -delta folded into the matrix products: d_head 40, pre-scaled Q, whole tiles, row scratch given, enough
  // workgroups for the two-fragment forms (the 64x64 self-attentions)
  if constexpr (DH == 40 && !TQ && !TK) {
    const long qb2 = (long)(a.N / 128) * a.H * a.B, kb2 = (long)(a.Nkv / 128) * a.H * a.B;
    if (a.q_prescaled && a.row_ws && fused_delta && g_attn_variant == 0 && a.N % 128 == 0 && a.Nkv % 128 == 0 &&
        qb2 >= 512 && (!a.dK || kb2 >= 512)) {
      static bool done_f = false;
      if (!done_f) {
        if (set_lds(&attn_bwd_dq_tr_kernel<DH, KF, TK, true, false, true>, LDS_DQ3) ||
            set_lds(&attn_bwd_dkv_tr_kernel<DH, KF, TQ, false, true>, LDS_DKV3))
          return CL_ELAUNCH;
        done_f = true;
      }
      hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, KF, TK, true, false, true>), dim3(a.N / 128, a.H, a.B), dim3(256), LDS_DQ3, st, a);
      if (a.dK) {
        // probe (cl_debug_attention_variant(21)): FOUR key fragments per wave (64 keys, one wave per SIMD: the Q / dO fragments of a
        // tile are read once per 64 keys instead of once per 32) -- VERDICT r5 item 4
        if (g_attn_variant_dkv4 && a.Nkv % 256 == 0) {
          static bool done4 = false;
          if (!done4) { if (set_lds(&attn_bwd_dkv_tr_kernel<DH, 4, TQ, false, true>, LDS_DKV3)) return CL_ELAUNCH; done4 = true; }
          hipLaunchKernelGGL((attn_bwd_dkv_tr_kernel<DH, 4, TQ, false, true>), dim3(a.Nkv / 256, a.H, a.B), dim3(256), LDS_DKV3, st, a);
        } else
        hipLaunchKernelGGL((attn_bwd_dkv_tr_kernel<DH, KF, TQ, false, true>), dim3(a.Nkv / 128, a.H, a.B), dim3(256), LDS_DKV3, st, a);
      }
      CL_CHECK_LAUNCH();
      return CL_OK;
    }
  }
  auto launch_dq = [&]() {
    const long qb2 = (long)((a.N + 64 * KF - 1) / (64 * KF)) * a.H * a.B;
    if (KF == 2 && qb2 >= 512) {
      dim3 grid((a.N + 127) / 128, a.H, a.B);
      if (fused_delta && prio) hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, KF, TK, true, true>), grid, dim3(256), LDS_DQ, st, a);
      else if (fused_delta) hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, KF, TK, true>), grid, dim3(256), LDS_DQ, st, a);
      else hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, KF, TK>), grid, dim3(256), LDS_DQ, st, a);
    } else {
      dim3 grid((a.N + 63) / 64, a.H, a.B);
      if (fused_delta) hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, 1, TK, true>), grid, dim3(256), LDS_DQ, st, a);
      else hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, 1, TK>), grid, dim3(256), LDS_DQ, st, a);
    }
  };
  if (fused_delta) launch_dq();            // dQ (+ delta) first: the dK/dV kernel reads delta
  if (a.dK) {
    // two key fragments per wave only when that still leaves enough workgroups to fill the chip
    const long blocks2 = (long)((a.Nkv + 64 * KF - 1) / (64 * KF)) * a.H * a.B;
    if (KF == 2 && blocks2 >= 512) {
      dim3 grid((a.Nkv + 127) / 128, a.H, a.B);
      if (prio) hipLaunchKernelGGL((attn_bwd_dkv_tr_kernel<DH, KF, TQ, true>), grid, dim3(256), LDS_DKV, st, a);
      else hipLaunchKernelGGL((attn_bwd_dkv_tr_kernel<DH, KF, TQ>), grid, dim3(256), LDS_DKV, st, a);
    } else {
      dim3 grid((a.Nkv + 63) / 64, a.H, a.B);
      hipLaunchKernelGGL((attn_bwd_dkv_tr_kernel<DH, 1, TQ>), grid, dim3(256), LDS_DKV, st, a);
    }
  }
  if (!fused_delta) launch_dq();
  CL_CHECK_LAUNCH();
  return CL_OK;
}

template <int DH>
static int launch_bwd_tr_t(const AttnBwdArgs& a, hipStream_t st) {
  const bool tq = a.N % 64, tk = a.Nkv % 64;
  if (tq) return tk ? launch_bwd_tr<DH, true, true>(a, st) : launch_bwd_tr<DH, true, false>(a, st);
  return tk ? launch_bwd_tr<DH, false, true>(a, st) : launch_bwd_tr<DH, false, false>(a, st);
}

int attn_bwd_tr(const AttnBwdArgs& a, hipStream_t st) {
  if ((a.ldq * 2) % 16 || (a.ldk * 2) % 16 || (a.ldv * 2) % 16 || (a.lddo * 2) % 16 || (a.ldo * 2) % 16) return CL_EINVAL;
  if ((a.lddq * 2) % 16 || a.lse_stride % 64 || a.lse_stride < a.N) return CL_EINVAL;
  if ((a.dK == nullptr) != (a.dV == nullptr)) return CL_EINVAL;
  if (a.row_ws && (reinterpret_cast<uintptr_t>(a.row_ws) & 15)) return CL_EINVAL;
  if (a.dK && ((a.lddk * 2) % 16 || (a.lddv * 2) % 16)) return CL_EINVAL;
  switch (a.DH) {
    case 8: return launch_bwd_tr_t<8>(a, st);
    case 16: return launch_bwd_tr_t<16>(a, st);
    case 32: return launch_bwd_tr_t<32>(a, st);
    case 40: return launch_bwd_tr_t<40>(a, st);
    case 80: return launch_bwd_tr_t<80>(a, st);
    case 160: return launch_bwd_tr_t<160>(a, st);
    default: return CL_EINVAL;
  }
}

}  // namespace cl
