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
  const float sl2 = p.scale * 1.4426950408889634f;

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
  const uint32_t krow = lq * ROWB + g * 16;                                    // b128: row lq, chunk g (+4 ks)
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
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      u32x4_t ka[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks)
        ka[ks] = (4 * ks + g < CPR) ? lds_read_b128(kt + krow + kf * 16 * ROWB + ks * 64) : u32x4_t{0u, 0u, 0u, 0u};
      lds_wait();
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        st[kf][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) Mma<bf16_t>::run(ka[ks], qf[f][ks], st[kf][f]);
      }
    }
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
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      u32x4_t va[2];
      va[0] = tr_frag<ROWB, 0>(vt + troff + i * 32);
      va[1] = tr_frag<ROWB, 1>(vt + troff + i * 32);
      lds_wait();
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[s], pb[s][f], ot[i][f]);
    }
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

// =============================================================================== forward, ping-pong schedule
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

// LA = fragment-read lookahead of the matrix phase, in groups of 4 MFMAs (LDS latency under 8 reading waves is
// several groups long); ABL = timing ablations for the probes (1: no exp2 in the vector phase, 2: no LDS reads in
// the matrix phase -- results are then wrong by construction)
// VAR (A/B variants, tests/tools/attn_bench.py): bit 0 = s_setprio(1) around the matrix phase's MFMA stream; bit 1 = static
// priority 1 for the second-dispatched wave group (guide T5, static form); bit 2 = single-issue v_fma_f32 instead of
// v_pk_fma_f32 in the softmax (MI355X_MICROARCH: packed fp32 VALU beside MFMAs costs more than two plain ones)
template <int DH, int LA = 4, int ABL = 0, int VAR = 0>
__global__ __launch_bounds__(512, 2) void attn_fwd_pp_kernel(AttnFwdArgs p, const void* __restrict__ V, long ldv,
                                                             int nqb, int remap) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, KSTEPS = G::KSTEPS, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE;
  constexpr int STAGE = 2 * TILE, QW = 2;
  constexpr bool ONES = (DH % 16) != 0;        // spare V^T rows exist: row DH carries the softmax denominator
  constexpr bool PADONES = ONES && G::CPRP > G::CPR;   // ... and the padded V tile already holds ones there
  constexpr int LROW = DH % 16;
  constexpr float RESCALE_THR = 6.0f;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 3 stages + 16 rows + 64 bytes of (zeroed) slack

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int g = lane >> 4, lq = lane & 15;
  int bh, qb;
  {
    const int id = blockIdx.x;
    if (remap) {   // all query blocks of one (batch, head) on one XCD: K/V are fetched into ONE L2
      const int xcd = id & 7, slot = id >> 3;
      bh = xcd + 8 * (slot / nqb); qb = slot - (slot / nqb) * nqb;
    } else { bh = id / nqb; qb = id - bh * nqb; }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qb * 256 + wave * 32;
  const float sl2 = p.scale * 1.4426950408889634f;

  // fragment over-reads (chunks >= CPR of the last rows) land in the slack: keep it finite (x 0 must stay 0)
  for (int i = tid; i < (16 * ROWB + 64) / 4; i += 512) reinterpret_cast<uint32_t*>(smem + 3 * STAGE)[i] = 0u;
  init_pads<DH>(smem, 6, 0u, 0x3F803F80u, tid, 512);   // K pad = 0 (meets Q zeros); V pad = 1.0: rows 40..47 of V^T

  u32x4_t qf[QW][KSTEPS];
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    const int row = q0 + f * 16 + lq;
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
  for (int f = 0; f < QW; ++f) { m_run[f] = -1e30f; l_run[f] = 0.f; }
  f32x4_t st[4][QW];
  u32x4_t pb[2][QW];

  // ---- DMA: wave w issues instructions w, w + 8, ... of a tile (1 KiB each, lane-linear image)
  constexpr int CPRP = G::CPRP, NJ = (CPRP + 7) / 8;
  int koff[NJ], voff[NJ];
  bool real[NJ];                                   // pad chunks of the LDS rows are not DMA targets
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
  const uint32_t krow = lq * ROWB + g * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const bool ones_lane = ONES && lq == LROW;
  const int nt = p.Nkv / 64;

  // ---- matrix phase: PV of the previous tile (PREV), then QK^T of this one; fragment reads one group ahead
  auto phaseM = [&](auto PREVc, uint32_t kt, uint32_t vt) {
    constexpr bool PREV = decltype(PREVc)::value;
    constexpr int NV = PREV ? DN : 0, NG = NV + 4;
    u32x4_t va[LA][2], ka[LA][KSTEPS];
    auto cnt_of = [](int j) constexpr { return j < NV ? 4 : KSTEPS; };   // LDS instructions of group j
    auto read_group = [&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      if constexpr (ABL == 2) {
        if constexpr (J < NV) { va[J % LA][0] = qf[0][0]; va[J % LA][1] = qf[1][0]; }
        else if constexpr (J < NG) {
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) ka[(J - NV) % LA][ks] = qf[0][ks];
        }
      } else if constexpr (J < NV) {
        va[J % LA][0] = tr_frag<ROWB, 0>(vt + troff + J * 32);
        va[J % LA][1] = tr_frag<ROWB, 1>(vt + troff + J * 32);
      } else if constexpr (J < NG) {
        constexpr int kf = J - NV;
        // chunks >= CPR read finite garbage that meets zeros of the Q fragment
        static_for<0, KSTEPS>([&](auto Kc) {
          constexpr int ks = decltype(Kc)::value;
          ka[kf % LA][ks] = lds_read_b128_off<kf * 16 * ROWB + ks * 64>(kt + krow);
        });
      }
    };
    static_for<0, LA - 1>([&](auto Jc) { read_group(Jc); });
    if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(1);
    static_for<0, NG>([&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      read_group(std::integral_constant<int, J + LA - 1>{});       // its ring slot was consumed by group J-1
      // groups J+1 .. J+LA-1 may stay outstanding (LDS returns in order)
      constexpr int pending = [&]() constexpr { int n = 0; for (int k = J + 1; k < J + LA && k < NG; ++k) n += cnt_of(k); return n; }();
      if constexpr (ABL != 2) lgkm_wait<(pending > 15 ? 15 : pending)>();
      if constexpr (J < NV) {
        pin(va[J % LA][0]); pin(va[J % LA][1]);
        u32x4_t a0 = va[J % LA][0], a1 = va[J % LA][1];
        if constexpr (ONES && !PADONES && J == DN - 1) {
          const uint32_t one2 = 0x3F803F80u;
          a0.x = ones_lane ? one2 : a0.x; a0.y = ones_lane ? one2 : a0.y; a0.z = ones_lane ? one2 : a0.z; a0.w = ones_lane ? one2 : a0.w;
          a1.x = ones_lane ? one2 : a1.x; a1.y = ones_lane ? one2 : a1.y; a1.z = ones_lane ? one2 : a1.z; a1.w = ones_lane ? one2 : a1.w;
        }
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(a0, pb[0][f], ot[J][f]);
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(a1, pb[1][f], ot[J][f]);
      } else {
        constexpr int kf = J - NV;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) pin(ka[kf % LA][ks]);
#pragma unroll
        for (int f = 0; f < QW; ++f) st[kf][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
          for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(ka[kf % LA][ks], qf[f][ks], st[kf][f]);
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(0);
  };

  // ---- vector phase: online softmax of st -> pb (registers only)
  auto phaseV = [&]() {
    float ml[QW];
    bool need = false;
#pragma unroll
    for (int f = 0; f < QW; ++f) {
      float mx = st[0][f][0];
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, st[kf][f][r]);
      ml[f] = mx;
      need |= (mx * sl2 > m_run[f] + RESCALE_THR);
    }
    if (__any(need)) {   // wave-uniform, rare after the first tiles: move the running maximum and rescale O (and l)
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        float mx = ml[f];
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run[f], mx * sl2);
        const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
        m_run[f] = m_new;
        l_run[f] *= alpha;
#pragma unroll
        for (int i = 0; i < DN; ++i) ot[i][f] *= alpha;
      }
    }
#pragma unroll
    for (int f = 0; f < QW; ++f) {
      float ls = 0.f;
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          f32x2_t x;
          if constexpr (VAR & 4) {
            const float nm = -m_run[f];
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x.x) : "v"(st[kf][f][2 * h2]), "v"(sl2), "v"(nm));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x.y) : "v"(st[kf][f][2 * h2 + 1]), "v"(sl2), "v"(nm));
          } else {
            x = f32x2_t{st[kf][f][2 * h2], st[kf][f][2 * h2 + 1]} * sl2 - m_run[f];   // v_pk_fma_f32
          }
          const float e0 = (ABL == 1) ? x.x : __builtin_amdgcn_exp2f(x.x);
          const float e1 = (ABL == 1) ? x.y : __builtin_amdgcn_exp2f(x.y);
          st[kf][f][2 * h2] = e0; st[kf][f][2 * h2 + 1] = e1;
          if constexpr (!ONES) ls += e0 + e1;
        }
      if constexpr (!ONES) l_run[f] += ls;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        f32x4_t tmp[2] = {st[2 * s][f], st[2 * s + 1][f]};
        pb[s][f] = PFrag<bf16_t>::make(tmp);
      }
  };

  issue(0, 0);
  issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // tiles 0, 1 landed; slack zeroed
  int sk = 0;                                        // ring stage of tile u;  tile u-1 sits in stage sp
  if constexpr (VAR & 2) { if (grp == 1) __builtin_amdgcn_s_setprio(1); }
  if (grp == 0) {
    int sp = 2;
    for (int u = 0; u < nt; ++u) {
      const int sn = (sk == 2) ? 0 : sk + 1;
      if (u >= 1 && u + 1 < nt) issue(u + 1, sn);                          // interval 2u
      if (u == 0) phaseM(std::false_type{}, lds0 + sk * STAGE, 0u);
      else phaseM(std::true_type{}, lds0 + sk * STAGE, lds0 + sp * STAGE + TILE);
      __builtin_amdgcn_s_barrier();
      phaseV();                                                            // interval 2u+1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      sp = sk; sk = sn;
    }
    // interval 2nt: PV of the last tile (reads only; nothing left to guard with a barrier but the count must match)
    {
      const uint32_t vt = lds0 + sp * STAGE + TILE;
      u32x4_t va[2];
#pragma unroll
      for (int i = 0; i < DN; ++i) {
        va[0] = tr_frag<ROWB, 0>(vt + troff + i * 32); va[1] = tr_frag<ROWB, 1>(vt + troff + i * 32);
        lds_wait();
        if (ONES && !PADONES && i == DN - 1 && ones_lane) { va[0] = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; va[1] = va[0]; }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[s], pb[s][f], ot[i][f]);
      }
    }
    __builtin_amdgcn_s_barrier();
  } else {
    int sp = 2;
    __builtin_amdgcn_s_barrier();                                          // interval 0: idle
    for (int u = 0; u < nt; ++u) {
      const int sn = (sk == 2) ? 0 : sk + 1, sn2 = (sn == 2) ? 0 : sn + 1;
      if (u == 0) phaseM(std::false_type{}, lds0 + sk * STAGE, 0u);       // interval 2u+1
      else phaseM(std::true_type{}, lds0 + sk * STAGE, lds0 + sp * STAGE + TILE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (u + 2 < nt) issue(u + 2, sn2);                                   // interval 2u+2
      phaseV();
      __builtin_amdgcn_s_barrier();
      sp = sk; sk = sn;
    }
    {                                                                      // interval 2nt+1
      const uint32_t vt = lds0 + sp * STAGE + TILE;
      u32x4_t va[2];
#pragma unroll
      for (int i = 0; i < DN; ++i) {
        va[0] = tr_frag<ROWB, 0>(vt + troff + i * 32); va[1] = tr_frag<ROWB, 1>(vt + troff + i * 32);
        lds_wait();
        if (ONES && !PADONES && i == DN - 1 && ones_lane) { va[0] = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u}; va[1] = va[0]; }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[s], pb[s][f], ot[i][f]);
      }
    }
  }

  // ---- epilogue: normalise, store O rows, log-sum-exp (log2 domain)
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    float l;
    if constexpr (ONES) {
      l = __shfl(ot[DN - 1][f][LROW & 3], lq + 16 * (LROW >> 2), 64);     // row DH of O^T = sum_k P
    } else {
      l = l_run[f];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
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
    if (p.LSE && g == 0) p.LSE[((long)b * p.H + h) * p.lse_stride + row] = m_run[f] + __builtin_amdgcn_logf(l);
  }
}

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
typedef float f32x16_t __attribute__((ext_vector_type(16)));

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
  const float sl2 = p.scale * 1.4426950408889634f;

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

// =============================================================================== forward, per-wave software pipeline
// Round-3 probe (tools/probe_interleave.hip): with the LDS operand reads in the stream, the barrier-phased
// ping-pong above is the slowest way to arrange one step's 18 MFMAs and ~100 softmax VALU (899 cycles per wave and step in the
// probe, 975 in the kernel); a wave that interleaves its OWN vector work between its OWN MFMAs takes 712.  The real streams are
// dependent (softmax(u) needs S(u), P.V(u) needs softmax(u)), so the wave keeps two tiles in flight: iteration u issues
//      MFMAs    P.V of tile u-1  (P from iteration u-1, V^T of tile u-1)   and   S^T = K Q^T of tile u+1
//      VALU     online softmax of tile u  (S from iteration u-1)
// -- three mutually independent pieces.  The max pass rides on the first four P.V MFMAs, the lazy-rescale decision follows,
// the exp2 / cvt / swap pass rides on the remaining P.V MFMAs and the six S MFMAs; a rescale of O (rare) is applied after the
// iteration's last P.V MFMA.  One s_barrier per tile (tile u+1 visible, slot of tile u-2 free), no wave-group stagger; K / V
// tiles live in a 4-slot ring (tile t is read in iterations t-1 .. t+1, its DMA is issued in iteration t-2).  All operand
// fragments of an iteration are requested up front (V^T at the top, K after the first MFMAs: never more than 15 LDS reads in
// flight) so that their latency is covered by the wave's own work.  Same arithmetic, layouts and epilogue as
// attn_fwd_hyb_kernel; S is double-buffered in registers, hence 8 waves per CU (256 VGPRs) instead of 16.
template <int DH, int NWAVES = 8, int AHEAD = 2, int RING = 4, bool SINGLE = false, bool FOLD = false>
__global__ __launch_bounds__(64 * NWAVES, (SINGLE ? 4 : NWAVES == 8 ? 1 : 3)) void attn_fwd_il_kernel(AttnFwdArgs p, const void* __restrict__ V, long ldv, int nqb, int remap) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE;
  constexpr int STAGE = 2 * TILE, QW = 2;
  constexpr int NK = (DH + 15) / 16;
  constexpr bool ONES = (DH % 16) != 0;
  constexpr bool PADONES = ONES && G::CPRP > G::CPR;
  static_assert(!ONES || PADONES, "spare V^T rows come from the padded LDS pitch");
  static_assert(2 * NK <= G::CPRP, "K fragment reads stay inside the LDS row");
  constexpr int LROW = DH % 16;
  constexpr float RESCALE_THR = 6.0f;
  // K / V ring: iteration u requests tile u + AHEAD; a tile is last read (its V) in iteration t + 1, so tiles u - 1 .. u + AHEAD
  // are live: AHEAD + 2 of RING slots (power of two).  NWAVES = 8: one workgroup of 256 queries per CU (2 waves per SIMD);
  // NWAVES = 4: 128 queries per workgroup, three workgroups per CU (3 waves per SIMD, three independent barrier domains).
  // SINGLE: S is NOT double-buffered (32 registers less: 4 waves per SIMD again, two 8-wave workgroups per CU).  The S MFMAs of
  // tile u + 1 then overwrite S(u) in place, half by half, each half as soon as the exp2 pass has consumed it: the first half's
  // three MFMAs run beside the exp2 pass of the second half, the second half's beside the last swap chunk only.
  static_assert(AHEAD >= 2 && AHEAD + 2 <= RING && (RING & (RING - 1)) == 0, "ring depth");
  static_assert(!SINGLE || DH == 40, "the in-place schedule is laid out for NK = 3");
  // FOLD: the caller hands over Q ALREADY multiplied by scale * log2(e) (in production that factor belongs in the to_q weights:
  // rounding q again costs accuracy), and -m_run travels in the first spare k slot of the 48-deep walk (Q column 40 = -m_run
  // as bf16, K pad column 40 = 1.0): the matrix product delivers s - m_run and the softmax starts at v_exp_f32 -- 16 packed
  // fma per step less (probe: 712 -> 630 cycles).  m_run is kept bf16-representable; when it moves (rare branch) the scores in
  // hand are corrected by the difference and the Q slot is rewritten for the S tiles still to come.
  static_assert(!FOLD || (DH == 40 && PADONES), "the fold uses the spare k slots of d_head 40");
  constexpr int NTHR = 64 * NWAVES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lq = lane & 15, l31 = lane & 31, hi = lane >> 5;
  int bh, qb;
  {
    const int id = blockIdx.x;
    if (remap) { const int xcd = id & 7, slot = id >> 3; bh = xcd + 8 * (slot / nqb); qb = slot - (slot / nqb) * nqb; }
    else { bh = id / nqb; qb = id - bh * nqb; }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qb * (32 * NWAVES) + wave * 32;
  const float sl2 = p.scale * 1.4426950408889634f;

  for (int i = tid; i < (16 * ROWB + 64) / 4; i += NTHR) reinterpret_cast<uint32_t*>(smem + RING * STAGE)[i] = 0u;
  if constexpr (FOLD) {   // K pad chunk = (1.0, 0, 0, ...): column DH meets -m_run in Q; V pad = 1.0 as below
    for (int i = tid; i < 2 * RING * 64; i += NTHR) {
      const bool vtile = (i >> 6) & 1;
      const uint4 w = vtile ? make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u) : make_uint4(0x00003F80u, 0u, 0u, 0u);
      *reinterpret_cast<uint4*>(smem + (long)i * ROWB + CPR * 16) = w;
    }
  } else
    init_pads<DH>(smem, 2 * RING, 0u, 0x3F803F80u, tid, NTHR);   // K pad = 0 (meets Q zeros); V pad = 1.0: rows DH.. of V^T

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
  float m_run = FOLD ? 0.f : -1e30f, l_run = 0.f;      // FOLD: S(0) is formed against m = 0 and corrected in the first iteration
  f32x16_t sa[2], sb[2];                       // S^T of the tile being soft-maxed / of the next one (roles alternate)
  u32x4_t pb[2][QW];                           // P^T operands of the tile whose P.V is pending

  constexpr int CPRP = G::CPRP, NJ = (CPRP + NWAVES - 1) / NWAVES;
  int koff[NJ], voff[NJ];
  bool real[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + NWAVES * j) * 64 + lane, r = c / CPRP, col = c - r * CPRP;
    real[j] = col < CPR;
    const int cc = (real[j] ? col : 0) * 16;
    koff[j] = r * (int)(p.ldk * 2) + cc;
    voff[j] = r * (int)(ldv * 2) + cc;
  }
  auto issue = [&](int t, int slot) {
    const char* kb = kbase + (long)t * 64 * p.ldk * 2;
    const char* vb = vbase + (long)t * 64 * ldv * 2;
    char* dst = smem + slot * STAGE;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (wave + NWAVES * j < CPRP && real[j]) {
        glds16(kb + koff[j], dst + (wave + NWAVES * j) * 1024);
        glds16(vb + voff[j], dst + TILE + (wave + NWAVES * j) * 1024);
      }
  };

  // wait until at most `tiles` of this wave's most recent tile requests are still in flight.  A wave issues 2 DMA instructions
  // (K, V) per tile for every j with wave + NWAVES j < CPRP: the count differs between waves, the branch is wave-uniform.
  int per = 0;
#pragma unroll
  for (int j = 0; j < NJ; ++j) per += (wave + NWAVES * j < CPRP) ? 2 : 0;
  auto dma_wait = [&](int tiles) {
    const int n = (tiles > AHEAD - 1 ? AHEAD - 1 : tiles) * per;
    switch (n) {
      case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
      case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
      case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
      case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;      // (over-waiting is always safe)
    }
  };
  static_assert(NJ <= 2, "dma_wait cases");
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int kr = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);             // see attn_fwd_hyb_kernel
  const uint32_t krow = kr * ROWB + hi * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const int nt = p.Nkv / 64;

  auto qk_tile = [&](auto Sc, const u32x4_t (&ka)[2][NK], f32x16_t& dst) {   // S^T of one 32-key half: NK MFMAs
    constexpr int s_ = decltype(Sc)::value;
    f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NK; ++j)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka[s_][j]), __builtin_bit_cast(bf16x8_t, qh[j]), acc, 0, 0, 0);
    dst = acc;
  };

  // ---- one iteration.  cur = S(u) (complete), nxt receives S(u+1); HAS_PV: u >= 1; HAS_QK: u + 1 < nt
  auto body = [&](auto PVc, auto QKc, f32x16_t (&cur)[2], f32x16_t (&nxt)[2], int u) {
    constexpr bool HAS_PV = decltype(PVc)::value, HAS_QK = decltype(QKc)::value;
    {   // tile u + 1 must have landed; tiles u + 2 .. u + AHEAD - 1 (those that exist) may still be in flight
      const int later = min(u + AHEAD - 1, nt - 1) - (u + 1);
      dma_wait(later < 0 ? 0 : later);
    }
    __builtin_amdgcn_s_barrier();
    if (u + AHEAD < nt) issue(u + AHEAD, (u + AHEAD) & (RING - 1));
    const uint32_t kt = lds0 + ((u + 1) & (RING - 1)) * STAGE, vt = lds0 + ((u + RING - 1) & (RING - 1)) * STAGE + TILE;   // K(u+1), V(u-1)
    // Fragment requests.  Default: everything up front (V^T here, K after part A).  SINGLE (128-register budget): just in time
    // -- V^T groups 0, 1 here, group 2 after part A into group 0's registers, K half 0 before the last P.V group, K half 1
    // before the S MFMAs of half 0 -- at most two groups / halves are live at any time.
    constexpr int NVG = SINGLE ? 2 : DN;
    static_assert(!SINGLE || DN == 3, "just-in-time fragment schedule");
    u32x4_t va[NVG][2], ka[2][NK];
    auto req_v = [&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      va[J % NVG][0] = tr_frag<ROWB, 0>(vt + troff + J * 32);
      va[J % NVG][1] = tr_frag<ROWB, 1>(vt + troff + J * 32);
    };
    auto req_k = [&](auto Sc) {
      constexpr int s_ = decltype(Sc)::value;
      static_for<0, NK>([&](auto Kc) {
        constexpr int j = decltype(Kc)::value;
        ka[s_][j] = lds_read_b128_off<s_ * 32 * ROWB + j * 32>(kt + krow);
      });
    };
    if constexpr (HAS_PV) static_for<0, NVG>([&](auto Jc) { req_v(Jc); });
    else if constexpr (SINGLE && HAS_QK) req_k(std::integral_constant<int, 0>{});
    constexpr int NVR = HAS_PV ? 4 * NVG : 0;            // LDS instructions in flight for V^T (2 per fragment, 2 fragments per group)
    constexpr int NKR = HAS_QK ? 2 * NK : 0;
    static_assert(4 * (DN - 1) + 2 * NK <= 15, "LDS reads in flight");

    // ---- part A: running maximum of the 32 scores of this lane's query, beside the first P.V group
    float mx = cur[0][0];
    auto max_slice = [&](auto Ic) {
      constexpr int i = decltype(Ic)::value;                  // 4 slices of 8 scores
#pragma unroll
      for (int r = 0; r < 8; ++r) mx = fmaxf(mx, cur[i >> 1][(i & 1) * 8 + r]);
    };
    if constexpr (HAS_PV) {
      lgkm_wait<(NVR - 4 > 15 ? 15 : NVR - 4)>();
      pin(va[0][0]); pin(va[0][1]);
      static_for<0, 4>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        Mma<bf16_t>::run(va[0][i >> 1], pb[i >> 1][i & 1], ot[0][i & 1]);
        max_slice(Ic);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      static_for<0, 4>([&](auto Ic) { max_slice(Ic); });
    }
    if constexpr (SINGLE) {
      if constexpr (HAS_PV) req_v(std::integral_constant<int, 2>{});         // into group 0's registers (its MFMAs are issued)
    } else if constexpr (HAS_QK) {
      req_k(std::integral_constant<int, 0>{});
      req_k(std::integral_constant<int, 1>{});
    }
    // ---- lazy rescale decision (wave-uniform branch, rare after the first tiles); O is rescaled at the END of the iteration
    float alpha = 1.0f;
    bool resc;
    if constexpr (FOLD) {
      constexpr bool FIRST = !HAS_PV;                       // S(0) was formed against m = 0
      resc = FIRST ? true : __any(mx > RESCALE_THR);       // cur holds s - m_run
      if (resc) {
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = FIRST ? m_run + mx : fmaxf(m_run, m_run + mx);
        const uint32_t mb = pack2bf(m_new, 0.f) & 0xffffu;              // bf16 (round to nearest even)
        const float m_b = __uint_as_float(mb << 16);
        const float d = m_b - m_run;
        alpha = FIRST ? 1.0f : __builtin_amdgcn_exp2f(-d);              // (nothing to rescale in the first iteration)
        m_run = m_b;
        l_run *= alpha;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
          for (int r = 0; r < 16; ++r) cur[s_][r] -= d;                // the tile in hand was formed against the old maximum
        if (hi) qh[NK - 1][0] = (qh[NK - 1][0] & 0xffff0000u) | (mb ^ 0x8000u);   // Q column DH = -m_run for the S tiles to come
      }
    } else {
      resc = __any(mx * sl2 > m_run + RESCALE_THR);
      if (resc) {
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * sl2);
        alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        l_run *= alpha;
      }
    }
    // ---- part B: exp2 / pack / swap in 18 chunks of 4 instructions, beside the remaining P.V groups and the S MFMAs
    float ls = 0.f;
    uint32_t pk[2][8];
    auto chunk = [&](auto Cc) {
      constexpr int c = decltype(Cc)::value;
      constexpr int s_ = c / 9, k = c % 9;
      if constexpr (k < 8) {
        f32x2_t x = f32x2_t{cur[s_][2 * k], cur[s_][2 * k + 1]};
        if constexpr (!FOLD) x = x * sl2 - m_run;
        const float e0 = __builtin_amdgcn_exp2f(x.x), e1 = __builtin_amdgcn_exp2f(x.y);
        if constexpr (!ONES) ls += e0 + e1;
        pk[s_][k] = pack2bf(e0, e1);
      } else {
        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[s_][0], pk[s_][2], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[s_][1], pk[s_][3], false, false);
        const auto s2 = __builtin_amdgcn_permlane16_swap(pk[s_][4], pk[s_][6], false, false);
        const auto s3 = __builtin_amdgcn_permlane16_swap(pk[s_][5], pk[s_][7], false, false);
        pb[s_][0] = u32x4_t{s0[0], s1[0], s2[0], s3[0]};
        pb[s_][1] = u32x4_t{s0[1], s1[1], s2[1], s3[1]};
      }
    };
    // MFMA slots of part B: (DN - 1) * 4 P.V MFMAs (weight 1 each), then 2 * NK S MFMAs (weight 2 each); the 18 chunks are
    // spread over them by weight.  The swap chunks (8 and 17) overwrite pb: chunk 8 must follow the last P.V MFMA.
    constexpr int NPV = HAS_PV ? (DN - 1) * 4 : 0, NQK = HAS_QK ? 2 * NK : 0, NM = NPV + NQK;
    constexpr int WTOT = NPV + 2 * NQK;
    auto cend = [](int i) constexpr {                  // chunks [cend(i - 1), cend(i)) follow MFMA slot i
      const int npv = HAS_PV ? (DN - 1) * 4 : 0, nqk = HAS_QK ? 2 * NK : 0, wtot = npv + 2 * nqk;
      int e;
      if (SINGLE && HAS_QK) {
        // chunks 0-7 read S half 0, 9-16 read half 1; slot npv starts overwriting half 0, slot npv + NK half 1
        if (i + 1 <= npv) e = i + 1 < 8 ? i + 1 : 8;
        else { const int kq = i - npv; e = kq < NK ? 8 + ((kq + 1) * 9) / NK : 18; }
        if (i + 1 <= npv && i + 1 == npv && e < 8) e = 8;
      } else {
        const int w = (i + 1 <= npv) ? (i + 1) : npv + 2 * (i + 1 - npv);
        e = (18 * w) / wtot;
        if (i + 1 <= npv && e > 8) e = 8;            // no swap chunk before the last P.V MFMA has been issued
      }
      if (i + 1 == npv + nqk) e = 18;
      return e;
    };
    constexpr int PRE = (SINGLE && HAS_QK && !HAS_PV) ? 8 : 0;     // first iteration, in place: half 0 is consumed before its MFMAs
    static_for<0, PRE>([&](auto Cc) { chunk(Cc); });
    static_assert(NM > 0 && WTOT > 0, "an iteration has matrix work");
    static_for<0, NM>([&](auto Ic) {
      constexpr int i = decltype(Ic)::value;
      if constexpr (i < NPV) {
        constexpr int J = 1 + i / 4, q = i % 4;
        if constexpr (q == 0) {
          if constexpr (SINGLE) {
            if constexpr (J == 1) lgkm_wait<4>();                               // group 2 still in flight
            else {
              if constexpr (HAS_QK) req_k(std::integral_constant<int, 0>{});
              lgkm_wait<(HAS_QK ? NK : 0)>();
            }
          } else {
            constexpr int left = NVR - 4 * (J + 1) + NKR;     // LDS reads issued after this group's
            lgkm_wait<(left > 15 ? 15 : left)>();
          }
          pin(va[J % NVG][0]); pin(va[J % NVG][1]);
        }
        Mma<bf16_t>::run(va[J % NVG][q >> 1], pb[q >> 1][q & 1], ot[J][q & 1]);
      } else {
        constexpr int m = i - NPV, s_ = m / NK, j = m % NK;
        if constexpr (SINGLE) {
          // half 1's fragment j is requested right after half 0's MFMA j (whose fragment is dead by then): 12 fragment registers
          if constexpr (s_ == 0 && j == 0) {
            lgkm_wait<0>();
#pragma unroll
            for (int jj = 0; jj < NK; ++jj) pin(ka[0][jj]);
          }
          if constexpr (s_ == 1) { lgkm_wait<NK - 1 - j>(); pin(ka[1][j]); }
        } else if constexpr (j == 0) {
          lgkm_wait<(s_ == 0 ? NK : 0)>();
#pragma unroll
          for (int jj = 0; jj < NK; ++jj) pin(ka[s_][jj]);
        }
        // (one accumulator chain per 32-key half; the first MFMA of a half starts from zero)
        if constexpr (j == 0) nxt[s_] = f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        nxt[s_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka[s_][j]), __builtin_bit_cast(bf16x8_t, qh[j]), nxt[s_], 0, 0, 0);
        if constexpr (SINGLE && s_ == 0) ka[1][j] = lds_read_b128_off<32 * ROWB + j * 32>(kt + krow);
      }
      constexpr int lo = i == 0 ? PRE : (cend(i - 1) > PRE ? cend(i - 1) : PRE), hi_ = cend(i) > lo ? cend(i) : lo;
      static_for<lo, hi_>([&](auto Cc) { chunk(Cc); });
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (!ONES) l_run += ls;
    if (resc) {                                          // after the iteration's last P.V MFMA: O(u-1) -> alpha O(u-1)
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        const float af = __shfl(alpha, 16 * f + lq, 64);
#pragma unroll
        for (int i = 0; i < DN; ++i) ot[i][f] *= af;
      }
    }
  };

#pragma unroll
  for (int t = 0; t < AHEAD; ++t)
    if (t < nt) issue(t, t);
  dma_wait(min(AHEAD, nt) - 1);            // tile 0
  __syncthreads();
  {   // S(0)
    u32x4_t ka[2][NK];
    static_for<0, 2>([&](auto Sc) {
      constexpr int s_ = decltype(Sc)::value;
      static_for<0, NK>([&](auto Kc) {
        constexpr int j = decltype(Kc)::value;
        ka[s_][j] = lds_read_b128_off<s_ * 32 * ROWB + j * 32>(lds0 + krow);
      });
    });
    lgkm_wait<0>();
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int j = 0; j < NK; ++j) pin(ka[s_][j]);
    qk_tile(std::integral_constant<int, 0>{}, ka, sa[0]);
    qk_tile(std::integral_constant<int, 1>{}, ka, sa[1]);
  }
  // nt >= 2 (launcher): first iteration without P.V, last without S, roles of sa / sb alternate (SINGLE: sb IS sa)
  f32x16_t (&sbr)[2] = SINGLE ? sa : sb;
  body(std::false_type{}, std::true_type{}, sa, sbr, 0);
  int u = 1;
  for (; u + 2 < nt; u += 2) {
    body(std::true_type{}, std::true_type{}, sbr, sa, u);
    body(std::true_type{}, std::true_type{}, sa, sbr, u + 1);
  }
  if (u + 1 < nt) {        // two iterations left: u (full) and u + 1 (last)
    body(std::true_type{}, std::true_type{}, sbr, sa, u);
    body(std::true_type{}, std::false_type{}, sa, sbr, u + 1);
  } else {                 // one left
    body(std::true_type{}, std::false_type{}, sbr, sa, u);
  }
  {   // P.V of the last tile
    const uint32_t vt = lds0 + ((nt - 1) & (RING - 1)) * STAGE + TILE;
    u32x4_t va[2];
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      va[0] = tr_frag<ROWB, 0>(vt + troff + i * 32); va[1] = tr_frag<ROWB, 1>(vt + troff + i * 32);
      lds_wait();
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[s_], pb[s_][f], ot[i][f]);
    }
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
template <int DH, int KF, bool TAIL, bool PRIO = false>
__global__ __launch_bounds__(256, (DH <= 80 ? 2 : 1)) void attn_bwd_dkv_tr_kernel(AttnBwdArgs p) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, KSTEPS = G::KSTEPS, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE, TI = G::TI;
  constexpr int STAGE = 2 * TILE + 512;     // Q tile, dO tile, lse[64], delta[64]
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lq = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kv_w = blockIdx.x * (64 * KF) + wave * (16 * KF);
  const float sl2 = p.scale * 1.4426950408889634f;

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
  auto issue = [&](int t, int buf) {
    char* base = smem + buf * STAGE;
    dma.issue(qbase, p.ldq * 2, qoff, t * 64, p.N, base, wave);
    dma.issue(dobase, p.lddo * 2, dooff, t * 64, p.N, base + TILE, wave);
    if (wave == (TI & 3) && lane < 32) {   // lse (lanes 0-15) and delta (16-31), 64 floats each; lse_stride % 64 == 0
      const float* src = lane < 16 ? lse + t * 64 + lane * 4 : dlt + t * 64 + (lane - 16) * 4;
      glds16(src, base + 2 * TILE);
    }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t rrow = lq * ROWB + g * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const int ntiles = (p.N + 63) / 64;
  init_pads<DH>(smem, 2, 0u, 0u, tid, 256);
  init_pads<DH>(smem + STAGE, 2, 0u, 0u, tid, 256);
  issue(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles) issue(t + 1, buf ^ 1);
    const uint32_t aQ = lds0 + buf * STAGE, adO = aQ + TILE, aL = adO + TILE;
    const int q0 = t * 64;

    // ---- S = Q K^T, dP = dO V^T  (rows = queries 16 qf + 4g + r, col = key lq)
    f32x4_t ps[KF][4], ds[KF][4];
    {
#pragma unroll
      for (int qf = 0; qf < 4; ++qf) {
        u32x4_t qa[KSTEPS], da[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const bool in = 4 * ks + g < CPR;
          qa[ks] = in ? lds_read_b128(aQ + rrow + qf * 16 * ROWB + ks * 64) : u32x4_t{0u, 0u, 0u, 0u};
          da[ks] = in ? lds_read_b128(adO + rrow + qf * 16 * ROWB + ks * 64) : u32x4_t{0u, 0u, 0u, 0u};
        }
        const u32x4_t l4 = lds_read_b128(aL + (qf * 16 + 4 * g) * 4);
        const u32x4_t d4 = lds_read_b128(aL + 256 + (qf * 16 + 4 * g) * 4);
        lds_wait();
        const float lv[4] = {__uint_as_float(l4.x), __uint_as_float(l4.y), __uint_as_float(l4.z), __uint_as_float(l4.w)};
        const float dv[4] = {__uint_as_float(d4.x), __uint_as_float(d4.y), __uint_as_float(d4.z), __uint_as_float(d4.w)};
        const int qrow = q0 + qf * 16 + 4 * g;
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) {
          f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) { Mma<bf16_t>::run(qa[ks], kb[kf][ks], sc); Mma<bf16_t>::run(da[ks], vb[kf][ks], dp); }
          if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
          // P = exp2(s * sl2 - lse), dS = P (dP - delta); the d_head^-0.5 factor of dS is applied once to dK in the
          // epilogue (linear).  Two rows per packed instruction.
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x2_t s2 = {sc[2 * h2], sc[2 * h2 + 1]}, l2 = {lv[2 * h2], lv[2 * h2 + 1]};
            const f32x2_t x = s2 * sl2 - l2;
            f32x2_t pr = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            const f32x2_t dd = f32x2_t{dp[2 * h2], dp[2 * h2 + 1]} - f32x2_t{dv[2 * h2], dv[2 * h2 + 1]};
            f32x2_t dsv = pr * dd;
            if constexpr (TAIL) {   // lse / delta pads may hold NaN
              if (qrow + 2 * h2 >= p.N) { pr.x = 0.f; dsv.x = 0.f; }
              if (qrow + 2 * h2 + 1 >= p.N) { pr.y = 0.f; dsv.y = 0.f; }
            }
            ps[kf][qf][2 * h2] = pr.x; ps[kf][qf][2 * h2 + 1] = pr.y;
            ds[kf][qf][2 * h2] = dsv.x; ds[kf][qf][2 * h2 + 1] = dsv.y;
          }
        }
      }
    }
    // ---- dV^T += dO^T . P ;  dK^T += Q^T . dS   (A operands by transpose reads of the same tiles)
    u32x4_t pb[KF][2], sb[KF][2];
#pragma unroll
    for (int kf = 0; kf < KF; ++kf)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) { pb[kf][s2] = PFrag<bf16_t>::make(&ps[kf][2 * s2]); sb[kf][s2] = PFrag<bf16_t>::make(&ds[kf][2 * s2]); }
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      u32x4_t oa[2], qt[2];
      oa[0] = tr_frag<ROWB, 0>(adO + troff + i * 32); oa[1] = tr_frag<ROWB, 1>(adO + troff + i * 32);
      qt[0] = tr_frag<ROWB, 0>(aQ + troff + i * 32); qt[1] = tr_frag<ROWB, 1>(aQ + troff + i * 32);
      lds_wait();
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kf = 0; kf < KF; ++kf)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { Mma<bf16_t>::run(oa[s2], pb[kf][s2], dvt[kf][i]); Mma<bf16_t>::run(qt[s2], sb[kf][s2], dkt[kf][i]); }
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    }
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
          float a[4] = {dkt[kf][i][0] * p.scale, dkt[kf][i][1] * p.scale, dkt[kf][i][2] * p.scale, dkt[kf][i][3] * p.scale};
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
template <int DH, int QF, bool TAIL, bool DELTA = false, bool PRIO = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_tr_kernel(AttnBwdArgs p) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, KSTEPS = G::KSTEPS, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE;
  constexpr int STAGE = 2 * TILE;           // K tile, V tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, lq = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q_w = blockIdx.x * (64 * QF) + wave * (16 * QF);
  const float sl2 = p.scale * 1.4426950408889634f;

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
  }
  const char* kbase = (const char*)p.K + ((long)b * p.Nkv * p.ldk + (long)h * DH) * 2;
  const char* vbase = (const char*)p.V + ((long)b * p.Nkv * p.ldv + (long)h * DH) * 2;

  f32x4_t dqt[QF][DN];
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int i = 0; i < DN; ++i) dqt[f][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t rrow = lq * ROWB + g * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const int ntiles = (p.Nkv + 63) / 64;
  TileDma<DH> dma; dma.init(wave, lane);
  int koff[TileDma<DH>::NJ], voff[TileDma<DH>::NJ];
  dma.offsets(p.ldk * 2, koff); dma.offsets(p.ldv * 2, voff);
  init_pads<DH>(smem, 4, 0u, 0u, tid, 256);
  dma.issue(kbase, p.ldk * 2, koff, 0, p.Nkv, smem, wave);
  dma.issue(vbase, p.ldv * 2, voff, 0, p.Nkv, smem + TILE, wave);
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles) {
      dma.issue(kbase, p.ldk * 2, koff, (t + 1) * 64, p.Nkv, smem + (buf ^ 1) * STAGE, wave);
      dma.issue(vbase, p.ldv * 2, voff, (t + 1) * 64, p.Nkv, smem + (buf ^ 1) * STAGE + TILE, wave);
    }
    const uint32_t aK = lds0 + buf * STAGE, aV = aK + TILE;
    const int kv0 = t * 64;

    // ---- S^T = K Q^T, dP^T = V dO^T  (rows = keys 16 kf + 4g + r, col = query lq)
    f32x4_t dst[QF][4];
    {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) {
        u32x4_t ka[KSTEPS], va[KSTEPS];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
          const bool in = 4 * ks + g < CPR;
          ka[ks] = in ? lds_read_b128(aK + rrow + kf * 16 * ROWB + ks * 64) : u32x4_t{0u, 0u, 0u, 0u};
          va[ks] = in ? lds_read_b128(aV + rrow + kf * 16 * ROWB + ks * 64) : u32x4_t{0u, 0u, 0u, 0u};
        }
        lds_wait();
#pragma unroll
        for (int f = 0; f < QF; ++f) {
          f32x4_t sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int ks = 0; ks < KSTEPS; ++ks) { Mma<bf16_t>::run(ka[ks], qb[f][ks], sc); Mma<bf16_t>::run(va[ks], ob[f][ks], dp); }
          if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
          // dS^T = P (dP - delta), two keys per packed instruction; d_head^-0.5 goes onto dQ in the epilogue
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x2_t x = f32x2_t{sc[2 * h2], sc[2 * h2 + 1]} * sl2 - lse_q[f];
            const f32x2_t pr = {__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
            f32x2_t dsv = pr * (f32x2_t{dp[2 * h2], dp[2 * h2 + 1]} - dlt_q[f]);
            if constexpr (TAIL) {
              if (kv0 + kf * 16 + 4 * g + 2 * h2 >= p.Nkv) dsv.x = 0.f;
              if (kv0 + kf * 16 + 4 * g + 2 * h2 + 1 >= p.Nkv) dsv.y = 0.f;
            }
            dst[f][kf][2 * h2] = dsv.x; dst[f][kf][2 * h2 + 1] = dsv.y;
          }
        }
      }
    }
    // ---- dQ^T += K^T . dS^T   (A = K^T by transpose reads of the K tile)
    u32x4_t sb[QF][2];
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) sb[f][s2] = PFrag<bf16_t>::make(&dst[f][2 * s2]);
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      u32x4_t kc[2];
      kc[0] = tr_frag<ROWB, 0>(aK + troff + i * 32); kc[1] = tr_frag<ROWB, 1>(aK + troff + i * 32);
      lds_wait();
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) Mma<bf16_t>::run(kc[s2], sb[f][s2], dqt[f][i]);
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    }
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

// =============================================================================== backward, ping-pong schedule
// Same two-groups-one-phase-apart structure as attn_fwd_pp_kernel, for the two backward kernels.  The unit of a
// phase is a HALF tile (32 of the 64 staged rows), which keeps the fp32 score registers at 32 per wave:
//   dK/dV: wave owns 16 KF keys (K, V fragments in registers), tiles {Q, dO, lse, delta} of 64 queries
//     M(h) = [dV^T += dO^T P, dK^T += Q^T dS of half h-1 (transpose reads)] + [S = Q K^T, dP = dO V^T of half h]
//     V(h) = P = exp2(S sl2 - lse), dS = P (dP - delta)  (packed fp32), bf16 fragments for the next M
//   dQ   : wave owns 16 QF queries (Q, dO fragments, lse, delta in registers), tiles {K, V} of 64 keys
//     M(h) = [dQ^T += K^T dS^T of half h-1] + [S^T = K Q^T, dP^T = V dO^T of half h];  V(h) = dS^T
// The d_head^-0.5 factor of dS is applied once, to dK / dQ, in the epilogue (linear).  3-stage tile ring: tile t is
// read in intervals 4t .. 4t+5, tile t+3 is issued at the start of interval 4t+6 into its stage and drained (vmcnt 0)
// before the barrier that ends interval 4t+7.  lse / delta blocks live behind the zeroed slack, so fragment
// over-reads (chunks >= CPR meet zero operand entries) never see non-finite bit patterns.
template <int DH, int KF, int LA = 3>
__global__ __launch_bounds__(512, 2) void attn_bwd_dkv_pp_kernel(AttnBwdArgs p, int nkb, int remap) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, KSTEPS = G::KSTEPS, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE;
  constexpr int STAGE = 2 * TILE, SLACK = 16 * ROWB + 64, LOFF = 3 * STAGE + SLACK;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int g = lane >> 4, lq = lane & 15;
  int bh, kbi;
  {
    const int id = blockIdx.x;
    if (remap) { const int xcd = id & 7, slot = id >> 3; bh = xcd + 8 * (slot / nkb); kbi = slot - (slot / nkb) * nkb; }
    else { bh = id / nkb; kbi = id - bh * nkb; }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int kv_w = kbi * (128 * KF) + wave * (16 * KF);
  const float sl2 = p.scale * 1.4426950408889634f;

  for (int i = tid; i < SLACK / 4; i += 512) reinterpret_cast<uint32_t*>(smem + 3 * STAGE)[i] = 0u;
  init_pads<DH>(smem, 6, 0u, 0u, tid, 512);

  u32x4_t kb[KF][KSTEPS], vb[KF][KSTEPS];
#pragma unroll
  for (int kf = 0; kf < KF; ++kf) {
    const int kr = kv_w + kf * 16 + lq;
    const char* kp = (const char*)p.K + (((long)b * p.Nkv + kr) * p.ldk + (long)h * DH) * 2;
    const char* vp = (const char*)p.V + (((long)b * p.Nkv + kr) * p.ldv + (long)h * DH) * 2;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = 4 * ks + g;
      kb[kf][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(kp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
      vb[kf][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(vp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
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
  f32x4_t sc[KF][2], dp[KF][2];
  u32x4_t pb[KF], sb[KF];

  constexpr int CPRP = G::CPRP, NJ = (CPRP + 7) / 8;
  int qoff[NJ], dooff[NJ];
  bool real[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + 8 * j) * 64 + lane, r = c / CPRP, col = c - r * CPRP;
    real[j] = col < CPR;
    const int cc = (real[j] ? col : 0) * 16;
    qoff[j] = r * (int)(p.ldq * 2) + cc;
    dooff[j] = r * (int)(p.lddo * 2) + cc;
  }
  auto issue = [&](int t, int stage) {
    const char* qb = qbase + (long)t * 64 * p.ldq * 2;
    const char* ob = dobase + (long)t * 64 * p.lddo * 2;
    char* dst = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (wave + 8 * j < CPRP && real[j]) {
        glds16(qb + qoff[j], dst + (wave + 8 * j) * 1024);
        glds16(ob + dooff[j], dst + TILE + (wave + 8 * j) * 1024);
      }
    if (wave == (CPRP & 7) && lane < 32) {   // lse (lanes 0-15) and delta (16-31), 64 floats each
      const float* src = lane < 16 ? lse + t * 64 + lane * 4 : dlt + t * 64 + (lane - 16) * 4;
      glds16(src, smem + LOFF + stage * 512);
    }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t rrow = lq * ROWB + g * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const int ntiles = p.N / 64, NH = 2 * ntiles;

  // ---- matrix phase.  cur = stage base of the tile of half hh, par = which 32 rows; prv / parp: the same for half hh-1
  auto phaseM = [&](auto PREVc, auto CURc, uint32_t cur, int par, uint32_t prv, int parp) {
    constexpr bool PREV = decltype(PREVc)::value, CUR = decltype(CURc)::value;
    constexpr int NV = PREV ? DN : 0, NA = CUR ? 2 : 0, NG = NV + NA;
    u32x4_t bo[LA][2], aq[LA][KSTEPS], ad[LA][KSTEPS];
    const uint32_t tq = prv + troff + parp * 32 * ROWB, to = tq + TILE;
    const uint32_t rq = cur + rrow + par * 32 * ROWB, ro = rq + TILE;
    auto cnt_of = [](int j) constexpr { return j < NV ? 4 : 2 * KSTEPS; };
    auto read_group = [&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      if constexpr (J < NV) {
        bo[J % LA][0] = tr_frag<ROWB, 0>(to + J * 32);
        bo[J % LA][1] = tr_frag<ROWB, 0>(tq + J * 32);
      } else if constexpr (J < NG) {
        constexpr int qf2 = J - NV;
        static_for<0, KSTEPS>([&](auto Kc) {
          constexpr int ks = decltype(Kc)::value;
          aq[qf2 % LA][ks] = lds_read_b128_off<qf2 * 16 * ROWB + ks * 64>(rq);
          ad[qf2 % LA][ks] = lds_read_b128_off<qf2 * 16 * ROWB + ks * 64>(ro);
        });
      }
    };
    static_for<0, LA - 1>([&](auto Jc) { read_group(Jc); });
    static_for<0, NG>([&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      read_group(std::integral_constant<int, J + LA - 1>{});
      constexpr int pending = [&]() constexpr { int n = 0; for (int k = J + 1; k < J + LA && k < NG; ++k) n += cnt_of(k); return n; }();
      lgkm_wait<(pending > 15 ? 15 : pending)>();
      if constexpr (J < NV) {
        pin(bo[J % LA][0]); pin(bo[J % LA][1]);
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) {
          Mma<bf16_t>::run(bo[J % LA][0], pb[kf], dvt[kf][J]);
          Mma<bf16_t>::run(bo[J % LA][1], sb[kf], dkt[kf][J]);
        }
      } else {
        constexpr int qf2 = J - NV;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) { pin(aq[qf2 % LA][ks]); pin(ad[qf2 % LA][ks]); }
#pragma unroll
        for (int kf = 0; kf < KF; ++kf) { sc[kf][qf2] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[kf][qf2] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
          for (int kf = 0; kf < KF; ++kf) {
            Mma<bf16_t>::run(aq[qf2 % LA][ks], kb[kf][ks], sc[kf][qf2]);
            Mma<bf16_t>::run(ad[qf2 % LA][ks], vb[kf][ks], dp[kf][qf2]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- vector phase: P and dS of the half (rows = queries 16 (2 par + qf2) + 4g + r, col = key lq) -> bf16 fragments
  auto phaseV = [&](int stage, int par) {
    const uint32_t aL = lds0 + LOFF + stage * 512 + (par * 32 + 4 * g) * 4;
    u32x4_t l4[2], d4[2];
    l4[0] = lds_read_b128_off<0>(aL); l4[1] = lds_read_b128_off<64>(aL);
    d4[0] = lds_read_b128_off<256>(aL); d4[1] = lds_read_b128_off<256 + 64>(aL);
    lgkm_wait<0>();
    pin(l4[0]); pin(l4[1]); pin(d4[0]); pin(d4[1]);
    f32x4_t ps[KF][2], ds[KF][2];
#pragma unroll
    for (int qf2 = 0; qf2 < 2; ++qf2) {
      const f32x2_t la = {__uint_as_float(l4[qf2].x), __uint_as_float(l4[qf2].y)}, lb = {__uint_as_float(l4[qf2].z), __uint_as_float(l4[qf2].w)};
      const f32x2_t na = {-__uint_as_float(d4[qf2].x), -__uint_as_float(d4[qf2].y)}, nb = {-__uint_as_float(d4[qf2].z), -__uint_as_float(d4[qf2].w)};
#pragma unroll
      for (int kf = 0; kf < KF; ++kf) {
        const f32x2_t xa = f32x2_t{sc[kf][qf2][0], sc[kf][qf2][1]} * sl2 - la;
        const f32x2_t xb = f32x2_t{sc[kf][qf2][2], sc[kf][qf2][3]} * sl2 - lb;
        const f32x2_t pa = {__builtin_amdgcn_exp2f(xa.x), __builtin_amdgcn_exp2f(xa.y)};
        const f32x2_t pc = {__builtin_amdgcn_exp2f(xb.x), __builtin_amdgcn_exp2f(xb.y)};
        const f32x2_t da = pa * (f32x2_t{dp[kf][qf2][0], dp[kf][qf2][1]} + na);
        const f32x2_t dc = pc * (f32x2_t{dp[kf][qf2][2], dp[kf][qf2][3]} + nb);
        ps[kf][qf2] = f32x4_t{pa.x, pa.y, pc.x, pc.y};
        ds[kf][qf2] = f32x4_t{da.x, da.y, dc.x, dc.y};
      }
    }
#pragma unroll
    for (int kf = 0; kf < KF; ++kf) { pb[kf] = PFrag<bf16_t>::make(ps[kf]); sb[kf] = PFrag<bf16_t>::make(ds[kf]); }
  };

  issue(0, 0);
  if (ntiles > 1) issue(1, 1);
  if (ntiles > 2) issue(2, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const auto T = std::true_type{};
  const auto F = std::false_type{};
  int stg = 0;                                     // ring stage of the tile of half hh
  if (grp == 0) {
    for (int hh = 0; hh < NH; ++hh) {
      const int par = hh & 1, sn = (stg == 2) ? 0 : stg + 1, sp = (stg == 0) ? 2 : stg - 1;
      if (par && hh >= 3 && (hh + 3) / 2 < ntiles) issue((hh + 3) / 2, sp);               // interval 2hh
      const uint32_t cur = lds0 + stg * STAGE, prv = lds0 + (par ? stg : sp) * STAGE;
      if (hh == 0) phaseM(F, T, cur, par, prv, par ^ 1); else phaseM(T, T, cur, par, prv, par ^ 1);
      __builtin_amdgcn_s_barrier();
      phaseV(stg, par);                                                                    // interval 2hh+1
      if (par) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (par) stg = sn;
    }
    { const int sp = (stg == 0) ? 2 : stg - 1; phaseM(T, F, 0u, 0, lds0 + sp * STAGE, 1); }   // interval 2NH
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();                                                          // interval 0: idle
    for (int hh = 0; hh < NH; ++hh) {
      const int par = hh & 1, sn = (stg == 2) ? 0 : stg + 1, sp = (stg == 0) ? 2 : stg - 1;
      const uint32_t cur = lds0 + stg * STAGE, prv = lds0 + (par ? stg : sp) * STAGE;
      if (hh == 0) phaseM(F, T, cur, par, prv, par ^ 1); else phaseM(T, T, cur, par, prv, par ^ 1);   // interval 2hh+1
      if (par) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (!par && hh >= 2 && hh / 2 + 2 < ntiles) issue(hh / 2 + 2, sp);                   // interval 2hh+2
      phaseV(stg, par);
      __builtin_amdgcn_s_barrier();
      if (par) stg = sn;
    }
    { const int sp = (stg == 0) ? 2 : stg - 1; phaseM(T, F, 0u, 0, lds0 + sp * STAGE, 1); }   // interval 2NH+1
  }

#pragma unroll
  for (int kf = 0; kf < KF; ++kf) {
    const int kr = kv_w + kf * 16 + lq;
    bf16_t* dkp = reinterpret_cast<bf16_t*>(p.dK) + ((long)b * p.Nkv + kr) * p.lddk + (long)h * DH;
    bf16_t* dvp = reinterpret_cast<bf16_t*>(p.dV) + ((long)b * p.Nkv + kr) * p.lddv + (long)h * DH;
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      const int d0 = i * 16 + 4 * g;
      if (d0 < DH) {
        float a[4] = {dkt[kf][i][0] * p.scale, dkt[kf][i][1] * p.scale, dkt[kf][i][2] * p.scale, dkt[kf][i][3] * p.scale};
        float c[4] = {dvt[kf][i][0], dvt[kf][i][1], dvt[kf][i][2], dvt[kf][i][3]};
        store4(dkp + d0, a);
        store4(dvp + d0, c);
      }
    }
  }
}

template <int DH, int QF, int LA = 3>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_pp_kernel(AttnBwdArgs p, int nqb, int remap) {
  using G = Geo<DH>;
  constexpr int CPR = G::CPR, KSTEPS = G::KSTEPS, DN = G::DN, ROWB = G::ROWB, TILE = G::TILE;
  constexpr int STAGE = 2 * TILE, SLACK = 16 * ROWB + 64;      // K tile, V tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int g = lane >> 4, lq = lane & 15;
  int bh, qbi;
  {
    const int id = blockIdx.x;
    if (remap) { const int xcd = id & 7, slot = id >> 3; bh = xcd + 8 * (slot / nqb); qbi = slot - (slot / nqb) * nqb; }
    else { bh = id / nqb; qbi = id - bh * nqb; }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int q_w = qbi * (128 * QF) + wave * (16 * QF);
  const float sl2 = p.scale * 1.4426950408889634f;

  for (int i = tid; i < SLACK / 4; i += 512) reinterpret_cast<uint32_t*>(smem + 3 * STAGE)[i] = 0u;
  init_pads<DH>(smem, 6, 0u, 0u, tid, 512);

  u32x4_t qb[QF][KSTEPS], ob[QF][KSTEPS];
  float lse_q[QF], ndl_q[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int qr = q_w + f * 16 + lq;
    const char* qp = (const char*)p.Q + (((long)b * p.N + qr) * p.ldq + (long)h * DH) * 2;
    const char* op = (const char*)p.dO + (((long)b * p.N + qr) * p.lddo + (long)h * DH) * 2;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = 4 * ks + g;
      qb[f][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(qp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
      ob[f][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(op + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
    lse_q[f] = p.LSE[((long)b * p.H + h) * p.lse_stride + qr];
    ndl_q[f] = -p.Delta[((long)b * p.H + h) * p.lse_stride + qr];
  }
  const char* kbase = (const char*)p.K + ((long)b * p.Nkv * p.ldk + (long)h * DH) * 2;
  const char* vbase = (const char*)p.V + ((long)b * p.Nkv * p.ldv + (long)h * DH) * 2;

  f32x4_t dqt[QF][DN];
#pragma unroll
  for (int f = 0; f < QF; ++f)
#pragma unroll
    for (int i = 0; i < DN; ++i) dqt[f][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  f32x4_t sc[QF][2], dp[QF][2];
  u32x4_t sb[QF];

  constexpr int CPRP = G::CPRP, NJ = (CPRP + 7) / 8;
  int koff[NJ], voff[NJ];
  bool real[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + 8 * j) * 64 + lane, r = c / CPRP, col = c - r * CPRP;
    real[j] = col < CPR;
    const int cc = (real[j] ? col : 0) * 16;
    koff[j] = r * (int)(p.ldk * 2) + cc;
    voff[j] = r * (int)(p.ldv * 2) + cc;
  }
  auto issue = [&](int t, int stage) {
    const char* kb = kbase + (long)t * 64 * p.ldk * 2;
    const char* vb = vbase + (long)t * 64 * p.ldv * 2;
    char* dst = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (wave + 8 * j < CPRP && real[j]) {
        glds16(kb + koff[j], dst + (wave + 8 * j) * 1024);
        glds16(vb + voff[j], dst + TILE + (wave + 8 * j) * 1024);
      }
  };

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t rrow = lq * ROWB + g * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const int ntiles = p.Nkv / 64, NH = 2 * ntiles;

  auto phaseM = [&](auto PREVc, auto CURc, uint32_t cur, int par, uint32_t prv, int parp) {
    constexpr bool PREV = decltype(PREVc)::value, CUR = decltype(CURc)::value;
    constexpr int NV = PREV ? 1 : 0, NA = CUR ? 2 : 0, NG = NV + NA;     // one transpose-read group (all DN fragments)
    u32x4_t kc[DN], ak[LA][KSTEPS], av[LA][KSTEPS];
    const uint32_t tk = prv + troff + parp * 32 * ROWB;
    const uint32_t rk = cur + rrow + par * 32 * ROWB, rv = rk + TILE;
    auto cnt_of = [](int j) constexpr { return j < NV ? 2 * DN : 2 * KSTEPS; };
    auto read_group = [&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      if constexpr (J < NV) {
        static_for<0, DN>([&](auto Ic) { constexpr int i = decltype(Ic)::value; kc[i] = tr_frag<ROWB, 0>(tk + i * 32); });
      } else if constexpr (J < NG) {
        constexpr int kf2 = J - NV;
        static_for<0, KSTEPS>([&](auto Kc) {
          constexpr int ks = decltype(Kc)::value;
          ak[kf2 % LA][ks] = lds_read_b128_off<kf2 * 16 * ROWB + ks * 64>(rk);
          av[kf2 % LA][ks] = lds_read_b128_off<kf2 * 16 * ROWB + ks * 64>(rv);
        });
      }
    };
    static_for<0, LA - 1>([&](auto Jc) { read_group(Jc); });
    static_for<0, NG>([&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      read_group(std::integral_constant<int, J + LA - 1>{});
      constexpr int pending = [&]() constexpr { int n = 0; for (int k = J + 1; k < J + LA && k < NG; ++k) n += cnt_of(k); return n; }();
      lgkm_wait<(pending > 15 ? 15 : pending)>();
      if constexpr (J < NV) {
#pragma unroll
        for (int i = 0; i < DN; ++i) pin(kc[i]);
#pragma unroll
        for (int i = 0; i < DN; ++i)
#pragma unroll
          for (int f = 0; f < QF; ++f) Mma<bf16_t>::run(kc[i], sb[f], dqt[f][i]);
      } else {
        constexpr int kf2 = J - NV;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) { pin(ak[kf2 % LA][ks]); pin(av[kf2 % LA][ks]); }
#pragma unroll
        for (int f = 0; f < QF; ++f) { sc[f][kf2] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[f][kf2] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
          for (int f = 0; f < QF; ++f) {
            Mma<bf16_t>::run(ak[kf2 % LA][ks], qb[f][ks], sc[f][kf2]);
            Mma<bf16_t>::run(av[kf2 % LA][ks], ob[f][ks], dp[f][kf2]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- vector phase: dS^T of the half (rows = keys 16 kf2 + 4g + r, col = query lq), registers only
  auto phaseV = [&]() {
#pragma unroll
    for (int f = 0; f < QF; ++f) {
      f32x4_t dst[2];
#pragma unroll
      for (int kf2 = 0; kf2 < 2; ++kf2) {
        const f32x2_t xa = f32x2_t{sc[f][kf2][0], sc[f][kf2][1]} * sl2 - lse_q[f];
        const f32x2_t xb = f32x2_t{sc[f][kf2][2], sc[f][kf2][3]} * sl2 - lse_q[f];
        const f32x2_t pa = {__builtin_amdgcn_exp2f(xa.x), __builtin_amdgcn_exp2f(xa.y)};
        const f32x2_t pc = {__builtin_amdgcn_exp2f(xb.x), __builtin_amdgcn_exp2f(xb.y)};
        const f32x2_t da = pa * (f32x2_t{dp[f][kf2][0], dp[f][kf2][1]} + ndl_q[f]);
        const f32x2_t dc = pc * (f32x2_t{dp[f][kf2][2], dp[f][kf2][3]} + ndl_q[f]);
        dst[kf2] = f32x4_t{da.x, da.y, dc.x, dc.y};
      }
      sb[f] = PFrag<bf16_t>::make(dst);
    }
  };

  issue(0, 0);
  if (ntiles > 1) issue(1, 1);
  if (ntiles > 2) issue(2, 2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const auto T = std::true_type{};
  const auto F = std::false_type{};
  int stg = 0;
  if (grp == 0) {
    for (int hh = 0; hh < NH; ++hh) {
      const int par = hh & 1, sn = (stg == 2) ? 0 : stg + 1, sp = (stg == 0) ? 2 : stg - 1;
      if (par && hh >= 3 && (hh + 3) / 2 < ntiles) issue((hh + 3) / 2, sp);
      const uint32_t cur = lds0 + stg * STAGE, prv = lds0 + (par ? stg : sp) * STAGE;
      if (hh == 0) phaseM(F, T, cur, par, prv, par ^ 1); else phaseM(T, T, cur, par, prv, par ^ 1);
      __builtin_amdgcn_s_barrier();
      phaseV();
      if (par) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (par) stg = sn;
    }
    { const int sp = (stg == 0) ? 2 : stg - 1; phaseM(T, F, 0u, 0, lds0 + sp * STAGE, 1); }
    __builtin_amdgcn_s_barrier();
  } else {
    __builtin_amdgcn_s_barrier();
    for (int hh = 0; hh < NH; ++hh) {
      const int par = hh & 1, sn = (stg == 2) ? 0 : stg + 1, sp = (stg == 0) ? 2 : stg - 1;
      const uint32_t cur = lds0 + stg * STAGE, prv = lds0 + (par ? stg : sp) * STAGE;
      if (hh == 0) phaseM(F, T, cur, par, prv, par ^ 1); else phaseM(T, T, cur, par, prv, par ^ 1);
      if (par) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (!par && hh >= 2 && hh / 2 + 2 < ntiles) issue(hh / 2 + 2, sp);
      phaseV();
      __builtin_amdgcn_s_barrier();
      if (par) stg = sn;
    }
    { const int sp = (stg == 0) ? 2 : stg - 1; phaseM(T, F, 0u, 0, lds0 + sp * STAGE, 1); }
  }

#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int qrow = q_w + f * 16 + lq;
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

// =============================================================================== host side
template <typename K>
static int set_lds(K kern, int bytes) {
  if (bytes > 65536 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
    return CL_ELAUNCH;
  return CL_OK;
}

int g_attn_variant = 0;    // probe hook: 1 = always the tile-synchronous kernels

// ping-pong forward: N a multiple of 256 queries, whole 64-key tiles, at least half a chip of workgroups
template <int DH, int LA, int ABL, bool ALONE = false, int VAR = 0>
static int launch_fwd_pp_t(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  // ALONE (probe): ask for more than half of the CU's LDS so that only one workgroup is resident per CU
  constexpr int LDS = ALONE ? 96 * 1024 : 3 * 2 * Geo<DH>::TILE + 16 * Geo<DH>::ROWB + 64;
  static bool done = false;
  if (!done) {
    if (set_lds(&attn_fwd_pp_kernel<DH, LA, ABL, VAR>, LDS)) return CL_ELAUNCH;
    done = true;
  }
  const int nqb = a.N / 256;
  const long grid = (long)nqb * a.H * a.B;
  const int remap = ((a.B * a.H) % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((attn_fwd_pp_kernel<DH, LA, ABL, VAR>), dim3((unsigned)grid), dim3(512), LDS, st, a, V, ldv, nqb, remap);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

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

template <int DH, int NWAVES, int AHEAD, int RING, bool SINGLE = false, bool FOLD = false>
static int launch_fwd_il_t(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  constexpr int LDS = RING * 2 * Geo<DH>::TILE + 16 * Geo<DH>::ROWB + 64;
  static bool done = false;
  if (!done) {
    if (set_lds(&attn_fwd_il_kernel<DH, NWAVES, AHEAD, RING, SINGLE, FOLD>, LDS)) return CL_ELAUNCH;
    done = true;
  }
  const int nqb = a.N / (32 * NWAVES);
  const long grid = (long)nqb * a.H * a.B;
  const int remap = ((a.B * a.H) % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL((attn_fwd_il_kernel<DH, NWAVES, AHEAD, RING, SINGLE, FOLD>), dim3((unsigned)grid), dim3(64 * NWAVES), LDS, st, a, V, ldv, nqb, remap);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

template <int DH>
static bool launch_fwd_pp(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st, int* rc) {
  if constexpr (DH == 40 || DH == 80) {
    const long grid = (long)(a.N / 256) * a.H * a.B;
    if (g_attn_variant == 1 || a.N % 256 || a.Nkv % 64 || a.Nkv < 128 || grid < 256) return false;
    switch (g_attn_variant) {
      case 13: *rc = launch_fwd_hyb_t<DH, 3>(a, V, ldv, st); break;                // 32x32x16 Q.K^T, lookahead 3 groups
      // 15 / 16 / 17: per-wave software pipeline (round-4 candidates, d_head 40 only: its fragment-read budget).  Measured at the
      // end of round 3 (profiles/r03_attention/fwd_wave_pipeline_v*.json): results bit-identical to the hybrid kernel; 8 waves per
      // workgroup, one workgroup per CU: 265 us (tile requests 1 ahead, variant 15) / 271 us (3 ahead, 17) against 244-247 us.
      // 16 = four waves per workgroup, three workgroups per CU (3 waves per SIMD): built, NOT yet run on a GPU.
      // 18 = S single-buffered and overwritten in place (4 waves per SIMD again): built, NOT yet run on a GPU.
      // 19 / 20 = 15 / 18 with the scale and -max folded into the matrix product: they expect Q PRE-MULTIPLIED by scale * log2(e)
      // and ignore `scale` (probe / A-B use only: tests/tools/attn_bench.py pre-scales Q for them): built, NOT yet run on a GPU.
      case 15: case 16: case 17: case 18: case 19: case 20:
        if constexpr (DH == 40) {
          if (g_attn_variant == 19) *rc = launch_fwd_il_t<DH, 8, 2, 4, false, true>(a, V, ldv, st);
          else if (g_attn_variant == 20) *rc = launch_fwd_il_t<DH, 8, 2, 4, true, true>(a, V, ldv, st);
          else if (g_attn_variant == 15) *rc = launch_fwd_il_t<DH, 8, 2, 4>(a, V, ldv, st);
          else if (g_attn_variant == 16) *rc = launch_fwd_il_t<DH, 4, 2, 4>(a, V, ldv, st);
          else if (g_attn_variant == 18) *rc = launch_fwd_il_t<DH, 8, 2, 4, true>(a, V, ldv, st);
          else *rc = launch_fwd_il_t<DH, 8, 4, 8>(a, V, ldv, st);
        } else *rc = launch_fwd_hyb_t<DH, 2>(a, V, ldv, st);
        break;
      case 14: *rc = launch_fwd_hyb_t<DH, 2>(a, V, ldv, st); break;                // ... lookahead 2          // 2, 5: A/B probes (tests/tools/attn_bench.py); 3 = ping-pong backward too
      case 2: *rc = launch_fwd_pp_t<DH, 2, 0>(a, V, ldv, st); break;
      case 5: *rc = launch_fwd_pp_t<DH, 4, 0, true>(a, V, ldv, st); break;
      case 6: *rc = launch_fwd_pp_t<DH, 4, 0, false, 1>(a, V, ldv, st); break;     // setprio around the matrix phase
      case 7: *rc = launch_fwd_pp_t<DH, 4, 0, false, 2>(a, V, ldv, st); break;     // static priority for the younger group
      case 8: *rc = launch_fwd_pp_t<DH, 4, 0, false, 4>(a, V, ldv, st); break;     // single-issue fp32 softmax math
      case 9: *rc = launch_fwd_pp_t<DH, 4, 0, false, 7>(a, V, ldv, st); break;     // all three
      case 10: *rc = launch_fwd_pp_t<DH, 4, 0, false, 3>(a, V, ldv, st); break;    // setprio + static priority
      case 12: *rc = launch_fwd_pp_t<DH, 4, 0>(a, V, ldv, st); break;              // round-2 form (no priorities)
      // default since round 3: the hybrid kernel (32x32x16 Q.K^T).  INTERLEAVED A/B on one box, 7 rounds, medians
      // (profiles/r03_attention/interleaved_*.json): B 8 x H 8, N 4096, d_head 40: 248.8 us (round-2 kernel) -> 240.4 us;
      // B = 32 (the DDIM shape) 972.9 -> 960.0 us; d_head 80: 38.1 -> 36.6 us / 129.0 -> 126.2 us.  The wave-priority
      // variants 6 / 7 / 10 are within +-1.5 % of the round-2 kernel and the single-issue softmax (8) is 9 % slower; the
      // 15-20 % "gains" a single-pass comparison showed for them were the first-variant clock ramp of the probe.
      default: *rc = launch_fwd_hyb_t<DH, 2>(a, V, ldv, st); break;
    }
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

template <int DH, bool TQ, bool TK>
static int launch_bwd_tr_sync(const AttnBwdArgs& a, hipStream_t st, bool skip_dq, bool fused_delta = false);

template <int DH, bool TQ, bool TK>
static int launch_bwd_tr(const AttnBwdArgs& a, hipStream_t st) {
  constexpr int KF = DH <= 40 ? 2 : 1;     // key / query fragments per wave (register budget: <= 256 VGPRs)
  constexpr int LDS_DKV = 2 * (2 * Geo<DH>::TILE + 512) + 64 + 16 * Geo<DH>::ROWB;
  constexpr int LDS_DQ = 2 * 2 * Geo<DH>::TILE + 64 + 16 * Geo<DH>::ROWB;
  static bool done = false;
  if (!done) {
    if (set_lds(&attn_bwd_dkv_tr_kernel<DH, KF, TQ>, LDS_DKV) || set_lds(&attn_bwd_dkv_tr_kernel<DH, 1, TQ>, LDS_DKV) ||
        set_lds(&attn_bwd_dq_tr_kernel<DH, KF, TK>, LDS_DQ) || set_lds(&attn_bwd_dq_tr_kernel<DH, 1, TK>, LDS_DQ))
      return CL_ELAUNCH;
    done = true;
  }
  // default path: the tile-synchronous dQ kernel forms delta itself and runs first; the ping-pong probe variant
  // (and anything that skips that kernel) keeps the separate delta launch
  if (g_attn_variant != 3 && g_attn_fuse_delta)
    return launch_bwd_tr_sync<DH, TQ, TK>(a, st, false, /*fused_delta=*/true);
  int rc = attn_delta(a, st);
  if (rc) return rc;
  if constexpr (!TQ && !TK && (DH == 40 || DH == 80)) {
    // ping-pong kernels: whole 64-row tiles on both sides, >= 3 tiles in the loop direction, enough workgroups
    constexpr int KFP = DH == 40 ? 2 : 1;
    constexpr int LDSP = 3 * 2 * Geo<DH>::TILE + 16 * Geo<DH>::ROWB + 64 + 3 * 512;
    static bool donep = false;
    if (!donep) {
      if (set_lds(&attn_bwd_dkv_pp_kernel<DH, KFP>, LDSP) || set_lds(&attn_bwd_dq_pp_kernel<DH, KFP>, LDSP)) return CL_ELAUNCH;
      donep = true;
    }
    const int remap = ((a.B * a.H) % 8 == 0) ? 1 : 0;
    bool dkv_done = false, dq_done = false;
    // (measured on MI355X, profiles/r02_attention_ab.json: the ping-pong BACKWARD kernels are correct but not faster than
    // the tile-synchronous ones -- one 8-wave workgroup per CU at 172-184 VGPRs, both pipes under 50 % -- so they are
    // opt-in, variant 3, until the in-wave MFMA/VALU interleave replaces the barrier-phased form)
    const bool want_pp = g_attn_variant == 3;
    if (want_pp && a.dK && a.Nkv % (128 * KFP) == 0 && a.N >= 192) {
      const int nkb = a.Nkv / (128 * KFP);
      const long grid = (long)nkb * a.H * a.B;
      if (grid >= 128) {
        hipLaunchKernelGGL((attn_bwd_dkv_pp_kernel<DH, KFP>), dim3((unsigned)grid), dim3(512), LDSP, st, a, nkb, remap);
        dkv_done = true;
      }
    }
    if (want_pp && a.N % (128 * KFP) == 0 && a.Nkv >= 192) {
      const int nqb = a.N / (128 * KFP);
      const long grid = (long)nqb * a.H * a.B;
      if (grid >= 128) {
        hipLaunchKernelGGL((attn_bwd_dq_pp_kernel<DH, KFP>), dim3((unsigned)grid), dim3(512), LDSP, st, a, nqb, remap);
        dq_done = true;
      }
    }
    if ((dkv_done || !a.dK) && dq_done) { CL_CHECK_LAUNCH(); return CL_OK; }
    if (dkv_done || dq_done) {     // mixed: finish with the tile-synchronous kernel for the other half
      AttnBwdArgs a2 = a;
      if (dkv_done) { a2.dK = nullptr; a2.dV = nullptr; }
      return launch_bwd_tr_sync<DH, TQ, TK>(a2, st, /*skip_dq=*/dq_done);
    }
  }
  return launch_bwd_tr_sync<DH, TQ, TK>(a, st, false);
}

template <int DH, bool TQ, bool TK>
static int launch_bwd_tr_sync(const AttnBwdArgs& a, hipStream_t st, bool skip_dq, bool fused_delta) {
  constexpr int KF = DH <= 40 ? 2 : 1;
  constexpr int LDS_DKV = 2 * (2 * Geo<DH>::TILE + 512) + 64 + 16 * Geo<DH>::ROWB;
  constexpr int LDS_DQ = 2 * 2 * Geo<DH>::TILE + 64 + 16 * Geo<DH>::ROWB;
  if (fused_delta) {   // dQ (+ delta) first, then dK/dV
    static bool done = false;
    if (!done) {
      if (set_lds(&attn_bwd_dq_tr_kernel<DH, KF, TK, true>, LDS_DQ) || set_lds(&attn_bwd_dq_tr_kernel<DH, 1, TK, true>, LDS_DQ) ||
          set_lds(&attn_bwd_dq_tr_kernel<DH, KF, TK, true, true>, LDS_DQ) || set_lds(&attn_bwd_dkv_tr_kernel<DH, KF, TQ, true>, LDS_DKV))
        return CL_ELAUNCH;
      done = true;
    }
    const long qb2 = (long)((a.N + 64 * KF - 1) / (64 * KF)) * a.H * a.B;
    if (KF == 2 && qb2 >= 512) {
      dim3 grid((a.N + 127) / 128, a.H, a.B);
      if (g_attn_variant == 11) hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, KF, TK, true, true>), grid, dim3(256), LDS_DQ, st, a);
      else hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, KF, TK, true>), grid, dim3(256), LDS_DQ, st, a);
    } else {
      dim3 grid((a.N + 63) / 64, a.H, a.B);
      hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, 1, TK, true>), grid, dim3(256), LDS_DQ, st, a);
    }
    skip_dq = true;
  }
  if (a.dK) {
    // two key fragments per wave only when that still leaves enough workgroups to fill the chip
    const long blocks2 = (long)((a.Nkv + 64 * KF - 1) / (64 * KF)) * a.H * a.B;
    if (KF == 2 && blocks2 >= 512) {
      dim3 grid((a.Nkv + 127) / 128, a.H, a.B);
      if (g_attn_variant == 11) hipLaunchKernelGGL((attn_bwd_dkv_tr_kernel<DH, KF, TQ, true>), grid, dim3(256), LDS_DKV, st, a);
      else hipLaunchKernelGGL((attn_bwd_dkv_tr_kernel<DH, KF, TQ>), grid, dim3(256), LDS_DKV, st, a);
    } else {
      dim3 grid((a.Nkv + 63) / 64, a.H, a.B);
      hipLaunchKernelGGL((attn_bwd_dkv_tr_kernel<DH, 1, TQ>), grid, dim3(256), LDS_DKV, st, a);
    }
  }
  if (skip_dq) { CL_CHECK_LAUNCH(); return CL_OK; }
  const long qblocks2 = (long)((a.N + 64 * KF - 1) / (64 * KF)) * a.H * a.B;
  if (KF == 2 && qblocks2 >= 512) {
    dim3 grid((a.N + 127) / 128, a.H, a.B);
    hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, KF, TK>), grid, dim3(256), LDS_DQ, st, a);
  } else {
    dim3 grid((a.N + 63) / 64, a.H, a.B);
    hipLaunchKernelGGL((attn_bwd_dq_tr_kernel<DH, 1, TK>), grid, dim3(256), LDS_DQ, st, a);
  }
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
