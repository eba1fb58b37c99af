// d_head-40 self-attention forward for a PRE-SCALED Q (CL_ATTN_Q_PRESCALED): the 64x64 level of SD1.5, where the attention
// time of a training / denoising step is (CrossAttention.forward, ldm/modules/attention.py:171-192).
//
// What bounds this product on gfx950 is VALU ISSUE next to the matrix pipe, and the two ADD on a SIMD (measured again here:
// see the ablations below).  Per wave and 32-query x 64-key step the hybrid kernel of attention_tr.hip issues ~125 vector
// instructions (32 max, 16 packed fma, 32 exp2, 16 cvt_pk, 8 lane swaps, moves, addresses) beside 18 MFMAs.  This kernel issues 49:
//   * no per-score multiply-add: Q arrives multiplied by d_head^-0.5 * log2(e) (the to_q projection's fp32 epilogue did it:
//     ONE rounding, like any stored q), and -m travels through the matrix product -- the 48-deep walk of d_head 40 has 8
//     spare contraction slots: Q column 40 holds -m (bf16), K's pad column 40 holds 1.0 -- so the accumulators deliver
//     s - m and the softmax starts at v_exp_f32;
//   * no running maximum after the first tile ("optimistic" pass): m is the row maximum of tile 0, rounded to bf16.  The
//     maximum is subtracted for RANGE, not for precision -- exp2, the bf16 rounding of P and the fp32 accumulation of P V
//     and sum P are all relative -- so a later score may exceed m by up to ~2^100 before anything overflows.  The epilogue
//     checks every row's denominator (it comes out of the matrix pipe: V^T's spare row 40 is ones) and, if any row of the
//     workgroup left [1e-30, 1e30], the whole workgroup repeats the block with a plain sequential pass that tracks the
//     maximum: correct for any input, one extra (slow) pass for inputs whose scores spread over > 60 nats;
//   * no lane swaps, no moves: BOTH products use v_mfma_f32_32x32x16_bf16.  S^T = K Q^T leaves lane (query, key half) with 16
//     scores per 32-key block, and exactly those registers, packed to bf16 IN PLACE, are the B operand of O^T += V^T P^T
//     (contraction index = key, permuted consistently with the transpose reads of V).  d_head 40 pays 48 of 48 contraction
//     slots in S and 40 of 64 output rows in P.V (448 instead of 384 matrix cycles per step: cheaper than the 26 VALU it saves).
// Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD"): a workgroup of four waves, each owning 64 queries as
// two independent 32-query blocks A and B, so that ONE instruction stream always has a ready MFMA of one block and ready
// softmax VALU of the other, ~3.4 VALU behind every MFMA; K / V fragments are read from LDS once per tile and serve both blocks:
//
//   per 64-key tile t and wave, 28 MFMAs in this order
//     gaps  0- 7   O_A^T += V(t)^T P_A(t)^T            8 MFMAs  |  exp2 / pack of block B, tile t        (first 8/14)
//     gaps  8-13   S_A^T(t+1) = K(t+1) Q_A^T           6 MFMAs  |  exp2 / pack of block B, tile t        (rest)
//     gaps 14-21   O_B^T += V(t)^T P_B(t)^T            8 MFMAs  |  exp2 / pack of block A, tile t+1      (first 8/14)
//     gaps 22-27   S_B^T(t+1) = K(t+1) Q_B^T           6 MFMAs  |  exp2 / pack of block A, tile t+1      (rest) + V(t+1) fragment reads
//
// 232 registers, no AGPRs (every MFMA is inline asm with "v" operands: with the builtin the allocator parked the scores in
// accumulation registers, 64 v_accvgpr_read per tile): two workgroups per CU, two waves per SIMD.  One s_barrier per tile,
// K / V tiles ride a 4-slot LDS-DMA ring.
// Measured (MI355X, B x H = 64, N = 4096, interleaved A/B, profiles/r04_attention/): hybrid kernel 248.7 us -> 198.1 us (867 TF/s,
// 35 % of the bf16 MFMA peak); B x H = 256: 983.7 -> 780.4 us.  Ablations of THIS kernel (same visit, results wrong by
// construction): without its MFMAs 105.5 us, without exp2 186 us, without LDS fragment reads 185 us, without barrier + DMA
// 179 us; the MFMAs alone are 28 x 32 cycles x 64 tiles x 4 workgroups per CU = 109 us at 2.1 GHz -- so vector and matrix
// work of a SIMD overlap by ~15 us of 105: they add, with one wave per SIMD (perfectly interleaved stream, 220 us) as with two.
// An 8-wave compiler-scheduled form of the same arithmetic (one 32-query block per wave, 16x16x32 P.V, four waves per SIMD)
// measured 207 us and was removed.
#include <type_traits>
#include "attn_common.h"
#include "attn_tr_util.h"

namespace cl {

namespace {

constexpr int X40_DH = 40;
using GX = Geo<X40_DH>;
constexpr int X40_RING = 4, X40_NW = 4, X40_THREADS = 64 * X40_NW;
constexpr int X40_STAGE = 2 * GX::TILE;
constexpr int X40_SLACK = 32 * GX::ROWB + 64;       // zeros behind the ring: fragment reads of the last rows run past a tile
constexpr int X40_LDS = X40_RING * X40_STAGE + X40_SLACK;

// VALU op k (0..47) of one block-tile's softmax: chunk c = k / 3 -> exp2 of score 2c, exp2 of score 2c + 1, pack.
// Gap g (0..13) of a half iteration carries ops [x40_op_end(g - 1), x40_op_end(g)).
constexpr int x40_op_end(int g) { return g < 0 ? 0 : (48 * (g + 1)) / 14; }

// S^T accumulators live in ARCHITECTURAL registers (the exp2s read them): c = a . b, then c += a . b
__device__ __forceinline__ void x40_mfma_v0(f32x16_t& c, const u32x4_t& a, const u32x4_t& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void x40_mfma_v(f32x16_t& c, const u32x4_t& a, const u32x4_t& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// The compiler does not know an inline-asm MFMA's latency: before anything but an MFMA reads its result, 11 wait states
// (8 passes + 3) must pass.  The drain takes the accumulators it protects as in/out operands: to the compiler an asm
// statement that does not mention them is no obstacle for their readers -- the first version of this drain ("s_nop" with a
// memory clobber only) had the epilogue's ds_bpermute of the denominator row scheduled BEFORE it, one s_nop 0 behind the
// last P.V MFMA; a stale denominator sent workgroups into the second pass at random (correct results, different roundings:
// 1-ulp flips in ~2 % of the outputs from run to run; found by tests/tools/debug_determinism.py).
__device__ __forceinline__ void x40_mfma_drain(f32x16_t& a, f32x16_t& b, f32x16_t& c, f32x16_t& d) {
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void x40_mfma_drain(f32x16_t& a, f32x16_t& b) {
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b));
}
// c += a . b (O^T accumulators).  Inline asm keeps every MFMA operand in architectural registers: with the builtin the allocator
// parked the SCORES in accumulation registers (64 v_accvgpr_read per tile in front of the exp2s); with "v" everywhere the
// kernel needs 224 registers, no AGPRs, and two workgroups share a CU (two waves per SIMD).
// GUARD: the compiler may have just MOVED the accumulator (it assigns different registers to O^T in the last-iteration
// instance and shuffles them right in front of the first MFMA -- found on hardware: register 0 of one accumulator came
// out stale) and it inserts no wait states in front of inline asm.
template <bool GUARD> __device__ __forceinline__ void x40_mfma_acc(f32x16_t& c, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (GUARD) asm volatile("s_nop 7\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

}  // namespace

__global__ __launch_bounds__(X40_THREADS, 2) void attn_fwd40_kernel(AttnFwdArgs p, const void* __restrict__ V, long ldv, int nqb,
                                                                     int remap) {
  constexpr int CPR = GX::CPR, CPRP = GX::CPRP, ROWB = GX::ROWB, TILE = GX::TILE;
  constexpr int STAGE = X40_STAGE, RING = X40_RING, NW = X40_NW, NK = 3, DH = X40_DH;
  constexpr int NJ = (CPRP + NW - 1) / NW;
  constexpr float RESCALE_THR = 6.0f;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l31 = lane & 31, hi = lane >> 5;
  int bh, qb;
  {
    const int id = blockIdx.x;
    if (remap) { const int xcd = id & 7, slot = id >> 3; bh = xcd + 8 * (slot / nqb); qb = slot - (slot / nqb) * nqb; }
    else { bh = id / nqb; qb = id - bh * nqb; }
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = qb * (64 * NW) + wave * 64;           // block A: q0 .. q0 + 31, block B: q0 + 32 .. q0 + 63

  for (int i = tid; i < X40_SLACK / 4; i += X40_THREADS) reinterpret_cast<uint32_t*>(smem + RING * STAGE)[i] = 0u;
  for (int i = tid; i < 2 * RING * 64; i += X40_THREADS) {   // K pad = (1.0, 0, ...): column 40 meets -m in Q; V pad = ones
    const bool vtile = (i >> 6) & 1;
    const uint4 w = vtile ? make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u) : make_uint4(0x00003F80u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(smem + (long)i * ROWB + CPR * 16) = w;
  }

  // Q fragments (B operand of S^T = K Q^T): col = query l31 of the block, k = 16 j + 8 hi .. + 7; chunk 5 = columns 40..47 = 0
  u32x4_t qh[2][NK];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const char* qp = (const char*)p.Q + (((long)b * p.N + q0 + 32 * x + l31) * p.ldq + (long)h * DH) * 2;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int c = 2 * j + hi;
      qh[x][j] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(qp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
  }
  const char* kbase = (const char*)p.K + ((long)b * p.Nkv * p.ldk + (long)h * DH) * 2;
  const char* vbase = (const char*)V + ((long)b * p.Nkv * ldv + (long)h * DH) * 2;

  int koff[NJ], voff[NJ];
  bool real[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = (wave + NW * j) * 64 + lane, r = c / CPRP, col = c - r * CPRP;
    real[j] = col < CPR;
    const int cc = (real[j] ? col : 0) * 16;
    koff[j] = r * (int)(p.ldk * 2) + cc;
    voff[j] = r * (int)(ldv * 2) + cc;
  }
  auto issue = [&](int t, int slot) {
    const char* kb = kbase + (long)t * 64 * p.ldk * 2;
    const char* vb = vbase + (long)t * 64 * ldv * 2;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (wave + NW * j < CPRP && real[j]) {
        char* dst = smem + slot * STAGE + (wave + NW * j) * 1024;
        glds16(kb + koff[j], dst);
        glds16(vb + voff[j], dst + TILE);
      }
  };
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // K fragment (A operand, M = key row l31 of a 32-key block, k = chunk 2 j + hi): rows are read in place
  const uint32_t krow = l31 * ROWB + hi * 16;
  // V^T fragment (A operand, M = d row 16 (g & 1) + (lane & 15) of a 32-row block, k = keys): lane group g transposes the 4 key
  // rows 4 (g >> 1) + 0..3 (second read: + 8) x 16 columns 16 (g & 1) ..; P's registers hold exactly those keys
  const uint32_t vrow = (4 * (g >> 1) + ((lane >> 2) & 3)) * ROWB + (g & 1) * 32 + (lane & 3) * 8;
  const int nt = p.Nkv / 64;

  f32x16_t sc[2][2];                   // [block][key half]: S^T - m, keys 32 s + (r & 3) + 8 (r >> 2) + 4 hi, query l31
  f32x16_t oT[2][2];                   // [block][d block]: O^T rows 32 db + (r & 3) + 8 (r >> 2) + 4 hi, query l31
  uint32_t pP[2][16];                  // [block]: packed P, word c = scores (2c, 2c + 1) of half c / 8: words 4i .. 4i + 3 = contraction step i
  float m_run[2];
  uint32_t k_addr, v_addr;             // fragment addresses of the tiles in hand: K(t+1), V(t) (loop-carried, wave-uniform steps)
  u32x4_t va[4][2];                    // V^T fragments [contraction step i][d block]
  u32x4_t ka[2][NK];

  auto set_q_minus_m = [&](int x, uint32_t mb) {
    const uint32_t w = mb ? (mb ^ 0x8000u) : 0u;
    if (hi) qh[x][NK - 1][0] = w;
  };
  // V^T fragment read r (0..15): contraction step r / 4, d block (r / 2) % 2, first / second row quad r % 2 (keys +0..3 / +8..11)
  auto req_v1 = [&](auto Rc) {
    constexpr int r = decltype(Rc)::value, i = r / 4, db = (r / 2) % 2, half = r % 2;
    const u32x2_t w = tr_read<i * 16 * ROWB + half * 8 * ROWB + db * 64>(v_addr);
    if constexpr (half == 0) { va[i][db].x = w.x; va[i][db].y = w.y; }
    else { va[i][db].z = w.x; va[i][db].w = w.y; }
  };
  auto req_v = [&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    static_for<4 * i, 4 * i + 4>([&](auto Rc) { req_v1(Rc); });
  };
  auto req_k = [&]() {
    static_for<0, 2>([&](auto Sc) {
      constexpr int s_ = decltype(Sc)::value;
      static_for<0, NK>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        ka[s_][j] = lds_read_b128_off<s_ * 32 * ROWB + j * 32>(k_addr);
      });
    });
  };
  auto pin_v = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int db = 0; db < 2; ++db) pin(va[i][db]);
  };
  auto pin_k = [&]() {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int j = 0; j < NK; ++j) pin(ka[s_][j]);
  };
  // MFMA q (0..7) of O_x^T += V^T P_x^T: step i = q / 2, d block q % 2 (consecutive MFMAs alternate accumulators)
  auto pv_mfma = [&](auto Xc, auto Qc, auto Gd) {
    constexpr int x = decltype(Xc)::value, q = decltype(Qc)::value, i = q / 2, db = q % 2;
    constexpr bool GUARD = decltype(Gd)::value;
    const u32x4_t pb = u32x4_t{pP[x][4 * i], pP[x][4 * i + 1], pP[x][4 * i + 2], pP[x][4 * i + 3]};
    // O^T lives in ACCUMULATION registers by construction (inline asm, "+a"): the wave holds > 256 live values, and left to
    // itself the allocator parks the SCORES there -- 64 v_accvgpr_read per tile in front of the exp2s.  Nothing but these
    // MFMAs touches O^T until the epilogue (same opcode, same vDst as SrcC: no wait states needed between them).
    x40_mfma_acc<GUARD>(oT[x][db], va[i][db], pb);
  };
  // MFMA q (0..5) of S_x^T = K Q_x^T: contraction step j = q / 2, key half q % 2
  auto s_mfma = [&](auto Xc, auto Qc) {
    constexpr int x = decltype(Xc)::value, q = decltype(Qc)::value, j = q / 2, s_ = q % 2;
    // (inline asm, "v": see pv_mfma.  In the pipelined pass the first exp2 of these scores is issued two MFMAs -- > 64 cycles --
    // after the last of them; the prologue and the sequential pass drain explicitly.)
    if constexpr (j == 0) x40_mfma_v0(sc[x][s_], ka[s_][j], qh[x][j]);
    else x40_mfma_v(sc[x][s_], ka[s_][j], qh[x][j]);
  };
  // softmax VALU op k (0..47) of block x: two exp2 and one pack per chunk.  The pack of chunk c is issued one chunk LATE (in
  // chunk c + 1's third slot; chunk 15's right behind it): v_cvt_pk_bf16_f32 straight behind the v_exp_f32 that feeds it
  // stalls the wave for the transcendental's latency, and with one or two waves per SIMD nobody else fills that hole
  float e_lo[2][2], e_hi[2][2];        // [block][chunk parity]
  auto sm_op = [&](auto Xc, auto Kc) {
    constexpr int x = decltype(Xc)::value, k = decltype(Kc)::value, c = k / 3, w = k % 3, s_ = c / 8, r = 2 * (c % 8);
    if constexpr (w == 0) e_lo[x][c & 1] = __builtin_amdgcn_exp2f(sc[x][s_][r]);
    else if constexpr (w == 1) e_hi[x][c & 1] = __builtin_amdgcn_exp2f(sc[x][s_][r + 1]);
    else {
      if constexpr (c > 0) pP[x][c - 1] = pack2bf(e_lo[x][(c - 1) & 1], e_hi[x][(c - 1) & 1]);
      if constexpr (c == 15) pP[x][15] = pack2bf(e_lo[x][1], e_hi[x][1]);
    }
  };
  auto sm_ops = [&](auto Xc, auto LOc, auto HIc) {
    static_for<decltype(LOc)::value, decltype(HIc)::value>([&](auto Kc) { sm_op(Xc, Kc); });
  };
  // tile-0 maximum of block x: m = bf16(row maximum); scores and Q column 40 re-based
  auto first_max = [&](auto Xc) {
    constexpr int x = decltype(Xc)::value;
    float mx = sc[x][0][0];
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[x][s_][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const uint32_t mb = pack2bf(mx, 0.f) & 0xffffu;
    const float m_b = __uint_as_float(mb << 16);
    m_run[x] = m_b;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[x][s_][r] -= m_b;
    set_q_minus_m(x, mb);
  };
  auto zero_state = [&]() {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
#pragma unroll
      for (int db = 0; db < 2; ++db) oT[x][db] = f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      m_run[x] = 0.f;
      set_q_minus_m(x, 0u);
    }
  };
  using XA = std::integral_constant<int, 0>;
  using XB = std::integral_constant<int, 1>;

  // ================================================================ optimistic, software-pipelined pass
  auto run_fast = [&]() {
    zero_state();
    issue(0, 0);
    issue(1, 1);                                           // (nt >= 2: launcher)
    dma_wait();
    __syncthreads();
    k_addr = lds0 + krow;                                  // K(0)
    v_addr = lds0 + TILE + vrow;                           // V(0)
    req_k();
    lgkm_wait<0>();
    pin_k();
    static_for<0, 6>([&](auto Qc) { s_mfma(XA{}, Qc); });
    static_for<0, 6>([&](auto Qc) { s_mfma(XB{}, Qc); });
    x40_mfma_drain(sc[0][0], sc[0][1], sc[1][0], sc[1][1]);
    first_max(XA{});
    first_max(XB{});
    static_for<0, 4>([&](auto Ic) { req_v(Ic); });         // V(0) fragments for iteration 0 (16 reads: waited for at its top)
    sm_ops(XA{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 48>{});     // P_A(0); block B's follows in iteration 0
    k_addr += STAGE;                                       // K(1)
    __builtin_amdgcn_sched_barrier(0);

    auto iteration = [&](auto NEXTc, int t) {
      constexpr bool HAS_NEXT = decltype(NEXTc)::value;    // tile t + 1 exists: S^T of it, softmax of block A's
      dma_wait();                                          // tile t + 1 (requested in iteration t - 1) has landed
      __builtin_amdgcn_s_barrier();
      if (t + 2 < nt) issue(t + 2, (t + 2) & (RING - 1));
      if constexpr (HAS_NEXT) req_k();                     // K(t+1): needed from gap 8 on
      lgkm_wait<(HAS_NEXT ? 2 * NK : 0)>();                // V(t) fragments (requested in the previous iteration's tail) landed
      pin_v();
      // ---- first half: block A's P.V and next S^T | block B's softmax of tile t
      static_for<0, 8>([&](auto Gc) {
        constexpr int gp = decltype(Gc)::value;
        pv_mfma(XA{}, Gc, std::integral_constant<bool, (!HAS_NEXT || gp < 2)>{});     // (guarded at region entry: see x40_mfma_acc)
        __builtin_amdgcn_sched_barrier(0);       // the gap's VALU stays BEHIND its MFMA (see the note at the second half)
        constexpr int lo = HAS_NEXT ? x40_op_end(gp - 1) : 6 * gp, hi_ = HAS_NEXT ? x40_op_end(gp) : 6 * (gp + 1);
        sm_ops(XB{}, std::integral_constant<int, lo>{}, std::integral_constant<int, hi_>{});
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (HAS_NEXT) {
        lgkm_wait<0>();
        pin_k();
        static_for<0, 6>([&](auto Gc) {
          constexpr int gp = decltype(Gc)::value;
          s_mfma(XA{}, Gc);
          __builtin_amdgcn_sched_barrier(0);
          sm_ops(XB{}, std::integral_constant<int, x40_op_end(7 + gp)>{}, std::integral_constant<int, x40_op_end(8 + gp)>{});
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      // ---- second half: block B's P.V and next S^T | block A's softmax of tile t + 1
      // V(t+1) fragment reads ride on gaps 16..27 (2 + 12 r / 16): the registers of contraction step i are free once its two
      // P.V MFMAs of block B are out (gap 15 + 2 i)
      if constexpr (HAS_NEXT) v_addr += ((t + 1) & (RING - 1)) ? STAGE : -(RING - 1) * STAGE;      // V(t+1)
      auto v_reads_after = [&](auto Gc) {                  // Gc = gap index within the second half (0..13)
        constexpr int gp = decltype(Gc)::value;
        static_for<0, 16>([&](auto Rc) {
          constexpr int r = decltype(Rc)::value;
          if constexpr (2 + (12 * r) / 16 == gp) req_v1(Rc);
        });
      };
      static_for<0, 8>([&](auto Gc) {
        constexpr int gp = decltype(Gc)::value;
        pv_mfma(XB{}, Gc, std::integral_constant<bool, !HAS_NEXT>{});
        // Pin the gap's exp2 / pack work BEHIND its MFMA.  Without this the compiler is free to move the builtin VALU of a gap in
        // front of the gap's inline-asm MFMA (it sees no dependence), and it did: the first exp2 of block A's new scores then
        // sat ONE matrix instruction behind the S^T MFMA that writes them instead of two -- 10 wait states where the 8-pass
        // MFMA needs 11 (tools/isa_mfma_hazards.py; no wrong result was ever observed, the margin was simply gone).
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HAS_NEXT) {
          v_reads_after(Gc);
          sm_ops(XA{}, std::integral_constant<int, x40_op_end(gp - 1)>{}, std::integral_constant<int, x40_op_end(gp)>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (HAS_NEXT) {
        static_for<0, 6>([&](auto Gc) {
          constexpr int gp = decltype(Gc)::value;
          s_mfma(XB{}, Gc);
          __builtin_amdgcn_sched_barrier(0);
          v_reads_after(std::integral_constant<int, 8 + gp>{});
          sm_ops(XA{}, std::integral_constant<int, x40_op_end(7 + gp)>{}, std::integral_constant<int, x40_op_end(8 + gp)>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        k_addr += ((t + 2) & (RING - 1)) ? STAGE : -(RING - 1) * STAGE;      // K(t+2)
      }
    };
    for (int t = 0; t + 1 < nt; ++t) iteration(std::true_type{}, t);
    iteration(std::false_type{}, nt - 1);
  };

  // ================================================================ conventional pass (running maximum), sequential: correctness net
  auto run_safe = [&]() {
    zero_state();
    issue(0, 0);
    issue(1, 1);
    for (int t = 0; t < nt; ++t) {
      dma_wait();
      __syncthreads();
      if (t >= 1 && t + 1 < nt) issue(t + 1, (t + 1) & (RING - 1));        // (tile t - 1's slot is free: 2 tiles live)
      k_addr = lds0 + (t & (RING - 1)) * STAGE + krow;
      v_addr = lds0 + (t & (RING - 1)) * STAGE + TILE + vrow;
      req_k();
      static_for<0, 4>([&](auto Ic) { req_v(Ic); });
      lgkm_wait<0>();
      pin_k(); pin_v();
      static_for<0, 2>([&](auto Xc) {
        constexpr int x = decltype(Xc)::value;
        static_for<0, 6>([&](auto Qc) { s_mfma(Xc, Qc); });
        x40_mfma_drain(sc[x][0], sc[x][1]);
        float mx = sc[x][0][0];
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[x][s_][r]);
        if (t == 0 || __any(mx > RESCALE_THR)) {            // sc holds s - m_run
          mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
          const float m_new = t == 0 ? mx : fmaxf(m_run[x], m_run[x] + mx);
          const uint32_t mb = pack2bf(m_new, 0.f) & 0xffffu;
          const float m_b = __uint_as_float(mb << 16);
          const float d = m_b - m_run[x];
          const float alpha = t == 0 ? 1.0f : __builtin_amdgcn_exp2f(-d);
          m_run[x] = m_b;
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int r = 0; r < 16; ++r) sc[x][s_][r] -= d;
          x40_mfma_drain(oT[x][0], oT[x][1]);                            // (inline-asm MFMAs wrote O^T: see pv_mfma)
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) oT[x][db][r] *= alpha;           // column = query = this lane
          set_q_minus_m(x, mb);
        }
        sm_ops(Xc, std::integral_constant<int, 0>{}, std::integral_constant<int, 48>{});
        static_for<0, 8>([&](auto Qc) { pv_mfma(Xc, Qc, std::true_type{}); });
      });
    }
  };

  run_fast();
  x40_mfma_drain(oT[0][0], oT[0][1], oT[1][0], oT[1][1]);  // (the last P.V MFMAs are inline asm: their results are read below)
  // every row's denominator (O^T row 40: d block 1, register 4 of the lower half-wave) must be an ordinary number
  float lsum[2];
  auto denominators = [&]() {
    bool bad = false;
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      lsum[x] = __shfl(oT[x][1][4], l31, 64);
      bad |= !(lsum[x] > 1e-30f && lsum[x] < 1e30f);
    }
    return bad;
  };
  if (__syncthreads_or(denominators() ? 1 : 0)) {
    run_safe();
    x40_mfma_drain(oT[0][0], oT[0][1], oT[1][0], oT[1][1]);
    denominators();
  }

  // ---- epilogue: normalise, store O rows, log-sum-exp (log2 domain).  Lane (query l31, hi) holds d = 4 hi + {0-3, 8-11, 16-19,
  // 24-27} of d block 0 and d = 32 + 4 hi + {0-3} of d block 1
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const float inv = 1.0f / lsum[x];
    const int row = q0 + 32 * x + l31;
    bf16_t* op = reinterpret_cast<bf16_t*>(p.O) + ((long)b * p.N + row) * p.ldo + (long)h * DH + 4 * hi;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      float v[4] = {oT[x][0][4 * q4] * inv, oT[x][0][4 * q4 + 1] * inv, oT[x][0][4 * q4 + 2] * inv, oT[x][0][4 * q4 + 3] * inv};
      store4(op + 8 * q4, v);
    }
    float v[4] = {oT[x][1][0] * inv, oT[x][1][1] * inv, oT[x][1][2] * inv, oT[x][1][3] * inv};
    store4(op + 32, v);
    if (p.LSE && hi == 0) p.LSE[((long)b * p.H + h) * p.lse_stride + row] = m_run[x] + __builtin_amdgcn_logf(lsum[x]);
  }
}

// The pre-scaled-Q forward applies when: Q is pre-scaled, d_head 40, whole 256-query blocks, whole 64-key tiles (>= 2),
// at least a chip's worth of workgroups.  Returns false when it does not (the caller falls through to the other kernels).
bool attn_fwd40_applies(const AttnFwdArgs& a) {
  const long grid = (long)(a.N / 256) * a.H * a.B;
  return a.q_prescaled && a.DH == X40_DH && a.N % 256 == 0 && a.Nkv % 64 == 0 && a.Nkv >= 128 && grid >= 256;
}

int attn_fwd40(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  static_assert(X40_LDS <= 65536, "no dynamic-LDS attribute needed");
  const int nqb = a.N / 256;
  const long grid = (long)nqb * a.H * a.B;
  const int remap = ((a.B * a.H) % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL(attn_fwd40_kernel, dim3((unsigned)grid), dim3(X40_THREADS), X40_LDS, st, a, V, ldv, nqb, remap);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

}  // namespace cl
