// d_head-40 self-attention forward for a PRE-SCALED Q (CL_ATTN_Q_PRESCALED): the 64x64 level of SD1.5, where the attention
// time of a training / denoising step is (CrossAttention.forward, ldm/modules/attention.py:171-192).
//
// What bounds this product on gfx950 is VALU ISSUE, not the matrix pipe: per wave and 64-key step the
// hybrid kernel of attention_tr.hip issues ~125 vector instructions (32 max, 16 packed fma, 32 exp2, 16 cvt_pk, 8 lane swaps,
// address arithmetic) beside 18 MFMAs, and every one of them costs an issue slot of >= 4 cycles.  This kernel removes 48 of
// them and keeps the rest under the MFMAs of the same wave:
//   * no per-score multiply-add: Q arrives multiplied by d_head^-0.5 * log2(e) (the to_q projection's fp32 epilogue did it:
//     ONE rounding, like any stored q), and -m travels through the matrix product -- the 48-deep walk of d_head 40 has 8
//     spare contraction slots: Q column 40 holds -m (bf16), K's pad column 40 holds 1.0 -- so the accumulators deliver
//     s - m and the softmax starts at v_exp_f32;
//   * no running maximum after the first tile ("optimistic" pass): m is the row maximum of tile 0, rounded to bf16.  The
//     maximum is subtracted for RANGE, not for precision -- exp2, the bf16 rounding of P and the fp32 accumulation of P V
//     and sum P are all relative -- so a later score may exceed m by up to ~2^100 before anything overflows.  The epilogue
//     checks every row's denominator (it comes out of the matrix pipe: V^T's spare row 40 is ones) and, if any row of the
//     workgroup left [1e-30, 1e30], the whole workgroup repeats the block with the conventional lazily-rescaled maximum
//     (same code, template flag): correct for any input, one extra pass for inputs whose scores spread over > 60 nats;
//   * per-wave software pipeline instead of barrier phases: iteration u issues P.V of tile u-1 and S^T = K Q^T of tile u+1
//     as its MFMAs and the exp2 / pack / swap of tile u as its VALU, 1-2 exp2 chunks behind every MFMA; S lives in ONE
//     register tile that the next tile's MFMAs overwrite half by half as the exp2 pass releases it (128 VGPRs: four waves per
//     SIMD, two workgroups per CU); operand fragments are requested just in time with counted lgkmcnt waits; one s_barrier
//     per tile; K / V tiles ride a 4-slot LDS-DMA ring.
// Layouts (K-row permutation, v_permlane16_swap hand-off of P to the 16x16x32 P.V product, transpose reads of V) are those
// of attn_fwd_hyb_kernel.  Measured on MI355X (B x H = 64, N = 4096): see DESIGN.md 3.2, round 4.
#include <type_traits>
#include "attn_common.h"
#include "attn_tr_util.h"

namespace cl {

namespace {

constexpr int F40_DH = 40;
using G40 = Geo<F40_DH>;
constexpr int F40_RING = 4, F40_AHEAD = 2;
constexpr int F40_STAGE = 2 * G40::TILE;
constexpr int F40_SLACK = 16 * G40::ROWB + 64;                   // zeros behind the ring: the last rows' fragment reads run past a tile
constexpr int F40_QFRAG = 3 * 1024;                              // per wave: its three Q operand fragments (64 lanes x 16 B each)
// NW = 8: 256 queries per workgroup, 128 registers (four waves per SIMD, two workgroups per CU), Q fragments in LDS;
// NW = 4: 128 queries per workgroup, 168 registers (three waves per SIMD, three workgroups per CU), Q fragments in registers
constexpr int f40_lds(int NW) { return F40_RING * F40_STAGE + F40_SLACK + (NW == 8 ? NW * F40_QFRAG : 0); }

// VALU chunks of one iteration: e0..e15 = exp2 + pack of score pairs (e_k: S half k / 8, register pair k % 8),
// w0 / w1 = the four lane swaps that turn a half's packed P into the two 16-query B operands of the P.V product.
// MFMA slots: 0..11 = P.V of the previous tile (three V^T groups of four), 12..17 = S^T of the next tile, contraction step
// j = (slot - 12) / 2 of half (slot - 12) % 2 -- the two halves alternate, so consecutive MFMAs never share an accumulator
// and one Q fragment serves two MFMAs back to back.  All sixteen e-chunks ride on the P.V slots: S^T of the next tile is
// formed IN PLACE (it overwrites the scores the e-chunks read), and the swaps, which overwrite the P operands the P.V
// MFMAs read, follow slot 11.  e_end(slot) = number of e-chunks issued once the slot's MFMA is out.
template <bool OPT, bool HAS_PV> struct F40Sched {
  static constexpr int pre() { return HAS_PV ? 0 : 16; }          // first iteration: no P.V to hide under
  static constexpr int e_end(int slot) {
    if (!HAS_PV || slot >= 12) return 16;
    if (OPT) return slot < 4 ? 2 * (slot + 1) : 8 + (slot - 3);
    return slot < 4 ? 0 : 2 * (slot - 3);                          // slots 0-3 carry the row maximum
  }
};

}  // namespace

template <int NW>
__global__ __launch_bounds__(64 * NW, (NW == 8 ? 4 : 3)) void attn_fwd40_kernel(AttnFwdArgs p, const void* __restrict__ V, long ldv, int nqb,
                                                                                int remap) {
  constexpr int F40_NW = NW, F40_THREADS = 64 * NW;
  constexpr bool QLDS = NW == 8;
  constexpr int NQR = QLDS ? 3 : 0;                    // LDS reads of a Q-fragment request
  constexpr int CPR = G40::CPR, CPRP = G40::CPRP, DN = G40::DN, ROWB = G40::ROWB, TILE = G40::TILE;
  constexpr int STAGE = F40_STAGE, RING = F40_RING, AHEAD = F40_AHEAD, QW = 2, NK = 3, DH = F40_DH;
  constexpr int LROW = DH % 16;                        // V^T row 40 = ones: O^T[40, q] = sum_k P[q, k]
  constexpr float RESCALE_THR = 6.0f;
  static_assert(CPRP == 6 && DN == 3, "d_head 40 geometry");
  constexpr int NJ = (CPRP + NW - 1) / NW;             // DMA instructions per wave, operand and tile
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
  const int q0 = qb * (32 * F40_NW) + wave * 32;

  // LDS: slack behind the ring (the last rows' fragment reads run past a tile) and the pad chunks: K pad = (1.0, 0, ...) -- column
  // 40 meets -m in Q --, V pad = ones -- rows 40..47 of V^T: the softmax denominator
  for (int i = tid; i < (16 * ROWB + 64) / 4; i += F40_THREADS) reinterpret_cast<uint32_t*>(smem + RING * STAGE)[i] = 0u;
  for (int i = tid; i < 2 * RING * 64; i += F40_THREADS) {
    const bool vtile = (i >> 6) & 1;
    const uint4 w = vtile ? make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u) : make_uint4(0x00003F80u, 0u, 0u, 0u);
    *reinterpret_cast<uint4*>(smem + (long)i * ROWB + CPR * 16) = w;
  }

  // Q as the B operand of the 32x32x16 product: col = query l31, k = 16 j + 8 hi .. +7; chunk 5 (columns 40..47) starts as zeros.
  // The three fragments live in LDS (lane-linear, this wave's own 3 KB) and are re-read for every tile: as 12 registers held
  // across the loop they pushed the kernel over its 128-register budget, and a spill reload inside the loop is a VMEM access
  // whose compiler-inserted vmcnt(0) also waits for the tile DMA just issued.
  char* const qfrag_p = smem + RING * STAGE + F40_SLACK + (QLDS ? wave * F40_QFRAG : 0) + lane * 16;
  const uint32_t qfrag = (uint32_t)(uintptr_t)qfrag_p;
  u32x4_t qh[QLDS ? 1 : NK];                           // (NW = 4: the fragments stay in registers)
  {
    const char* qp = (const char*)p.Q + (((long)b * p.N + q0 + l31) * p.ldq + (long)h * DH) * 2;
#pragma unroll
    for (int j = 0; j < NK; ++j) {
      const int c = 2 * j + hi;
      const u32x4_t q = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(qp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
      if constexpr (QLDS) *reinterpret_cast<u32x4_t*>(qfrag_p + j * 1024) = q;
      else qh[j] = q;
    }
  }
  // Q column 40 = -m (bf16 bits `mb` of m): word 0 of fragment 2 of the upper half-wave (columns 40, 41); read back by this lane only
  auto set_q_minus_m = [&](uint32_t mb) {
    const uint32_t w = mb ? (mb ^ 0x8000u) : 0u;
    if constexpr (QLDS) { if (hi) *reinterpret_cast<uint32_t*>(qfrag_p + (NK - 1) * 1024) = w; }
    else { if (hi) qh[NK - 1][0] = w; }
  };
  const char* kbase = (const char*)p.K + ((long)b * p.Nkv * p.ldk + (long)h * DH) * 2;
  const char* vbase = (const char*)V + ((long)b * p.Nkv * ldv + (long)h * DH) * 2;

  // tile DMA: wave w < 6 moves chunk-instruction w of K and of V (64 lanes x 16 B; lanes that land on a pad chunk are masked)
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
  // (a tile is requested two iterations before its first use: by then a plain vmcnt(0) costs nothing)
  auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int kr = (l31 & 19) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);             // K-row permutation: see attn_fwd_hyb_kernel
  const uint32_t krow = kr * ROWB + hi * 16;
  const uint32_t troff = (4 * g + ((lane >> 2) & 3)) * ROWB + (lane & 3) * 8;
  const int nt = p.Nkv / 64;

  uint32_t k_addr, v_addr;                     // fragment read addresses of the iteration in hand (see body)
  f32x4_t ot[DN][QW];
  float m_run;                                 // of query l31 (both half-waves keep the same value), bf16-representable
  f32x16_t sc[2];                              // S^T - m of the tile in hand: keys 32 s + (r & 3) + 8 (r >> 2) + 4 hi, query l31
  u32x4_t pb[2][QW];                           // P^T operands of the tile whose P.V is pending

  // ---- one iteration: softmax of tile u (scores in sc), P.V of tile u-1, S^T of tile u+1 (into sc, in place)
  auto body = [&](auto OPTc, auto PVc, auto QKc, int u) {
    constexpr bool OPT = decltype(OPTc)::value, HAS_PV = decltype(PVc)::value, HAS_QK = decltype(QKc)::value;
    using SCH = F40Sched<OPT, HAS_PV>;
    constexpr bool FIRST = !HAS_PV;
    constexpr bool TRACK = FIRST || !OPT;          // this iteration looks at the row maximum
    dma_wait();                                               // tile u + 1 (requested in iteration u - 1) has landed
    __builtin_amdgcn_s_barrier();
    if (u + AHEAD < nt) issue(u + AHEAD, (u + AHEAD) & (RING - 1));
    // Operand addresses: ONE loop-carried register per operand, advanced by a wave-uniform step at the end of the iteration,
    // + instruction immediates.  (Lane-constant bases recombined with the ring slot every iteration were hoisted out of the
    // loop and spilled: a spill reload is a VMEM access whose vmcnt(0) also waits for the tile DMA just issued.)
    //   k_addr -> K(u+1) in slot (u + 1) % 4,  v_addr -> V(u-1) in slot (u - 1) % 4
    u32x4_t va[2][2];                              // V^T fragments: group J in va[J % 2]
    u32x4_t kf[2][2], qf[2];                       // S^T step j: K fragments of both halves and the Q fragment in buffer j % 2
    auto req_v = [&](auto Jc) {
      constexpr int J = decltype(Jc)::value;
      va[J % 2][0] = tr_frag_off<ROWB, 0, J * 32>(v_addr);
      va[J % 2][1] = tr_frag_off<ROWB, 1, J * 32>(v_addr);
    };
    constexpr int NRS = 2 + (QLDS ? 1 : 0);        // LDS reads of one S^T step request
    auto req_s = [&](auto Jc) {
      constexpr int j = decltype(Jc)::value;
      kf[j % 2][0] = lds_read_b128_off<j * 32>(k_addr);
      kf[j % 2][1] = lds_read_b128_off<32 * ROWB + j * 32>(k_addr);
      if constexpr (QLDS) qf[j % 2] = lds_read_b128_off<j * 1024>(qfrag);
      else qf[j % 2] = qh[j];
    };
    if constexpr (HAS_PV) { req_v(std::integral_constant<int, 0>{}); req_v(std::integral_constant<int, 1>{}); }

    uint32_t pk[2][8];
    auto echunk = [&](auto Cc) {
      constexpr int c = decltype(Cc)::value, s_ = c / 8, k = c % 8;
      const float e0 = __builtin_amdgcn_exp2f(sc[s_][2 * k]), e1 = __builtin_amdgcn_exp2f(sc[s_][2 * k + 1]);
      pk[s_][k] = pack2bf(e0, e1);
    };
    auto wchunk = [&](auto Sc) {
      constexpr int s_ = decltype(Sc)::value;
      // (P0,P2) (P1,P3) (P4,P6) (P5,P7): rows 1 / 3 of the first <-> rows 0 / 2 of the second
      const auto s0 = __builtin_amdgcn_permlane16_swap(pk[s_][0], pk[s_][2], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(pk[s_][1], pk[s_][3], false, false);
      const auto s2 = __builtin_amdgcn_permlane16_swap(pk[s_][4], pk[s_][6], false, false);
      const auto s3 = __builtin_amdgcn_permlane16_swap(pk[s_][5], pk[s_][7], false, false);
      pb[s_][0] = u32x4_t{s0[0], s1[0], s2[0], s3[0]};
      pb[s_][1] = u32x4_t{s0[1], s1[1], s2[1], s3[1]};
    };
    auto valu_after = [&](auto Sc) {               // the e-chunks that follow MFMA slot `slot`
      constexpr int slot = decltype(Sc)::value;
      constexpr int lo = slot == 0 ? SCH::pre() : SCH::e_end(slot - 1), hi_ = SCH::e_end(slot);
      static_for<lo, (hi_ > lo ? hi_ : lo)>([&](auto Cc) { echunk(Cc); });
    };

    // ---- row maximum (first tile; every tile in the conventional pass), beside the first P.V group where there is one
    float mx = 0.f;
    auto max_slice = [&](auto Ic) {
      constexpr int i = decltype(Ic)::value;                  // 4 slices of 8 scores
      if constexpr (i == 0) mx = sc[0][0];
#pragma unroll
      for (int r = 0; r < 8; ++r) mx = fmaxf(mx, sc[i >> 1][(i & 1) * 8 + r]);
    };
    auto pv_mfma = [&](auto Ic) {
      constexpr int i = decltype(Ic)::value, J = i / 4, q = i % 4;
      Mma<bf16_t>::run(va[J % 2][q >> 1], pb[q >> 1][q & 1], ot[J][q & 1]);
    };
    if constexpr (HAS_PV) {
      lgkm_wait<4>();                                         // group 0 landed, group 1 in flight
      pin(va[0][0]); pin(va[0][1]);
      static_for<0, 4>([&](auto Ic) {
        pv_mfma(Ic);
        if constexpr (TRACK) max_slice(Ic);
        else valu_after(Ic);
        __builtin_amdgcn_sched_barrier(0);
      });
      req_v(std::integral_constant<int, 2>{});                // into group 0's registers (its MFMAs are issued)
    } else {
      static_for<0, 4>([&](auto Ic) { max_slice(Ic); });
    }
    // ---- maximum decision (wave-uniform branch).  FIRST: S(0) was formed against m = 0: m = bf16(row maximum), always.
    float alpha = 1.0f;
    bool resc = false;
    if constexpr (TRACK) {
      resc = FIRST ? true : __any(mx > RESCALE_THR);          // sc holds s - m_run
      if (resc) {
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));               // the other half-wave holds the other keys of this query
        const float m_new = FIRST ? mx : fmaxf(m_run, m_run + mx);
        const uint32_t mb = pack2bf(m_new, 0.f) & 0xffffu;    // bf16, round to nearest even
        const float m_b = __uint_as_float(mb << 16);
        const float d = m_b - m_run;
        alpha = FIRST ? 1.0f : __builtin_amdgcn_exp2f(-d);
        m_run = m_b;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[s_][r] -= d;        // the tile in hand was formed against the old m
        set_q_minus_m(mb);                                    // Q column 40 = -m for the S tiles to come
      }
    }
    static_for<0, SCH::pre()>([&](auto Cc) { echunk(Cc); });
    // ---- the remaining P.V groups
    if constexpr (HAS_PV) {
      static_for<4, 12>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value, J = i / 4, q = i % 4;
        if constexpr (q == 0) {
          if constexpr (J == 1) lgkm_wait<4>();               // group 1 landed, group 2 in flight
          else {
            if constexpr (HAS_QK) req_s(std::integral_constant<int, 0>{});
            lgkm_wait<(HAS_QK ? NRS : 0)>();                  // group 2 landed, step 0 of S^T in flight
          }
          pin(va[J % 2][0]); pin(va[J % 2][1]);
        }
        pv_mfma(Ic);
        valu_after(Ic);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else if constexpr (HAS_QK) {
      req_s(std::integral_constant<int, 0>{});                // (after the decision: column 40 of the Q fragment just changed)
    }
    wchunk(std::integral_constant<int, 0>{});                 // every P.V MFMA that read the old P operands is out
    __builtin_amdgcn_sched_barrier(0);
    // ---- S^T of tile u + 1, in place, halves alternating; the second swap chunk rides on the first MFMA
    if constexpr (HAS_QK) {
      static_for<0, NK>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        if constexpr (j + 1 < NK) req_s(std::integral_constant<int, j + 1>{});
        lgkm_wait<(j + 1 < NK ? NRS : 0)>();
        pin(kf[j % 2][0]); pin(kf[j % 2][1]); pin(qf[j % 2]);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          if constexpr (j == 0) sc[s_] = f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          sc[s_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, kf[j % 2][s_]), __builtin_bit_cast(bf16x8_t, qf[j % 2]), sc[s_], 0, 0, 0);
          if constexpr (j == 0) {
            if (s_ == 0) wchunk(std::integral_constant<int, 1>{});
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    } else {
      wchunk(std::integral_constant<int, 1>{});
    }
    k_addr += ((u + 2) & (RING - 1)) ? STAGE : -(RING - 1) * STAGE;
    v_addr += (u & (RING - 1)) ? STAGE : -(RING - 1) * STAGE;
    if constexpr (TRACK && HAS_PV) {
      if (resc) {                                             // after the iteration's last P.V MFMA: O(u-1) -> alpha O(u-1)
#pragma unroll
        for (int f = 0; f < QW; ++f) {
          const float af = __shfl(alpha, 16 * f + lq, 64);
#pragma unroll
          for (int i = 0; i < DN; ++i) ot[i][f] *= af;
        }
      }
    }
  };

  // ---- one pass over the keys
  auto run = [&](auto OPTc) {
#pragma unroll
    for (int i = 0; i < DN; ++i)
#pragma unroll
      for (int f = 0; f < QW; ++f) ot[i][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    m_run = 0.f;
    set_q_minus_m(0u);
    issue(0, 0);
    issue(1, 1);                              // (nt >= 2: launcher)
    dma_wait();
    __syncthreads();
    k_addr = lds0 + krow;                                     // K(0), slot 0
    v_addr = lds0 + (RING - 1) * STAGE + TILE + troff;        // "V(-1)", slot 3: first used (as V(0), slot 0) in iteration 1
    {   // S(0), formed against m = 0 and corrected in the first iteration
      const uint32_t k_addr0 = k_addr;
      u32x4_t ka[2][NK], qa[NK];
      static_for<0, NK>([&](auto Kc) {
        constexpr int j = decltype(Kc)::value;
        if constexpr (QLDS) qa[j] = lds_read_b128_off<j * 1024>(qfrag);
        else qa[j] = qh[j];
      });
      static_for<0, 2>([&](auto Sc) {
        constexpr int s_ = decltype(Sc)::value;
        static_for<0, NK>([&](auto Kc) {
          constexpr int j = decltype(Kc)::value;
          ka[s_][j] = lds_read_b128_off<s_ * 32 * ROWB + j * 32>(k_addr0);
        });
      });
      lgkm_wait<0>();
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
        for (int j = 0; j < NK; ++j) { pin(ka[s_][j]); pin(qa[j]); }
        f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NK; ++j)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ka[s_][j]), __builtin_bit_cast(bf16x8_t, qa[j]), acc, 0, 0, 0);
        sc[s_] = acc;
      }
    }
    k_addr += STAGE;                                          // K(1), slot 1
    body(OPTc, std::false_type{}, std::true_type{}, 0);
    for (int u = 1; u + 1 < nt; ++u) body(OPTc, std::true_type{}, std::true_type{}, u);
    body(OPTc, std::true_type{}, std::false_type{}, nt - 1);
    {   // P.V of the last tile
      static_for<0, DN>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        u32x4_t va[2];
        va[0] = tr_frag_off<ROWB, 0, i * 32>(v_addr); va[1] = tr_frag_off<ROWB, 1, i * 32>(v_addr);
        lds_wait();
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
          for (int f = 0; f < QW; ++f) Mma<bf16_t>::run(va[s_], pb[s_][f], ot[i][f]);
      });
    }
  };

  run(std::true_type{});
  // every row's denominator (matrix-pipe sum of the bf16 P, O^T row 40) must be an ordinary number; otherwise the whole
  // workgroup (the K / V ring is collective) repeats the block with the conventional running maximum
  float lsum[QW];
  auto denominators = [&]() {
    bool bad = false;
#pragma unroll
    for (int f = 0; f < QW; ++f) {
      lsum[f] = __shfl(ot[DN - 1][f][LROW & 3], lq + 16 * (LROW >> 2), 64);
      bad |= !(lsum[f] > 1e-30f && lsum[f] < 1e30f);
    }
    return bad;
  };
  if (__syncthreads_or(denominators() ? 1 : 0)) {
    run(std::false_type{});
    denominators();
  }

  // ---- epilogue: normalise, store O rows, log-sum-exp (log2 domain)
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    const float m = __shfl(m_run, 16 * f + lq, 64);
    const float inv = 1.0f / lsum[f];
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
    if (p.LSE && g == 0) p.LSE[((long)b * p.H + h) * p.lse_stride + row] = m + __builtin_amdgcn_logf(lsum[f]);
  }
}

// The pre-scaled-Q forward applies when: Q is pre-scaled, d_head 40, whole 256-query blocks, whole 64-key tiles (>= 2),
// at least a chip's worth of workgroups.  Returns false when it does not (the caller falls through to the other kernels).
bool attn_fwd40_applies(const AttnFwdArgs& a) {
  const long grid = (long)(a.N / 256) * a.H * a.B;
  return a.q_prescaled && a.DH == F40_DH && a.N % 256 == 0 && a.Nkv % 64 == 0 && a.Nkv >= 128 && grid >= 256;
}

int g_attn_fwd40_waves = 8;        // probe hook: 8 (default) or 4 waves per workgroup

template <int NW>
static int launch_fwd40(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  constexpr int LDS = f40_lds(NW);
  static bool done = false;
  if (!done) {
    if (LDS > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd40_kernel<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return CL_ELAUNCH;
    done = true;
  }
  const int nqb = a.N / (32 * NW);
  const long grid = (long)nqb * a.H * a.B;
  const int remap = ((a.B * a.H) % 8 == 0) ? 1 : 0;
  hipLaunchKernelGGL(attn_fwd40_kernel<NW>, dim3((unsigned)grid), dim3(64 * NW), LDS, st, a, V, ldv, nqb, remap);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

int attn_fwd40(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st) {
  if (g_attn_fwd40_waves == 1) return attn_fwd40x(a, V, ldv, st);
  return g_attn_fwd40_waves == 4 ? launch_fwd40<4>(a, V, ldv, st) : launch_fwd40<8>(a, V, ldv, st);
}

}  // namespace cl
