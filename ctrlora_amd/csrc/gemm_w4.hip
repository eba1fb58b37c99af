// Loader / consumer tile kernel for the MFMA-bound contractions (bf16; launch configurations 40 / 41 of gemm.hip).
//
//   out[M,N] = epilogue( A1[M,K1].W1[N,K1]^T  (+ A2[M,K2].W2[N,K2]^T) )          linear, or implicit 3x3 conv over NHWC
//
// Which products: the ResBlock / Downsample / Upsample 3x3 convolutions (ldm/modules/diffusionmodules/openaimodel.py:108-118,
// 150,203,229) and the deep-K linears (attention.py:59-76 FeedForward out, 163-170 projections at K >= 1280) -- everything the
// 256 x 160 ping-pong tile kernel (gemm_fl_kernel, gemm.hip) served at 42-44 % of the matrix peak.  What limited that kernel was
// its synchronisation skeleton: eight waves, four s_barrier-delimited sections per 128-byte stage (two waves of a SIMD swap
// the "load" and "multiply" roles at every barrier), 20-MFMA sections of v_mfma_f32_16x16x32_bf16, 9 ds_read_b128 per 20 MFMAs.
//
// Structure here: the two roles are two KINDS of wave, for the whole tile.
//   * waves 0-3, the CONSUMERS, one per SIMD: each owns 64 rows x BN columns of the 256 x BN tile and issues nothing but
//     v_mfma_f32_32x32x16_bf16 (inline asm, accumulators in the accumulation registers) and the ds_read_b128 of the next k-step's
//     fragments, two reads per MFMA gap: 2 + BN/32 reads per 2 BN/32 MFMAs of 32 cycles each (BN 160: 7 reads per 320 matrix
//     cycles, 35 % of the LDS read bandwidth; the 16x16x32 form needs 9 per 320).  The product is formed TRANSPOSED
//     (D^T[n x m] = W[n x k] . X^T[k x m]: W is the A operand), so a lane ends up with 4 consecutive output columns of ONE row
//     per 8-column group; v_permlane32_swap pairs the half-waves' groups and every lane stores 8 consecutive columns (16 / 32
//     bytes) straight from registers -- no LDS staging, no barrier in the epilogue;
//   * waves 4-7, the LOADERS, one per SIMD: all address generation (linear / stride-1 conv with a 9-bit tap mask / generic
//     conv) and every global_load_lds_dwordx4 (8 rows x 128 B per instruction, XOR-swizzled on the source side exactly as in
//     gemm_fl_kernel: the 32-row fragment reads are bank-conflict-free).  A loader's LDS-DMA issue (tens of cycles each) and its
//     VALU run beside the consumer's MFMA stream on the same SIMD instead of inside it;
//   * ONE s_barrier per stage.  Ring of 3 stages (156 KB).  B(s) = "stage s has landed and nobody reads stage s-2 any more":
//       loader   ... DMA(s+1) | vmcnt: own DMA(s) landed | B(s) | DMA(s+2) -> slot of stage s-1 | vmcnt | B(s+1) ...
//       consumer ... k-steps 0..2 of stage s-1 | lgkmcnt(0): last reads of stage s-1 retired | B(s) | k-step 3 of stage s-1 with
//                    the reads of stage s, k-step 0 | k-steps 0..2 of stage s | B(s+1) ...
//     RAW: every loader's counted vmcnt for stage s precedes B(s); the first read of stage s is issued after it.
//     WAR: the slot of stage s-1 is refilled (DMA(s+2)) after B(s), which every consumer passes with lgkmcnt(0) after its last
//          read of stage s-1 (issued in k-step 2 of that stage).
//     A loader sits in B(s+1) about one stage ahead of the consumers: a DMA has two stages (~2500 cycles) to land.
// All waves of a workgroup share one register allocation: 512 threads = two waves per SIMD = 256 registers, of which the
// consumer holds 160 accumulators + 56 fragment registers + 8 addresses, the loader ~50 addresses.
#include <type_traits>
#include "gemm.h"
#include "gemm_epi.h"

namespace cl {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int I, int N, typename F> __device__ __forceinline__ void w4_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); w4_for<I + 1, N>(f); }
}
template <int OFF> __device__ __forceinline__ void w4_rd128(u32x4_t& v, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
// c += a . b; every operand placement is explicit: fragments "v", accumulators "a" -- except the fifth 32-column block's, which
// live in architectural registers ("v"): with accumulation registers in use hipcc (ROCm 7.2) splits a 256-register budget
// 128 | 128, so 160 accumulators cannot all be "a" (268 spills), while 128 "a" + 32 "v" + 56 fragment registers fit.
// (The compiler pads nothing around an asm MFMA: the stream below never reads an accumulator, and the epilogue drains first.)
template <bool AG> __device__ __forceinline__ void w4_mfma(f32x16_t& c, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ void w4_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void w4_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// wait states around the asm MFMAs that the compiler does not know about: after the v_accvgpr_write of the zero fill (SHORT),
// and 16 passes after the last MFMA before anything else reads its result
template <bool SHORT, bool AG> __device__ __forceinline__ void w4_pad(f32x16_t& a, f32x16_t& b) {
  if constexpr (SHORT) { if constexpr (AG) asm volatile("s_nop 7" : "+a"(a), "+a"(b)); else asm volatile("s_nop 7" : "+v"(a), "+v"(b)); }
  else { if constexpr (AG) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(a), "+a"(b)); else asm volatile("s_nop 15\n\ts_nop 15" : "+v"(a), "+v"(b)); }
}

enum { W4_LINEAR = 0, W4_CONV_S1 = 1, W4_CONV_ANY = 2 };

// Issue plans of the fragment reads: the gap (= behind MFMA g) that read position ps goes out in, per k-step kind.
//   kind 0 (k-steps 0, 1)  W0 X0 | X1 | W1 | - | W2 | - | W3 ...      kind 1 (k-step 2)  two per gap from gap 0
//   kind 2 (k-step 3)      kind 0 pushed back by one gap              kinds 3 / 4 (SCHED 0)  RPG per gap from gap 1 / 0
template <int NF, int RPG = 2> constexpr int w4_gap_of(int kind, int ps) {
  if (kind == 1) return ps / 2;
  if (kind == 3) return 1 + ps / RPG;
  if (kind == 4) return ps / RPG;
  const int g = ps <= 1 ? 0 : (ps == 2 ? 1 : 2 * (ps - 2));
  return kind == 2 ? (g == 0 ? 1 : (g < 4 ? g + 1 : g)) : g;
}
// reads of plan `kind` that are issued before MFMA m (i.e. in gaps < m)
template <int NF> constexpr int w4_issued_before(int kind, int m) {
  int n = 0;
  for (int ps = 0; ps < 2 + NF; ++ps) n += w4_gap_of<NF>(kind, ps) < m ? 1 : 0;
  return n;
}

// ABL (template parameter; non-zero instances exist in probe builds only, tools/probe_gemm.hip -DW4_PROBE): bit 0 = consumers skip
// their fragment reads, bit 1 = loaders skip their DMA, bit 2 = no stores -- wrong results by construction, they price the ingredients
#define W4_ABL(bit) ((ABL & (bit)) != 0)
#ifdef W4_PROBE
int g_w4_abl_host = 0, g_w4_sched_host = 1;
// wave 0 of every workgroup stores s_memtime at four points of its tile: entry | stage 0 landed | main loop done | stores retired
__device__ unsigned long long* g_w4_timing = nullptr;
#define W4_STAMP(i) do { if (g_w4_timing && threadIdx.x == 0) g_w4_timing[(long)blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4_STAMP(i) do { } while (0)
#endif

constexpr int W4_BM = 256, W4_NC = 4, W4_NL = 4, W4_R = 3;

// NF: 32-column fragment blocks per tile row (BN = 32 NF: 5 -> 160, 4 -> 128).  RPG: fragment reads per MFMA gap.
template <int NF, int MODE, int SCHED, int ABL = 0>
__global__ __launch_bounds__(64 * (W4_NC + W4_NL)) void gemm_w4_kernel(GemmParams p, int tiles_m, int tiles_n,
                                                                      float* __restrict__ slab) {
  typedef bf16_t T;
  constexpr int BN = 32 * NF;
  constexpr int XI = W4_BM / 8, WI = BN / 8;            // DMA instructions (8 rows x 128 B) per stage: X rows, W rows
  constexpr int XJ = XI / W4_NL, WJ = WI / W4_NL;       // ... per loader wave
  constexpr int G = XJ + WJ;
  constexpr int GA = W4_ABL(2) ? 0 : (W4_ABL(8) ? 0 : XJ) + (W4_ABL(16) ? 0 : WJ);   // (G in every product build)
  constexpr int SLOT = (XI + WI) * 1024;
  constexpr int RPG = 2;
  constexpr int NFR = 2 + NF;                           // fragments per k-step and consumer: X0 X1 W0 .. W(NF-1)
  static_assert(WI % W4_NL == 0, "every loader issues the same number of DMA instructions (counted vmcnt)");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- XCD-aware tile id (as gemm_fl_kernel): XCD x (= workgroup id mod 8) owns a contiguous range of tiles
  const int nt = tiles_m * tiles_n;
  if ((int)blockIdx.x >= nt * max(p.splitk, 1)) return;     // launch-tag workgroups (debug_hooks.h)
  int pid = blockIdx.x;
  const int zsplit = pid / nt;
  pid -= zsplit * nt;
  {
    const int q = nt >> 3, r = nt & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (pid / tiles_n) * W4_BM, n0 = (pid % tiles_n) * BN;

  const int cpt = p.K1 / 64;  // stages per tap
  const int ks1 = (MODE == W4_LINEAR ? 1 : 9) * cpt;
  const int ks2 = (MODE == W4_LINEAR) ? p.K2 / 64 : 0;
  int kbeg = 0, kend = ks1 + ks2;
  if (p.splitk > 1) {
    const int per = (kend + p.splitk - 1) / p.splitk;
    kbeg = zsplit * per;
    kend = min(kend, kbeg + per);
  }
  const int total = kend - kbeg;                            // >= 1 (launcher)

  if (wave >= W4_NC) {
    // =================================================================================== loader
    const int lw = wave - W4_NC;
    const int lrow = lane >> 3, lslot = lane & 7;
    const char* pa[XJ];            // LINEAR: running pointer.  CONV: centre pixel (S1) / base (ANY)
    const char* a2[XJ];
    uint32_t vmask[XJ];            // CONV_S1: bit t = tap t reads inside the image
    int ab[XJ], ay[XJ], ax[XJ];    // CONV_ANY: output pixel coordinates
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
      const int inst = j * W4_NL + lw;
      const int chunk = (lslot ^ (((inst & 1) << 2) | (lrow >> 1))) * 16;
      int r = m0 + inst * 8 + lrow;
      r = min(r, p.M - 1);
      a2[j] = nullptr; vmask[j] = 0; ab[j] = ay[j] = ax[j] = 0;
      if constexpr (MODE == W4_LINEAR) {
        const bool in2 = kbeg >= ks1;
        a2[j] = p.A2 ? (const char*)p.A2 + ((long)r * p.lda2 + (p.a2_group_n ? (long)(n0 / p.a2_group_n) * p.K2 : 0)) * sizeof(T) + chunk
                     : nullptr;
        pa[j] = in2 ? a2[j] + (long)(kbeg - ks1) * 128
                    : (const char*)p.A1 + ((long)r * p.lda1 + (p.a1_group_n ? (long)(n0 / p.a1_group_n) * p.K1 : 0)) * sizeof(T) +
                          chunk + (long)kbeg * 128;
      } else {
        const int ox = r % p.Wout; const int t = r / p.Wout;
        const int oy = t % p.Hout, ob = t / p.Hout;
        if constexpr (MODE == W4_CONV_S1) {
          pa[j] = (const char*)p.A1 + ((((long)ob * p.Hin + oy) * p.Win + ox) * p.lda1) * sizeof(T) + chunk;
          uint32_t m = 0;
#pragma unroll
          for (int tp = 0; tp < 9; ++tp) {
            const int vy = oy + tp / 3 - 1, vx = ox + tp % 3 - 1;
            if (((unsigned)vy < (unsigned)p.Hin) & ((unsigned)vx < (unsigned)p.Win)) m |= 1u << tp;
          }
          vmask[j] = m;
        } else {
          pa[j] = (const char*)p.A1 + chunk;
          ax[j] = ox; ay[j] = oy; ab[j] = ob;
        }
      }
    }
    const char* pw[WJ]; const char* w2[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int inst = j * W4_NL + lw;
      const int chunk = (lslot ^ (((inst & 1) << 2) | (lrow >> 1))) * 16;
      int n = n0 + inst * 8 + lrow;
      n = min(n, p.N - 1);
      w2[j] = (MODE == W4_LINEAR && p.W2) ? (const char*)p.W2 + ((long)n * p.ldw2) * sizeof(T) + chunk : nullptr;
      pw[j] = (MODE == W4_LINEAR && kbeg >= ks1)
                  ? w2[j] + (long)(kbeg - ks1) * 128
                  : (const char*)p.W1 + ((long)n * p.ldw1) * sizeof(T) + chunk + (long)kbeg * 128;
    }
    const char* zpage = (const char*)p.zero_page + lslot * 16;

    // wave-uniform walk over (tap, channel chunk) for the conv modes
    int kt_next = kbeg;
    int tap = (MODE == W4_LINEAR) ? 0 : kbeg / cpt;
    int cc = (MODE == W4_LINEAR) ? 0 : kbeg - tap * cpt;
    const long pixb = (long)p.lda1 * sizeof(T);
    long tapoff = 0;
    const char* cur[XJ];
#pragma unroll
    for (int j = 0; j < XJ; ++j) cur[j] = nullptr;
    if constexpr (MODE == W4_CONV_S1) {
      tapoff = ((long)(tap / 3 - 1) * p.Win + (tap % 3 - 1)) * pixb;
#pragma unroll
      for (int j = 0; j < XJ; ++j)   // a split-K workgroup may start in the middle of a tap
        cur[j] = ((vmask[j] >> tap) & 1u) ? pa[j] + tapoff + (long)cc * 128 : zpage + (long)cc * 128;
    }

    auto issue = [&](int slot) {
      char* Xs = smem + slot * SLOT;
      char* Ws = Xs + XI * 1024;
      if constexpr (W4_ABL(2)) { ++kt_next; return; }
      if constexpr (MODE == W4_LINEAR) {
        if (ks2 && kt_next == ks1) {   // switch to the second K segment (LoRA up-projection)
#pragma unroll
          for (int j = 0; j < XJ; ++j) pa[j] = a2[j];
#pragma unroll
          for (int j = 0; j < WJ; ++j) pw[j] = w2[j];
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) { if constexpr (!W4_ABL(8)) glds16(pa[j], Xs + (j * W4_NL + lw) * 1024); pa[j] += 128; }
      } else if constexpr (MODE == W4_CONV_S1) {
        // cur[j] walks the channel chunks of the current tap (+128 B per stage); lanes whose tap falls outside the image walk
        // the zero page instead (at least (K1 / 64 + 1) * 128 bytes long); the select against the 9-bit mask happens only when
        // the tap changes
        if (cc == 0) {
#pragma unroll
          for (int j = 0; j < XJ; ++j) cur[j] = ((vmask[j] >> tap) & 1u) ? pa[j] + tapoff : zpage;
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) { if constexpr (!W4_ABL(8)) glds16(cur[j], Xs + (j * W4_NL + lw) * 1024); cur[j] += 128; }
        const bool wrap = cc + 1 == cpt;
        cc = wrap ? 0 : cc + 1;
        tap += wrap ? 1 : 0;
        const int ky = (tap * 11) >> 5;           // tap / 3 for tap in [0, 9]
        tapoff = ((long)(ky - 1) * p.Win + (tap - 3 * ky - 1)) * pixb;
      } else {
        const int ky = tap / 3, kx = tap - ky * 3;
        const int sy = (p.mode == GEMM_CONV_S2 || p.mode == GEMM_CONV_S2A) ? 2 : 1;
        const int po = (p.mode == GEMM_CONV_S2A) ? 0 : 1;     // left / top padding
        const bool virt = (p.mode == GEMM_CONV_UP2) | (p.mode == GEMM_CONV_T2);
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
          const int vy = ay[j] * sy + ky - po, vx = ax[j] * sy + kx - po;
          bool ok; int iy, ix;
          if (virt) {
            ok = ((unsigned)vy < (unsigned)(2 * p.Hin)) & ((unsigned)vx < (unsigned)(2 * p.Win));
            if (p.mode == GEMM_CONV_T2) ok = ok & !((vy | vx) & 1);
            iy = vy >> 1; ix = vx >> 1;
          } else {
            ok = ((unsigned)vy < (unsigned)p.Hin) & ((unsigned)vx < (unsigned)p.Win);
            iy = vy; ix = vx;
          }
          const long pix = ((long)ab[j] * p.Hin + iy) * p.Win + ix;
          const char* src = ok ? pa[j] + pix * pixb + (long)cc * 128 : zpage;
          glds16(src, Xs + (j * W4_NL + lw) * 1024);
        }
        if (++cc == cpt) { cc = 0; ++tap; }
      }
#pragma unroll
      for (int j = 0; j < WJ; ++j) { if constexpr (!W4_ABL(16)) glds16(pw[j], Ws + (j * W4_NL + lw) * 1024); pw[j] += 128; }
      ++kt_next;
    };

    issue(0);
    if (total > 1) { issue(1); w4_vm<GA>(); } else { w4_vm<0>(); }
    __builtin_amdgcn_s_barrier();                             // B(0)
    __builtin_amdgcn_sched_barrier(0);
    int slot2 = 2;
    for (int s = 0; s < total; ++s) {                         // (B(total) only balances the consumers' uniform stage body)
      if (s + 2 < total) { issue(slot2); w4_vm<GA>(); } else { w4_vm<0>(); }
      __builtin_amdgcn_s_barrier();                           // B(s+1)
      __builtin_amdgcn_sched_barrier(0);
      slot2 = (slot2 == W4_R - 1) ? 0 : slot2 + 1;
    }
    return;
  }

  // ===================================================================================== consumer
  const int l31 = lane & 31, hi = lane >> 5;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // fragment (rows 32 b + l31 of a tile, k-step j of the stage: logical 16-byte chunk 2 j + hi) -> slot (2 j) ^ (hi ^ swz(l31))
  uint32_t xa[4], wa[4];
  {
    const int t = hi ^ ((l31 >> 1) & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xa[j] = lds0 + (wave * 64 + l31) * 128 + (((2 * j) ^ t) * 16);
      wa[j] = lds0 + XI * 1024 + l31 * 128 + (((2 * j) ^ t) * 16);
    }
  }
  f32x16_t acc[NF][2];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][i][r] = 0.f;
  u32x4_t F[2][NFR];

  // Fragment reads are issued in the order the MFMAs need them -- position 0 1 2 3 4 .. = W0 X0 X1 W1 W2 .. (MFMA m multiplies
  // W[m / 2] by X[m % 2]) -- and LDS returns in order, so "fragment at position q has landed" is lgkmcnt(reads issued after it).
  auto read_pos = [&](auto Bc, auto Jc, auto Pc) {
    constexpr int b = decltype(Bc)::value, j = decltype(Jc)::value, ps = decltype(Pc)::value;
    constexpr int q = ps == 0 ? 2 : (ps <= 2 ? ps - 1 : ps);         // index into F[b]: X0 X1 W0 W1 ..
    if constexpr (W4_ABL(1)) return;
    else if constexpr (q < 2) w4_rd128<q * 4096>(F[b][q], xa[j]);
    else w4_rd128<(q - 2) * 4096>(F[b][q], wa[j]);
  };

  W4_STAMP(0);
  __builtin_amdgcn_s_barrier();                               // B(0): stage 0 has landed
  __builtin_amdgcn_sched_barrier(0);
  W4_STAMP(1);
  w4_for<0, NFR>([&](auto Pc) { read_pos(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, Pc); });
  // (the accumulators were just written by v_accvgpr_write: wait states before the first MFMA reads them as SrcC)
  w4_for<0, NF>([&](auto Fc) { constexpr int f = decltype(Fc)::value; w4_pad<true, (f < 4)>(acc[f][0], acc[f][1]); });
  w4_lgkm0();
  __builtin_amdgcn_sched_barrier(0);

  int slot = 0;
  // One stage = four k-steps of 2 NF MFMAs.  k-step j multiplies the fragments in F[j & 1] while the reads of the next k-step fill
  // F[(j + 1) & 1]; k-step 3 opens with B(s+1) and then reads k-step 0 of stage s+1.  The body is the same for EVERY stage: the last
  // one passes a balancing barrier and reads a stale slot into registers nobody uses.
  // Issue plans (gap g = behind MFMA g) per k-step kind, SCHED 1:
  //   j = 0, 1   W0 X0 | X1 | W1 | - | W2 | - | W3 | - | W4     one k-step (~320 cycles) between a read and its MFMA
  //   j = 2      W0 X0 | X1 W1 | W2 W3 | W4                      early: they must have retired at B(s+1) (WAR on the slot)
  //   j = 3      - | W0 X0 | X1 | W1 | W2 | - | W3 | - | W4      gap 0 moves the eight read addresses to the next slot
  // and the wait in front of MFMA m allows (reads issued behind the fragment it needs) outstanding.  SCHED 0: RPG reads per gap
  // from gap 0 (1 in k-step 3) and lgkmcnt(0) at the end of every k-step.
  auto stage = [&]() {
    w4_for<0, 4>([&](auto Jc) {
      constexpr int j = decltype(Jc)::value, cb = j & 1, nb = cb ^ 1, jn = (j + 1) & 3;
      constexpr int KIND = (SCHED == 0) ? (j == 3 ? 3 : 4) : (j < 2 ? 0 : (j == 2 ? 1 : 2));       // plan of THIS k-step's reads
      if constexpr (j == 3) {
        if constexpr (SCHED != 0) w4_lgkm0();                 // this stage's last reads (issued early in k-step 2) have retired
        __builtin_amdgcn_s_barrier();                         // B(s+1)
        __builtin_amdgcn_sched_barrier(0);
      }
      w4_for<0, 2 * NF>([&](auto Mc) {
        constexpr int m = decltype(Mc)::value;
        if constexpr (SCHED != 0 && j != 3 && (m < 2 || (m & 1) == 0)) {
          // fragments at positions <= need(m) have landed when at most the reads behind them are outstanding
          constexpr int need = m == 0 ? 1 : (m == 1 ? 2 : (m >> 1) + 2);
          constexpr int allow = (NFR - 1 - need) + w4_issued_before<NF>(KIND, m);
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(allow) : "memory");
        }
        if constexpr (!W4_ABL(32)) w4_mfma<((m >> 1) < 4)>(acc[m >> 1][m & 1], F[cb][2 + (m >> 1)], F[cb][m & 1]);
        if constexpr (j == 3 && m == 0) {
          __builtin_amdgcn_sched_barrier(0);
          const int d = (slot == W4_R - 1) ? -(W4_R - 1) * SLOT : SLOT;
#pragma unroll
          for (int k = 0; k < 4; ++k) { xa[k] += d; wa[k] += d; }
          slot = (slot == W4_R - 1) ? 0 : slot + 1;
          __builtin_amdgcn_sched_barrier(0);
        }
        w4_for<0, NFR>([&](auto Pc) {
          if constexpr (w4_gap_of<NF, RPG>(KIND, decltype(Pc)::value) == m)
            read_pos(std::integral_constant<int, nb>{}, std::integral_constant<int, jn>{}, Pc);
        });
      });
      if constexpr (SCHED == 0) w4_lgkm0();                   // the next k-step's fragments (for j = 2: this stage's last reads)
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  for (int s = 0; s < total; ++s) stage();
  w4_lgkm0();                                                 // (the stale reads of the last k-step: nothing may be in flight past here)
  __builtin_amdgcn_sched_barrier(0);
  W4_STAMP(2);

  // ---- epilogue: the MFMA results may be read 16 passes after the last issue; nothing padded that for an asm MFMA
  w4_for<0, NF>([&](auto Fc) { constexpr int f = decltype(Fc)::value; w4_pad<false, (f < 4)>(acc[f][0], acc[f][1]); });
  if (W4_ABL(4) || p.act == 77) return;                       // act 77: timing probe only (skip the stores)
  const EpiArgs e = epi_of(p);
  const int row_w = m0 + wave * 64 + l31;
  // 8 consecutive columns 16 b + 8 hi .. + 7 of the tile (b = 2 f + q) from two 4-column groups of either half-wave:
  // v_permlane32_swap hands lanes 0-31 the partner's group 2 q, lanes 32-63 the partner's group 2 q + 1
  auto cols8 = [&](auto Fc, auto Ic, auto Qc, float (&v)[8]) {
    constexpr int f = decltype(Fc)::value, i = decltype(Ic)::value, q = decltype(Qc)::value;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[f][i][8 * q + k]), __float_as_uint(acc[f][i][8 * q + 4 + k]),
                                                       false, false);
      v[k] = __uint_as_float(sw[0]); v[4 + k] = __uint_as_float(sw[1]);
    }
  };
  if (p.act == ACT_GEGLU && !slab) {
    // value columns [0, 80) of the tile pair with gate columns [80, 160): 8-column block b with block b + 5, same lane
    if constexpr (NF == 5) {
      w4_for<0, 2>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        const int grow = row_w + 32 * i;
        w4_for<0, 5>([&](auto Bc) {
          constexpr int b = decltype(Bc)::value, gb = b + 5;
          float v[8], g[8];
          cols8(std::integral_constant<int, b / 2>{}, Ic, std::integral_constant<int, b % 2>{}, v);
          cols8(std::integral_constant<int, gb / 2>{}, Ic, std::integral_constant<int, gb % 2>{}, g);
          const int c0 = 16 * b + 8 * hi;
          if (grow < p.M) {
            if (e.bias) {
#pragma unroll
              for (int k = 0; k < 8; ++k) { v[k] += e.bias[n0 + c0 + k]; g[k] += e.bias[n0 + 80 + c0 + k]; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= gelu_f(g[k]);
            if (p.out_f32) store8(reinterpret_cast<float*>(e.C) + (long)grow * e.ldc + n0 / 2 + c0, v);
            else store8(reinterpret_cast<T*>(e.C) + (long)grow * e.ldc + n0 / 2 + c0, v);
          }
        });
      });
    }
    return;
  }
  w4_for<0, 2>([&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    const int grow = row_w + 32 * i;
    w4_for<0, NF>([&](auto Fc) {
      constexpr int f = decltype(Fc)::value;
      w4_for<0, 2>([&](auto Qc) {
        constexpr int q = decltype(Qc)::value;
        float v[8];
        cols8(Fc, Ic, Qc, v);
        const int gcol = n0 + 32 * f + 16 * q + 8 * hi;
        if (grow < p.M && gcol < p.N) {
          if (slab) store8(slab + ((long)zsplit * p.M + grow) * p.N + gcol, v);    // split-K partial: raw accumulators
          else epilogue8<T>(e, v, grow, gcol);
        }
      });
    });
  });
#ifdef W4_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  W4_STAMP(3);
#endif
}

template <int NF, int MODE, int ABL = 0, int SCHED = 1>
int launch_w4_mode(const GemmParams& p0, hipStream_t stream) {
  constexpr int BN = 32 * NF, SMEM = W4_R * (W4_BM / 8 + BN / 8) * 1024;
  auto kern = &gemm_w4_kernel<NF, MODE, SCHED, ABL>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return CL_ELAUNCH;
    attr_set = true;
  }
  if ((p0.a1_group_n && p0.a1_group_n % BN) || (p0.a2_group_n && p0.a2_group_n % BN)) return CL_EINVAL;   // a tile would straddle groups
  GemmParams p = p0;
  const int tm = (p.M + W4_BM - 1) / W4_BM, tn = (p.N + BN - 1) / BN;
  const long tiles = (long)tm * tn;
  const int steps = ((MODE == W4_LINEAR ? 1 : 9) * p.K1 + p.K2) / 64;
  float* slab;
  gemm_pick_splitk(p, tiles, steps, 256, 4, &slab, stream);
  const long nvirt = tiles * p.splitk;
  gemm_tag_note(nvirt, 64 * (W4_NC + W4_NL));
  hipLaunchKernelGGL(kern, dim3((unsigned)(nvirt + gemm_cur_tag())), dim3(64 * (W4_NC + W4_NL)), SMEM, stream, p, tm, tn, slab);
  if (slab) gemm_launch_splitk_reduce_bf16(p, slab, stream);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

template <int NF>
int launch_w4_nf(const GemmParams& p, hipStream_t stream) {
  if (p.mode == GEMM_LINEAR) return launch_w4_mode<NF, W4_LINEAR>(p, stream);
  if (p.K2) return CL_EINVAL;   // a second K segment exists for linear operands only
#ifdef W4_PROBE
  if constexpr (NF == 5) {
    if (p.mode == GEMM_CONV_S1) {
      if (g_w4_sched_host == 0) return launch_w4_mode<NF, W4_CONV_S1, 0, 0>(p, stream);
      switch (g_w4_abl_host) {
#define W4_CASE(a) case a: return launch_w4_mode<NF, W4_CONV_S1, a>(p, stream);
        W4_CASE(1) W4_CASE(2) W4_CASE(3) W4_CASE(4) W4_CASE(5) W4_CASE(6) W4_CASE(7) W4_CASE(13) W4_CASE(21) W4_CASE(37) W4_CASE(33)
#undef W4_CASE
        default: break;
      }
    }
  }
#endif
  if (p.mode == GEMM_CONV_S1) return launch_w4_mode<NF, W4_CONV_S1>(p, stream);
  return launch_w4_mode<NF, W4_CONV_ANY>(p, stream);
}

}  // namespace

#ifdef W4_PROBE
void w4_abl_set(int v) { g_w4_abl_host = v; }
void w4_sched_set(int v) { g_w4_sched_host = v; }
void w4_timing_set(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w4_timing), &buf, sizeof(buf)); }
#endif

// CL_EINVAL = "not a product this kernel covers" (the caller falls back to the other tile kernels)
int launch_gemm_w4(const GemmParams& p, hipStream_t stream, int bn) {
  if (p.atomic || p.K1 % 64 || p.K2 % 64 || p.K1 <= 0) return CL_EINVAL;
  if (p.act == ACT_GEGLU && bn != 160) return CL_EINVAL;
  if (p.act == ACT_GEGLU_SPLIT || p.ln_gamma) return CL_EINVAL;
  if (bn == 160) return launch_w4_nf<5>(p, stream);
  if (bn == 128) return launch_w4_nf<4>(p, stream);
  return CL_EINVAL;
}

}  // namespace cl
