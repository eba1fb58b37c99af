// x-stationary streaming product for the wide-N / short-K linears (bf16; launch configuration 34 of gemm.hip).
//
//   C[M, N] = ( [A1 | A2] . [W1 | W2]^T + bias ) * alpha           K1 + K2 <= 768, N % 32 == 0
//
// Which products: the GEGLU projection (ldm/modules/attention.py:49-56: dim -> 8 dim), the FeedForward output's data gradient
// (attention.py:59-76: dim -> 4 dim), the fused q | k | v of CrossAttention (attention.py:163-171) with their rank-r LoRA
// (cldm/lora.py:285-291) as the second K segment -- K = 320 / 640 (+ 128) against N = 960 ... 5120 at M = 8192 ... 131072.
// In the tile kernel (gemm_fl_kernel) such a product is set-up | load | 5 ... 10 MFMA stages | epilogue per 256 x 160 tile, one
// workgroup per CU, and the epilogue -- 80 KB of stores per tile -- is as long as the rest: (32768, 2560, 320) ran 107 us with
// its stores and 75 us without them (profiles/r05_gemm_xs/), every store request a full 64-byte write (TCC_EA0_WRREQ_64B =
// TCC_EA0_WRREQ): the stores are not malformed, they are serialised behind the tile's compute.
//
// Structure here: a workgroup of 4 waves owns 128 rows of x and a RUN of output columns.
//   * every wave loads its 32 rows of [x | t] ONCE, straight into registers, as the B-operand fragments of
//     v_mfma_f32_32x32x16_bf16 (K = 320: 80 VGPRs): x never touches LDS;
//   * [W | B] streams through LDS in chunks of 32 output columns, LDS-DMA (global_load_lds_dwordx4) RING chunks deep, XOR-swizzled on
//     the SOURCE side so that the fragment reads (ds_read_b128: row n, 16-byte slot) are bank-conflict-free;
//   * per chunk and wave: K / 16 MFMAs of the TRANSPOSED product C^T[32 n x 32 m] += W[32 n x 16 k] . x^T[16 k x 32 m], so a lane ends
//     up with 4 consecutive output columns of one row per 8-column group; v_permlane32_swap pairs the half-waves' groups into 16-byte
//     row-contiguous stores (T21 of the CDNA4 guide), and chunk c - 1 is stored behind the MFMAs of chunk c (two accumulator sets):
//     loads, MFMAs and stores of a workgroup overlap, and two workgroups share a CU at K <= 448;
//   * one counted vmcnt covers the DMA ring AND the stores (gfx9: one counter, in-order retirement).
// Measured (profiles/r05_gemm_xs/): (32768, 2560, 320) 107 -> 70 us, (32768, 1280, 320) 55 -> 44, (8192, 5120, 640) 79 -> 70.
//
// Epilogues: bias, alpha / alpha_n, beta * residual (XS_RES), fused GEGLU on natural-order rows (XS_GEGLU, act 3).  Prologue: LayerNorm
// of the x rows in registers (GemmParams::ln_gamma).
// Not covered (the launcher returns CL_EINVAL and gemm.hip falls back to its own rules): fp32 storage, conv modes, rowbias / SiLU,
// fp32 output, K segments other than {320, 640} (+ {0, 128}), grouped first segments.
#include <type_traits>
#include "gemm.h"

namespace cl {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
template <int OFF> __device__ __forceinline__ u32x4_t xs_rd128(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void xs_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// s_waitcnt vmcnt(n) for a run-time n (the count is an instruction immediate; 6 bits on gfx9)
__device__ __forceinline__ void xs_vm_wait(int n) {
  switch (n) {
#define VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    VMW(0) VMW(1) VMW(2) VMW(3) VMW(4) VMW(5) VMW(6) VMW(7) VMW(8) VMW(9) VMW(10) VMW(11) VMW(12) VMW(13) VMW(14) VMW(15) VMW(16)
    VMW(17) VMW(18) VMW(19) VMW(20) VMW(21) VMW(22) VMW(23) VMW(24) VMW(25) VMW(26) VMW(27) VMW(28) VMW(29) VMW(30) VMW(31) VMW(32)
#undef VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

enum { XS_PLAIN = 0, XS_RES = 1, XS_GEGLU = 2 };

// gelu(g) = 0.5 g (1 + erf(g / sqrt 2)) for the bf16 GEGLU epilogue: erfc by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 absolute,
// three orders below the bf16 rounding of the product it feeds; ocml's erff costs ~4x the VALU issue, and the 16 evaluations per
// lane and block were what bounded the fused projection: 310 us with it at (131072, 2560, 320) against 310 for the plain product
// that stores twice the bytes).  1 + erf is formed without cancellation on either side: c = erfc(|z|), then c or 2 - c.
__device__ __forceinline__ float gelu_as(float g) {
  const float z = fabsf(g) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float c = pl * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  return 0.5f * g * (g < 0.f ? c : 2.0f - c);
}

// 16 bytes, global -> VGPRs, outside the compiler's own vmcnt bookkeeping (the caller waits with a counted vmcnt and then ties
// the registers with an empty asm); early-clobber: the destination may not alias the address pair
__device__ __forceinline__ void xs_gload128(u32x4_t& dst, const void* src) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(src) : "memory");
}

}  // namespace

// KS1 / KS2: 16-element k-steps of the two K segments.  RING: LDS chunk slots.  MINW: waves per SIMD the register budget allows for
// (2 = two workgroups per CU).  EPI: XS_PLAIN  C = (acc + bias) alpha;  XS_RES  ... + beta residual;  XS_GEGLU  W's rows are
// [value (N / 2) | gate (N / 2)], C[M, N / 2] = (value + bias) * gelu(gate + bias) -- chunks alternate value / gate rows of the same
// 32 output columns.  cpw: 32-column output blocks per workgroup; blockIdx.y selects the run of columns.
template <int KS1, int KS2, int RING, int MINW, int EPI, bool LNP = false>
__global__ __launch_bounds__(256, MINW) void gemm_xs_kernel(GemmParams p, int cpw) {
  static_assert(!LNP || KS2 == 0, "the LayerNorm prologue normalises the first K segment: no second one");
  constexpr int KS = KS1 + KS2;
  constexpr int ROWB = KS * 32;                 // bytes of one [W | B] row image
  constexpr int CPRW = ROWB / 16;               // 16-byte slots per row
  constexpr bool SW16 = (ROWB % 256) == 0;      // rows start on the same bank: swizzle over 16 slots instead of 8
  constexpr int GS = SW16 ? 8 : 4;              // k-steps per swizzle group
  constexpr int CHUNK = 32 * ROWB, PIECES = CHUNK / 1024, DPC = PIECES / 4, D = RING - 1;
  static_assert(PIECES % 4 == 0 && CPRW % (2 * GS) == 0 && (KS1 * 2) % (2 * GS) == 0, "segment / swizzle-group alignment");
  static_assert(RING == 2 || RING == 3, "ring depth");
  constexpr bool GEGLU = EPI == XS_GEGLU, RES = EPI == XS_RES;
  constexpr int CPB = GEGLU ? 2 : 1;            // chunks per 32-column output block
  constexpr int SPB = 2, RPB = RES ? 2 : 0;     // store / residual-load instructions per output block and wave
  constexpr int NACC = GEGLU ? 4 : 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];          // RING chunks of [W | B], then the bias image (fp32)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int ncols = GEGLU ? p.N / 2 : p.N;      // output columns
  const int n_base = (int)blockIdx.y * cpw * 32;
  const int nob = min(cpw, (ncols - n_base) / 32);       // output blocks of this workgroup
  if (nob <= 0 || (int)blockIdx.x >= (p.M + 127) / 128) return;      // (past M: launch-tag workgroups, csrc/debug_hooks.h)
  const int nch = nob * CPB;                             // chunks (= loop iterations)
  // rows past M repeat row M - 1: same operands, same results, the same bytes stored twice -- no predication anywhere, so the
  // instruction counts the vmcnt arithmetic relies on are exact
  const long row = min((long)blockIdx.x * 128 + wave * 32 + l31, (long)p.M - 1);

  // ---- [x | t] rows of this wave -> registers (B operand: column = row l31 of the wave's block, k = 16 j + 8 hi .. + 7)
  u32x4_t xa[KS];
  {
    const char* xp = (const char*)p.A1 + row * p.lda1 * 2 + hi * 16;
#pragma unroll
    for (int j = 0; j < KS1; ++j) xa[j] = *reinterpret_cast<const u32x4_t*>(xp + j * 32);
    if constexpr (KS2 > 0) {
      const long goff = p.a2_group_n ? (long)(n_base / p.a2_group_n) * (KS2 * 16) : 0;
      const char* tp = (const char*)p.A2 + (row * p.lda2 + goff) * 2 + hi * 16;
#pragma unroll
      for (int j = 0; j < KS2; ++j) xa[KS1 + j] = *reinterpret_cast<const u32x4_t*>(tp + j * 32);
    }
  }
  float* lnaff = reinterpret_cast<float*>(smem + RING * CHUNK) + cpw * CPB * 32;     // LNP: gamma[K1] | beta[K1] (fp32)
  if constexpr (LNP) {
    for (int i = tid; i < 16 * KS1; i += 256) { lnaff[i] = p.ln_gamma[i]; lnaff[16 * KS1 + i] = p.ln_beta[i]; }
    // LayerNorm prologue (attention.py:271-275 norm1/2/3, cl_layernorm_fwd's arithmetic): lanes (l31, hi = 0 / 1) hold one row
    // between them -- 8 KS1 elements each; two passes over the registers (mean, then centred squares), fp32; the normalised row
    // is rounded to bf16 as the stand-alone kernel stores it and goes back into the fragment registers
    constexpr float invK = 1.0f / (16 * KS1);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < KS1; ++j) {
      const uint32_t w[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * invK;
    // (the fragments pass through an empty asm between the passes: otherwise the compiler keeps all 16 KS1 unpacked floats
    // of pass one alive for passes two and three -- 250 spilled registers)
#pragma unroll
    for (int j = 0; j < KS1; ++j) asm volatile("" : "+v"(xa[j]));
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < KS1; ++j) {
      const uint32_t w[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d0 = __uint_as_float(w[e] << 16) - mean, d1 = __uint_as_float(w[e] & 0xffff0000u) - mean;
        sq += d0 * d0 + d1 * d1;
      }
    }
    sq += __shfl_xor(sq, 32, 64);
#pragma unroll
    for (int j = 0; j < KS1; ++j) asm volatile("" : "+v"(xa[j]));
    const float rstd = rsqrtf(sq * invK + p.ln_eps);
    if (p.ln_stats && blockIdx.y == 0 && hi == 0) { p.ln_stats[row * 2] = mean; p.ln_stats[row * 2 + 1] = rstd; }
    // gamma | beta from their LDS image (staged below the bias image by the whole workgroup), four k-steps at a time so that
    // the affine vectors of the whole row are never live together
    __syncthreads();
    const uint32_t ga = (uint32_t)(uintptr_t)lnaff + hi * 32;          // gamma of k = 16 j + 8 hi .. + 7 at ga + 64 j
    u32x4_t rr[2][4];                                                 // (raw LDS reads, one k-step ahead: the compiler would issue all 4 KS1 at once)
    rr[0][0] = xs_rd128<0>(ga); rr[0][1] = xs_rd128<16>(ga); rr[0][2] = xs_rd128<KS1 * 64>(ga); rr[0][3] = xs_rd128<KS1 * 64 + 16>(ga);
    sfor<0, KS1>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (j + 1 < KS1) {
        constexpr int n = j + 1;
        rr[n & 1][0] = xs_rd128<n * 64>(ga); rr[n & 1][1] = xs_rd128<n * 64 + 16>(ga);
        rr[n & 1][2] = xs_rd128<KS1 * 64 + n * 64>(ga); rr[n & 1][3] = xs_rd128<KS1 * 64 + n * 64 + 16>(ga);
        xs_lgkm<4>();
      } else {
        xs_lgkm<0>();
      }
      u32x4_t (&r)[4] = rr[j & 1];
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(r[q]));
      const float gg[8] = {__uint_as_float(r[0].x), __uint_as_float(r[0].y), __uint_as_float(r[0].z), __uint_as_float(r[0].w),
                           __uint_as_float(r[1].x), __uint_as_float(r[1].y), __uint_as_float(r[1].z), __uint_as_float(r[1].w)};
      const float bb[8] = {__uint_as_float(r[2].x), __uint_as_float(r[2].y), __uint_as_float(r[2].z), __uint_as_float(r[2].w),
                           __uint_as_float(r[3].x), __uint_as_float(r[3].y), __uint_as_float(r[3].z), __uint_as_float(r[3].w)};
      const uint32_t w[4] = {xa[j].x, xa[j].y, xa[j].z, xa[j].w};
      uint32_t o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = (__uint_as_float(w[e] << 16) - mean) * rstd * gg[2 * e] + bb[2 * e];
        const float v1 = (__uint_as_float(w[e] & 0xffff0000u) - mean) * rstd * gg[2 * e + 1] + bb[2 * e + 1];
        o[e] = pack2bf(v0, v1);
      }
      xa[j] = u32x4_t{o[0], o[1], o[2], o[3]};
      asm volatile("" : "+v"(xa[j]));        // (volatile: ordered before the next k-step's reads -- the arithmetic cannot be sunk past them)
    });
  }
  // bias image: [nob * 32 floats] (GEGLU: value part, then the gate part)
  float* sb = reinterpret_cast<float*>(smem + RING * CHUNK);
  constexpr int NBV = (2560 + 255) / 256;
  float bv[NBV];
#pragma unroll
  for (int i = 0; i < NBV; ++i) {
    const int cidx = tid + i * 256;
    int src = n_base + cidx;
    if (GEGLU && cidx >= nob * 32) src = ncols + n_base + (cidx - nob * 32);
    bv[i] = (p.bias && cidx < nch * 32) ? p.bias[src] : 0.f;
  }

  // ---- chunk DMA: piece q of a chunk image = 64 slots of 16 bytes; slot 64 q + lane sits in row n at physical slot s and holds the
  // LOGICAL slot s ^ swz(n) of that row of [W | B] (swz permutes within aligned groups of 8 / 16 slots; K1 is whole groups)
  int woff[DPC];          // byte offset from the chunk's first row in W1 (or W2), bit 31 = "second segment"
#pragma unroll
  for (int j = 0; j < DPC; ++j) {
    const int q = (wave + 4 * j) * 64 + lane, n = q / CPRW, s = q - n * CPRW;
    const int swz = SW16 ? (((n >> 1) & 7) | ((n & 1) << 3)) : ((n >> 1) & 7);
    const int ls = s ^ swz;
    woff[j] = ls < KS1 * 2 ? (int)(n * p.ldw1 * 2 + ls * 16) : (int)((n * p.ldw2 * 2 + (ls - KS1 * 2) * 16) | 0x80000000u);
  }
  auto issue = [&](int c) {
    const int wrow = GEGLU ? ((c & 1) ? ncols : 0) + n_base + (c >> 1) * 32 : n_base + c * 32;     // first W row of the chunk
    const char* w1 = (const char*)p.W1 + (long)wrow * p.ldw1 * 2;
    const char* w2 = KS2 > 0 ? (const char*)p.W2 + (long)wrow * p.ldw2 * 2 : w1;
    char* dst = smem + (c % RING) * CHUNK;
#pragma unroll
    for (int j = 0; j < DPC; ++j) {
      const char* src = (KS2 > 0 && woff[j] < 0) ? w2 + (woff[j] & 0x7fffffff) : w1 + woff[j];
      glds16(src, dst + (wave + 4 * j) * 1024);
    }
  };
  // [W | B] fragment (A operand: row = output column l31 of the chunk, k = 16 j + 8 hi): logical slot 2 j + hi of row l31 =
  // physical slot (2 j) ^ (hi ^ swz(l31)): the XOR only touches the slot's position inside its swizzle group
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t fo[GS];
  {
    const int swz = SW16 ? (((l31 >> 1) & 7) | ((l31 & 1) << 3)) : ((l31 >> 1) & 7);
    const int t = hi ^ swz;
#pragma unroll
    for (int m = 0; m < GS; ++m) fo[m] = lds0 + l31 * ROWB + (((2 * m) ^ t) * 16);
  }

#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < nch) issue(s);
#pragma unroll
  for (int i = 0; i < NBV; ++i) {
    const int cidx = tid + i * 256;
    if (cidx < nch * 32) sb[cidx] = bv[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the bias image is complete before the first barrier releases its readers

  const float alpha = p.alpha, beta = p.beta;
  const int alpha_n = p.alpha_n;
  // lane (m = l31, hi) holds columns 8 (r / 4) + 4 hi + (r % 4) of a block; after the swaps lanes 0-31 hold columns 16 q .. + 7,
  // lanes 32-63 columns 16 q + 8 .. + 15 (16-byte row-contiguous stores; the residual is loaded in that layout and un-swapped)
  bf16_t* yrow = reinterpret_cast<bf16_t*>(p.C) + row * p.ldc + n_base + 8 * hi;
  const bf16_t* rrow = RES ? reinterpret_cast<const bf16_t*>(p.residual) + row * p.ldr + n_base + 8 * hi : nullptr;
  u32x4_t rs[2];
  auto load_res = [&](int b) {
    if constexpr (RES) { xs_gload128(rs[0], rrow + b * 32); xs_gload128(rs[1], rrow + b * 32 + 16); }
  };
  auto store_block = [&](int b, f32x16_t a, const f32x16_t& gate) {
    if constexpr (GEGLU) {
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] *= gelu_as(gate[i]);
    } else {
      const float al = (alpha_n > 0 && n_base + b * 32 >= alpha_n) ? 1.0f : alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] *= al;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if constexpr (RES) {
        asm volatile("" : "+v"(rs[q]));
        const auto ux = __builtin_amdgcn_permlane32_swap(rs[q].x, rs[q].z, false, false);   // back to the accumulator layout
        const auto uy = __builtin_amdgcn_permlane32_swap(rs[q].y, rs[q].w, false, false);
        const uint32_t r4[4] = {ux[0], uy[0], ux[1], uy[1]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          a[8 * q + 2 * i] += beta * __uint_as_float(r4[i] << 16);
          a[8 * q + 2 * i + 1] += beta * __uint_as_float(r4[i] & 0xffff0000u);
        }
      }
      uint32_t ax = pack2bf(a[8 * q], a[8 * q + 1]), ay = pack2bf(a[8 * q + 2], a[8 * q + 3]);
      uint32_t bx = pack2bf(a[8 * q + 4], a[8 * q + 5]), by = pack2bf(a[8 * q + 6], a[8 * q + 7]);
      const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
      const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
      const u32x4_t w = {rx[0], ry[0], rx[1], ry[1]};
      *reinterpret_cast<u32x4_t*>(yrow + b * 32 + 16 * q) = w;
    }
  };
  // Vector-memory issue order of iteration i:  residual loads of block i - 1 | DMA of chunk i + D | (MFMAs) | stores of the block
  // that completed with chunk i - 1.  Block b completes with chunk CPB b + CPB - 1.
  auto block_done_before = [&](int i) { return i >= CPB && (i % CPB) == 0; };     // iteration i stores block i / CPB - 1

  // accumulator sets are indexed statically (chunk c lives in set c % NACC): the loop is written NACC iterations at a time
  f32x16_t acc[NACC];
  auto iter = [&](int c, auto SLOT) {
    constexpr int S = decltype(SLOT)::value;
    // This wave's DMA of chunk c (issued in iteration c - D, after that iteration's residual loads) has landed when at most the
    // operations issued AFTER it are outstanding: the stores of iteration c - D, and everything of iterations c - D + 1 .. c - 1.
    // (In the first D - 1 iterations the DMAs of chunks c + 1 .. D - 1 -- issued by the prologue, right behind DMA(c) -- are
    // among them: found by tests/test_xs_vmcnt_model.py, which replays this arithmetic; without the term iteration 0 of a
    // 3-slot ring waited for chunk 1 as well.)
    int allow = 0;
#pragma unroll
    for (int j = 1; j < D; ++j) allow += (c + j < D && c + j < nch) ? DPC : 0;
#pragma unroll
    for (int k = D; k >= 1; --k) {
      const int i = c - k;
      if (i < 0) continue;
      if (k != D) allow += (block_done_before(i) ? RPB : 0) + ((i + D < nch) ? DPC : 0);
      allow += block_done_before(i) ? SPB : 0;
    }
    xs_vm_wait(allow);
    __builtin_amdgcn_s_barrier();                      // every wave's pieces landed; the slot of chunk c - 1 is free
    __builtin_amdgcn_sched_barrier(0);
    const bool fin = (S % CPB) == 0 && c >= CPB;       // = block_done_before(c): c = S mod NACC, NACC a multiple of CPB
    if (fin) load_res(c / CPB - 1);
    if (c + D < nch) issue(c + D);
    const uint32_t cb = (c % RING) * CHUNK;
    // accumulators start from the bias of their columns (fp32, from the LDS image)
    u32x4_t b4[4];
    const int bidx = GEGLU ? ((c & 1) ? nob * 32 : 0) + (c >> 1) * 32 : c * 32;
    const uint32_t sba = lds0 + RING * CHUNK + (bidx + 4 * hi) * 4;
    b4[0] = xs_rd128<0>(sba); b4[1] = xs_rd128<32>(sba); b4[2] = xs_rd128<64>(sba); b4[3] = xs_rd128<96>(sba);
    // fragment reads AHEAD k-steps in front of their MFMAs
    u32x4_t wf[KS];
    constexpr int AHEAD = 4;
    sfor<0, AHEAD>([&](auto J) { constexpr int j = decltype(J)::value; wf[j] = xs_rd128<(j / GS) * (GS * 32)>(cb + fo[j % GS]); });
    xs_lgkm<AHEAD>();                                  // the four bias reads are older than every fragment read
    f32x16_t a;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      asm volatile("" : "+v"(b4[q]));
      a[4 * q] = __uint_as_float(b4[q].x); a[4 * q + 1] = __uint_as_float(b4[q].y);
      a[4 * q + 2] = __uint_as_float(b4[q].z); a[4 * q + 3] = __uint_as_float(b4[q].w);
    }
    sfor<0, KS>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (j + AHEAD < KS) {
        wf[j + AHEAD] = xs_rd128<((j + AHEAD) / GS) * (GS * 32)>(cb + fo[(j + AHEAD) % GS]);
        xs_lgkm<AHEAD>();
      } else {
        xs_lgkm<KS - 1 - j>();
      }
      asm volatile("" : "+v"(wf[j]));
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[j]), __builtin_bit_cast(bf16x8_t, xa[j]), a, 0, 0, 0);
    });
    acc[S] = a;
    if (fin) {                                         // the finished block's stores go out behind this chunk's MFMAs
      if constexpr (RES) xs_vm_wait((c + D < nch) ? DPC : 0);      // its residual (older than this iteration's DMA) has landed
      store_block(c / CPB - 1, acc[(S + NACC - CPB) % NACC], acc[(S + NACC - 1) % NACC]);
    }
  };
  for (int c0 = 0; c0 < nch; c0 += NACC)
    sfor<0, NACC>([&](auto SLOT) { if (c0 + decltype(SLOT)::value < nch) iter(c0 + decltype(SLOT)::value, SLOT); });
  if constexpr (RES) { load_res(nob - 1); xs_vm_wait(0); }
  sfor<0, NACC>([&](auto SLOT) {                       // the last block: value (and gate) sets by their static index
    constexpr int S = decltype(SLOT)::value;
    if ((CPB * (nob - 1)) % NACC == S) store_block(nob - 1, acc[S], acc[(S + CPB - 1) % NACC]);
  });
}

// (Tried and dropped, profiles/r05_gemm_xs/probe_xs_three_workgroups_per_cu.log: the K = 320 forms with a 2-chunk ring at THREE
// workgroups per CU (166 VGPRs allow it) -- plain N = 320 15.4 -> 14.6 us, N = 1280 / 2560 unchanged, the residual and GEGLU forms
// spill at 168 registers and lose 15-100 %.)

namespace {

template <int KS1, int KS2, int RING, int MINW, int EPI, bool LNP = false>
int launch_xs(const GemmParams& p, hipStream_t stream, int nsplit) {
  constexpr int KS = KS1 + KS2, CHUNK = 32 * KS * 32, CPB = EPI == XS_GEGLU ? 2 : 1, MAXB = 80 / CPB;
  const int ncols = EPI == XS_GEGLU ? p.N / 2 : p.N;
  const int blocks = ncols / 32;
  const int groups = p.a2_group_n ? p.N / p.a2_group_n : 1;
  const int bpg = blocks / groups;                       // 32-column blocks per group: a workgroup's run never straddles groups
  const int rb = (p.M + 127) / 128;
  if (nsplit <= 0) {
    // as many workgroups as the chip holds at once (two per CU at K <= 448, one above) where the run stays >= 5 blocks
    const long want = MINW >= 2 ? 256L * MINW : 256;
    nsplit = 1;
    while ((long)rb * groups * nsplit < want && bpg % (nsplit * 2) == 0 && bpg / (nsplit * 2) >= 5) nsplit *= 2;
  }
  if (nsplit > bpg) nsplit = bpg;
  while (bpg % nsplit) --nsplit;
  while (bpg / nsplit > MAXB) {                           // the bias image holds 2560 floats
    int k = nsplit + 1;
    while (k <= bpg && bpg % k) ++k;
    nsplit = k;
  }
  const int cpw = bpg / nsplit;
  const int smem = RING * CHUNK + cpw * CPB * 32 * 4 + (LNP ? 2 * 16 * KS1 * 4 : 0);
  auto kern = &gemm_xs_kernel<KS1, KS2, RING, MINW, EPI, LNP>;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RING * CHUNK + 2560 * 4 + (LNP ? 2 * 16 * KS1 * 4 : 0)) != hipSuccess)
      return CL_ELAUNCH;
    attr_set = true;
  }
  // launch tag: `tag` extra (empty) columns of workgroups -- the trace shows (rb + tag) * groups * nsplit workgroups
  gemm_tag_note((long)rb * groups * nsplit, 256);
  hipLaunchKernelGGL(kern, dim3((unsigned)(rb + gemm_cur_tag()), (unsigned)(groups * nsplit)), dim3(256), smem, stream, p, cpw);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

template <int EPI>
int launch_xs_ln(const GemmParams& p, hipStream_t stream, int nsplit) {
  if (p.K2 || !p.ln_beta) return CL_EINVAL;
  if (p.K1 == 320) return launch_xs<20, 0, 3, 2, EPI, true>(p, stream, nsplit);
  if (p.K1 == 640) return launch_xs<40, 0, 3, 1, EPI, true>(p, stream, nsplit);
  return CL_EINVAL;
}

template <int EPI>
int launch_xs_k(const GemmParams& p, hipStream_t stream, int nsplit) {
  if (p.K1 == 320 && p.K2 == 0) return launch_xs<20, 0, 3, 2, EPI>(p, stream, nsplit);
  if (p.K1 == 320 && p.K2 == 128) return launch_xs<20, 8, 2, 2, EPI>(p, stream, nsplit);
  if (p.K1 == 640 && p.K2 == 0) return launch_xs<40, 0, 3, 1, EPI>(p, stream, nsplit);
  if (p.K1 == 640 && p.K2 == 128) return launch_xs<40, 8, 3, 1, EPI>(p, stream, nsplit);
  return CL_EINVAL;
}

}  // namespace

// CL_EINVAL = "not a product this kernel covers" (the caller falls back to the tile kernels); nsplit 0 = the launcher's rule
int launch_gemm_xs(const GemmParams& p, hipStream_t stream, int nsplit) {
  const bool geglu = p.act == ACT_GEGLU_SPLIT;
  if (p.mode != GEMM_LINEAR || p.atomic || p.out_f32 || (p.act != ACT_NONE && !geglu) || p.rowbias || p.a1_group_n) return CL_EINVAL;
  if (p.N % (geglu ? 64 : 32) || p.M < 128 || (p.alpha_n % 32)) return CL_EINVAL;
  if (geglu && (p.residual || p.alpha != 1.0f || p.alpha_n || p.a2_group_n)) return CL_EINVAL;
  if (p.a2_group_n && (p.a2_group_n % 32 || p.N % p.a2_group_n || !p.K2)) return CL_EINVAL;
  if (p.lda1 % 8 || p.ldw1 % 8 || p.ldc % 8 || (p.K2 && (p.lda2 % 8 || p.ldw2 % 8)) || (p.residual && p.ldr % 8)) return CL_EINVAL;   // 16-byte vectors
  if ((long)32 * p.ldw1 * 2 + 1536 >= (1L << 31) || (p.K2 && (long)32 * p.ldw2 * 2 + 1536 >= (1L << 31))) return CL_EINVAL;
  if (p.ln_gamma) {
    if (p.residual) return CL_EINVAL;
    return geglu ? launch_xs_ln<XS_GEGLU>(p, stream, nsplit) : launch_xs_ln<XS_PLAIN>(p, stream, nsplit);
  }
  if (geglu) return launch_xs_k<XS_GEGLU>(p, stream, nsplit);
  if (p.residual) return launch_xs_k<XS_RES>(p, stream, nsplit);
  return launch_xs_k<XS_PLAIN>(p, stream, nsplit);
}

}  // namespace cl
