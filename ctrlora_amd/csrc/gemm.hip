// MFMA GEMM / implicit-GEMM 3x3 convolution core for gfx950 (MI355X).
//
//   out[M,N] = epilogue( A1[M,K1].W1[N,K1]^T  (+ A2[M,K2].W2[N,K2]^T) )
//
// One kernel family serves every dense contraction on the CtrLoRA hot path:
//   * nn.Linear / LoRACompatibleLinear  (cldm/lora.py:285-291): the rank-r
//     LoRA up-projection is folded in as a second K segment
//     ([x | xA^T] . [W | B]^T), fp32-accumulated in the same MFMA chain;
//   * 1x1 convs incl. ControlNet zero-convs (cldm/cldm.py:281-282) with the
//     control_scale multiply and the `skip + control` add (cldm/cldm.py:41)
//     in the epilogue, writing straight into the decoder concat buffer (ldc);
//   * ResBlock / Downsample / Upsample 3x3 convs
//     (ldm/modules/diffusionmodules/openaimodel.py:108-118,150,203,229) as
//     implicit GEMM over NHWC activations, with bias + time-embedding add
//     (openaimodel.py:272) or skip add (:274) in the epilogue;
//   * all data-gradients (same kernel, pre-transposed / tap-flipped weights).
//
// Structure (per workgroup): NW wave64s in a WGM x WGN grid over a BM x BN
// output tile.  K is walked in 64-byte "substeps" (32 bf16 / 16 f32 per row):
// a substep's A and B rows are staged HBM->LDS by global_load_lds_dwordx4 (no
// VGPR round trip) into one slot of an R-slot LDS ring; a slot is a stack of
// 16-row x 64-byte groups (1 KiB = one wave-wide DMA instruction), which is
// conflict-free for the ds_read_b128 fragment reads of v_mfma_f32_16x16x32_bf16
// (or 4x v_mfma_f32_16x16x4_f32 in fp32 parity mode).  Each loop iteration
// consumes KSUB substeps behind ONE raw s_barrier and a COUNTED s_waitcnt
// vmcnt, so R-KSUB substeps of DMA stay in flight across the barrier while
// the MFMAs run.  BN = 160 tiles exist because every wide channel count of
// SD1.5 (320/640/960/1280/1920/2560) is a multiple of 160: no N-padding waste
// and workgroup counts that are multiples of the 256 CUs at the hot shapes.
//
// Deep-K / small-MN products (the 8x8 and 16x16 levels: K = 9*1280..9*2560,
// M = B*64 .. B*256) are split along K: each split writes an fp32 partial slab
// into the host-provided workspace and a second kernel sums the slabs and
// applies the epilogue -- deterministic, unlike atomics.  (fp32 atomics remain
// for the weight-gradient accumulation into the flat gradient buffer.)
//
// Workgroup ids are remapped so that each XCD (private L2) owns a contiguous
// range of output tiles: neighbouring tiles share A rows / conv halos.
#include <type_traits>
#include "gemm.h"
#include <algorithm>
#include <map>
#include <vector>
#include <utility>
#include "mma.h"
#include "gemm_epi.h"

namespace cl {

// Split-K scratch: a default region plus optional per-stream regions, so that two streams running
// contractions concurrently (ControlNet trunk || UNet encoder) never share partial slabs.
static void* g_ws = nullptr;
static long g_ws_bytes = 0;
struct StreamWs { hipStream_t st; void* p; long bytes; };
static StreamWs g_sws[4] = {};
static thread_local hipStream_t t_stream = nullptr;     // stream of the launch being prepared
void gemm_set_workspace(void* p, long bytes) { g_ws = p; g_ws_bytes = bytes; }
int gemm_set_stream_workspace(hipStream_t st, void* p, long bytes) {
  for (auto& e : g_sws) if (e.st == st || e.p == nullptr) { e.st = st; e.p = p; e.bytes = bytes; return CL_OK; }
  return CL_EINVAL;
}
static void ws_for(hipStream_t st, void** p, long* bytes) {
  for (auto& e : g_sws) if (e.p && e.st == st) { *p = e.p; *bytes = e.bytes; return; }
  *p = g_ws; *bytes = g_ws_bytes;
}
void gemm_get_workspace(void** p, long* bytes) { ws_for(t_stream, p, bytes); }
void gemm_get_workspace_for(hipStream_t st, void** p, long* bytes) { ws_for(st, p, bytes); }

// ---- launch tags (debug hook)
int g_gemm_tag_on = 0;
static thread_local int t_tag = 0;
struct TagEntry { long v[12]; };
static std::vector<TagEntry> g_tags;
int gemm_cur_tag() { return t_tag; }
void gemm_tag_note(long real_wgs, int wg_size) {
  if (t_tag > 0 && t_tag <= (int)g_tags.size()) { g_tags[t_tag - 1].v[9] = real_wgs; g_tags[t_tag - 1].v[10] = wg_size; }
}
int gemm_tag_count() { return (int)g_tags.size(); }
int gemm_tag_get(int i, long* out) {
  if (i < 0 || i >= (int)g_tags.size()) return CL_EINVAL;
  for (int k = 0; k < 12; ++k) out[k] = g_tags[i].v[k];
  return CL_OK;
}
static int tag_for(const GemmParams& p, int dtype) {
  if (!g_gemm_tag_on) return 0;
  const long key[8] = {dtype, p.mode, p.M, p.N, p.K1, p.K2, p.act, p.residual ? 1 : 0};
  for (size_t i = 0; i < g_tags.size(); ++i)
    if (std::equal(key, key + 8, g_tags[i].v)) { ++g_tags[i].v[11]; return (int)i + 1; }
  if (g_tags.size() >= 255) return 0;
  TagEntry e{};
  std::copy(key, key + 8, e.v);
  e.v[8] = (long)g_tags.size() + 1; e.v[11] = 1;
  g_tags.push_back(e);
  return (int)g_tags.size();
}

template <int N> __device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Accumulator tile -> memory: each wave stages 32 (or 16) x WN fp32 through LDS so that residual
// reads and output writes are whole 16-byte vectors along rows, then applies the epilogue.
template <typename T, int FM, int FN, bool PAIR = false>
__device__ __forceinline__ void store_tile(const GemmParams& p, f32x4_t (&acc)[FM][FN], float* stg, int row0,
                                           int col0, int lane, float* slab, int zsplit, int wn = 0,
                                           float* stg_partner = nullptr) {
  constexpr int WN = FN * 16;
  constexpr int EST = WN + 4;
  constexpr int LPR = WN / 8;      // lanes per output row
  constexpr int RPI = 64 / LPR;    // rows per read iteration (lanes idle when 64 % LPR != 0)
  constexpr int PASSES = (FM + 1) / 2;
  constexpr int FPP = (FM >= 2) ? 2 : 1;  // m-frags per pass
  constexpr int ITERS = (FPP * 16 + RPI - 1) / RPI;
  const EpiArgs e = epi_of(p);
  // PAIR: the two waves of a tile row (wn = 0: value columns, wn = 1: gate columns) combine for GEGLU
  const bool geglu = PAIR && p.act == ACT_GEGLU && !slab;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
    for (int i2 = 0; i2 < FPP; ++i2) {
      const int i = ps * 2 + i2;
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stg[(i2 * 16 + (lane >> 4) * 4 + r) * EST + j * 16 + (lane & 15)] = acc[i][j][r];
    }
    if constexpr (PAIR) {
      if (geglu) {
        __syncthreads();   // both waves' staging is complete (uniform branch: p.act)
        const float* sv = wn == 0 ? stg : stg_partner;
        const float* sg = wn == 0 ? stg_partner : stg;
        const int n0 = col0 - wn * WN;                 // first column of the 2*WN-wide tile
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          if ((it & 1) != wn) continue;                // the two waves share the rows
          const int rr = it * RPI + lane / LPR;
          const int cg = (lane % LPR) * 8;
          const int grow = row0 + ps * 32 + rr;
          if (lane < RPI * LPR && rr < FPP * 16 && grow < p.M) {
            float v[8], gt[8];
            const float4 a0 = *reinterpret_cast<const float4*>(&sv[rr * EST + cg]);
            const float4 a1 = *reinterpret_cast<const float4*>(&sv[rr * EST + cg + 4]);
            const float4 g0 = *reinterpret_cast<const float4*>(&sg[rr * EST + cg]);
            const float4 g1 = *reinterpret_cast<const float4*>(&sg[rr * EST + cg + 4]);
            v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
            gt[0] = g0.x; gt[1] = g0.y; gt[2] = g0.z; gt[3] = g0.w; gt[4] = g1.x; gt[5] = g1.y; gt[6] = g1.z; gt[7] = g1.w;
            if (e.bias) {
#pragma unroll
              for (int k = 0; k < 8; ++k) { v[k] += e.bias[n0 + cg + k]; gt[k] += e.bias[n0 + WN + cg + k]; }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= gelu_f(gt[k]);
            if (p.out_f32) store8(reinterpret_cast<float*>(e.C) + (long)grow * e.ldc + n0 / 2 + cg, v);
            else store8(reinterpret_cast<T*>(e.C) + (long)grow * e.ldc + n0 / 2 + cg, v);
          }
        }
        __syncthreads();   // staging free for the next pass
        continue;
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int rr = it * RPI + lane / LPR;
      const int cg = (lane % LPR) * 8;
      const int grow = row0 + ps * 32 + rr;
      const int gcol = col0 + cg;
      if (lane < RPI * LPR && rr < FPP * 16 && grow < p.M && gcol < p.N) {
        float v[8];
        const float4 s0 = *reinterpret_cast<const float4*>(&stg[rr * EST + cg]);
        const float4 s1 = *reinterpret_cast<const float4*>(&stg[rr * EST + cg + 4]);
        v[0] = s0.x; v[1] = s0.y; v[2] = s0.z; v[3] = s0.w;
        v[4] = s1.x; v[5] = s1.y; v[6] = s1.z; v[7] = s1.w;
        if (slab) {   // split-K partial: raw accumulators, epilogue applied by the reduce kernel
          store8(slab + ((long)zsplit * p.M + grow) * p.N + gcol, v);
        } else if (p.act != 77) {   // act 77: timing probe only (skip the stores)
          epilogue8<T>(e, v, grow, gcol);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, int KSUB, int R>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void gemm_kernel(GemmParams p, int tiles_m, int tiles_n,
                                                              float* __restrict__ slab) {
  constexpr int NW = WGM * WGN;
  constexpr int KPB = 64 / (int)sizeof(T);   // elements per 64-byte substep row
  constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
  constexpr int AG = BM / 16, BG = BN / 16;  // 1 KiB groups per slot
  constexpr int AJ = AG / NW, BJ = (BG + NW - 1) / NW;
  constexpr int G = AJ + BJ;                 // DMA instructions per wave per substep
  constexpr int D = R - KSUB;                // prefetch distance in substeps
  constexpr int SLOT = (AG + BG) * 1024;
  constexpr int EST = WN + 4;                // padded fp32 row stride of the epilogue staging
  constexpr int EROWS = (FM >= 2) ? 32 : 16;
  constexpr int EPI_BYTES = NW * EROWS * EST * 4;
  constexpr int SMEM = (R * SLOT > EPI_BYTES) ? R * SLOT : EPI_BYTES;
  static_assert(AG % NW == 0 && WM % 16 == 0 && WN % 16 == 0 && WN % 8 == 0, "tile shape");
  static_assert(D >= KSUB && D >= 1, "ring too shallow");
  static_assert(FM == 1 || FM % 2 == 0, "FM");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // ---- XCD-aware tile id: XCD x (= workgroup id mod 8) owns tiles [x*nt/8, (x+1)*nt/8)
  const int nt = tiles_m * tiles_n;
  if ((int)blockIdx.x >= nt * max(p.splitk, 1)) return;     // launch-tag workgroups (debug_hooks.h)
  int pid = blockIdx.x;
  const int zsplit = pid / nt;
  pid -= zsplit * nt;
  {
    const int q = nt >> 3, r = nt & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (pid / tiles_n) * BM, n0 = (pid % tiles_n) * BN;

  const int taps = (p.mode == GEMM_LINEAR) ? 1 : 9;
  const int cpt = p.K1 / KPB;  // substeps per tap
  const int ks1 = taps * cpt;
  const int ks2 = p.K2 / KPB;
  int kbeg = 0, kend = ks1 + ks2;
  if (p.splitk > 1) {
    const int per = (kend + p.splitk - 1) / p.splitk;
    kbeg = zsplit * per;
    kend = min(kend, kbeg + per);
  }

  // ---- per-lane source rows (fixed across the K loop) ----
  const int lrow = lane >> 2;          // row within a 16-row DMA group
  const int lchk = (lane & 3) * 16;    // 16-byte chunk within the 64-byte substep row
  const char* a1[AJ]; const char* a2[AJ];
  int ab[AJ], ay[AJ], ax[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    int r = m0 + (j * NW + wave) * 16 + lrow;
    r = min(r, p.M - 1);
    a2[j] = p.A2 ? (const char*)p.A2 + ((long)r * p.lda2 + (p.a2_group_n ? (long)(n0 / p.a2_group_n) * p.K2 : 0)) * sizeof(T) + lchk
                 : nullptr;
    if (p.mode == GEMM_LINEAR) {
      a1[j] = (const char*)p.A1 + ((long)r * p.lda1 + (p.a1_group_n ? (long)(n0 / p.a1_group_n) * p.K1 : 0)) * sizeof(T) + lchk;
      ab[j] = ay[j] = ax[j] = 0;
    } else {
      const int ox = r % p.Wout; const int t = r / p.Wout;
      ax[j] = ox; ay[j] = t % p.Hout; ab[j] = t / p.Hout;
      a1[j] = (const char*)p.A1 + lchk;
    }
  }
  const char* w1[BJ]; const char* w2[BJ];
  int bgrp[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    // the last wave(s) re-issue the final group when BG is not a multiple of NW (same bytes, same place):
    // every wave then has exactly G DMA instructions per substep, which the counted vmcnt relies on
    bgrp[j] = min(j * NW + wave, BG - 1);
    int n = n0 + bgrp[j] * 16 + lrow;
    n = min(n, p.N - 1);
    w1[j] = (const char*)p.W1 + ((long)n * p.ldw1) * sizeof(T) + lchk;
    w2[j] = p.W2 ? (const char*)p.W2 + ((long)n * p.ldw2) * sizeof(T) + lchk : nullptr;
  }
  const char* zpage = (const char*)p.zero_page + lchk;

  // NOTE: element-wise selects only -- selecting between the pointer ARRAYS (w1 vs w2) in two
  // branches makes hipcc spill them to scratch and serialise the DMA behind scratch loads.
  auto issue = [&](int kt, int slot) {
    char* As = smem + slot * SLOT;
    char* Bs = As + AG * 1024;
    const bool seg2 = kt >= ks1;
    const long koff = (long)(seg2 ? kt - ks1 : kt) * 64;
    if (seg2 || p.mode == GEMM_LINEAR) {
#pragma unroll
      for (int j = 0; j < AJ; ++j) glds16((seg2 ? a2[j] : a1[j]) + koff, As + (j * NW + wave) * 1024);
    } else {
      const int tap = kt / cpt, cc = kt - tap * cpt;
      const int ky = tap / 3, kx = tap - ky * 3;
      const int sy = (p.mode == GEMM_CONV_S2 || p.mode == GEMM_CONV_S2A) ? 2 : 1;
      const int po = (p.mode == GEMM_CONV_S2A) ? 0 : 1;     // left / top padding
      const bool virt = (p.mode == GEMM_CONV_UP2) | (p.mode == GEMM_CONV_T2);
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
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
        const char* src = ok ? a1[j] + (pix * p.lda1 + (long)cc * KPB) * sizeof(T) : zpage;
        glds16(src, As + (j * NW + wave) * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) glds16((seg2 ? w2[j] : w1[j]) + koff, Bs + bgrp[j] * 1024);
  };

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment addresses: row (lane&15) of a 16-row group, 16-byte chunk (lane>>4)
  const int frag_off = ((lane & 15) * 4 + (lane >> 4)) * 16;
  const uint32_t lds_base = (uint32_t)(uintptr_t)smem;
  const uint32_t a_frag = lds_base + (wm * FM) * 1024 + frag_off;
  const uint32_t b_frag = lds_base + AG * 1024 + (wn * FN) * 1024 + frag_off;

  // Fragment reads are inline asm: hipcc otherwise drains vmcnt(0) in front of every
  // ds_read it can see while an LDS-DMA is in flight, serialising prefetch and MFMA.
  auto read_frags = [&](int slot, u32x4_t (&af)[FM], u32x4_t (&bfr)[FN]) {
    const uint32_t aa = a_frag + slot * SLOT, ba = b_frag + slot * SLOT;
#pragma unroll
    for (int i = 0; i < FM; ++i)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[i]) : "v"(aa), "i"(i * 1024) : "memory");
#pragma unroll
    for (int j = 0; j < FN; ++j)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[j]) : "v"(ba), "i"(j * 1024) : "memory");
  };
  // tie the wait to the registers so the MFMAs cannot be hoisted above it
  auto wait_frags = [&](u32x4_t (&af)[FM], u32x4_t (&bfr)[FN]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(bfr[j]));
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma_all = [&](const u32x4_t (&af)[FM], const u32x4_t (&bfr)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
  };

  // ---- prologue: D substeps in flight
  const int total = kend - kbeg;
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < total) issue(kbeg + s, s % R);

  for (int it = 0; it < total; it += KSUB) {
    // substeps it .. it+KSUB-1 must have landed; (D-KSUB) newer substeps may stay in flight
    if (it + D <= total) wait_vm<(D - KSUB) * G>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // every wave's DMA landed; slots of iteration it-KSUB are free
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < KSUB; ++s) {
      const int kn = it + D + s;
      if (kn < total) issue(kbeg + kn, kn % R);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KSUB == 1) {
      u32x4_t af[FM], bfr[FN];
      read_frags(it % R, af, bfr);
      wait_frags(af, bfr);
      mma_all(af, bfr);
    } else {
      static_assert(KSUB == 2, "KSUB");
      u32x4_t af0[FM], bf0[FN], af1[FM], bf1[FN];
      const bool two = it + 1 < total;
      read_frags(it % R, af0, bf0);
      wait_frags(af0, bf0);
      // second substep's fragment reads are issued before, and land under, the first substep's MFMAs
      // (plain lgkmcnt(0) waits only: a counted lgkmcnt would also count compiler-issued s_loads)
      if (two) read_frags((it + 1) % R, af1, bf1);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(af0, bf0);
      if (two) {
        wait_frags(af1, bf1);
        mma_all(af1, bf1);
      }
    }
  }
  __syncthreads();  // all waves done with the operand slots; reuse LDS for the epilogue

  // ---- epilogue
  store_tile<T, FM, FN, (WGN == 2 && FN == 5)>(p, acc, reinterpret_cast<float*>(smem) + wave * (EROWS * EST),
                                               m0 + wm * WM, n0 + wn * WN, lane, slab, zsplit, wn,
                                               reinterpret_cast<float*>(smem) + (wave ^ 1) * (EROWS * EST));
}

// ---------------------------------------------------------------------------------------------
// Full-line variant (the production path whenever every K segment is a multiple of 128 bytes).
//
// A pipeline stage is 128 bytes of K per row -- a whole cache line per row per DMA, which is what
// the vector-memory path wants (64-byte half-line fetches cost the same address-processing slots) --
// staged 8 rows per global_load_lds instruction.  The LDS image of an instruction is lane-linear
// ([8 rows][128 B]), so the bank-conflict fix is an XOR swizzle applied on the SOURCE side: the lane
// that fills 16-byte slot q of row r fetches logical chunk q ^ ((r >> 1) & 7); the MFMA fragment read
// of (row r, chunk c) then goes to slot c ^ ((r >> 1) & 7).  For the ds_read_b128 lane groups of a
// 16x16x32 fragment (16 rows x chunks {g, g+4}) this spreads every group over all sixteen 16-byte
// bank slots: conflict-free (measured: SQ_LDS_BANK_CONFLICT = 2 % of SQ_LDS_IDX_ACTIVE).
//
// 8 waves (4 x 2) own a 256 x BN tile; 3 LDS slots; one barrier per stage, in the MIDDLE of it:
//     ds_read k-half 1 of stage s | MFMA k-half 0 (operands already in registers) interleaved with the
//     DMA issue of stage s+2 | vmcnt(G): own DMA of stage s+1 landed, stage s+2 stays in flight |
//     s_barrier | ds_read k-half 0 of stage s+1 | MFMA k-half 1
// so a DMA has ~1.5 stages to land, every LDS read is covered by an MFMA batch, and MFMA issue
// resumes straight after the barrier.
//
// The per-stage address generation is the other half of the cost (PMC on the first version: 246
// VALU+SALU instructions per stage per wave against 40 MFMAs), so the kernel is specialised on the
// addressing MODE and everything is incremental: running row pointers (+128 B per stage) for linear
// operands and weights; for the stride-1 3x3 conv a per-lane centre-pixel pointer plus a 9-bit tap
// validity mask computed once, and a wave-uniform (scalar) tap offset updated when the tap changes.
enum { FL_LINEAR = 0, FL_CONV_S1 = 1, FL_CONV_ANY = 2 };

#ifdef FL_TIMING
// probe builds only (tools/probe_gemm.hip): wave 0 of every workgroup stores s_memtime at five points of its tile
__device__ unsigned long long* g_fl_timing = nullptr;
#define FL_STAMP(i) do { if (g_fl_timing && threadIdx.x == 0) { g_fl_timing[(long)vpid * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0) { g_fl_timing[(long)vpid * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    /* HW_ID */ \
                    g_fl_timing[(long)vpid * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 20); } } } while (0)   /* XCC_ID */
void fl_timing_set(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fl_timing), &buf, sizeof(buf)); }
#else
#define FL_STAMP(i) do { } while (0)
#endif

// One output tile (virtual workgroup id `vpid` in [0, tiles * splitk)) of the full-line kernel.
template <typename T, int BM, int BN, int WGM, int WGN, int MODE, int R, int PRIO>
__device__ __forceinline__ void fl_tile(const GemmParams& p, int tiles_m, int tiles_n, float* __restrict__ slab,
                                        int vpid, char* smem) {
  constexpr int NW = WGM * WGN;
  static_assert(R == 2 || R == 3, "ring depth");
  constexpr int KPS = 128 / (int)sizeof(T);  // elements per stage row
  constexpr int WM = BM / WGM, WN = BN / WGN, FM = WM / 16, FN = WN / 16;
  constexpr int AI = BM / 8, BI = BN / 8;    // DMA instructions (8 rows x 128 B) per stage
  constexpr int AJ = AI / NW, BJ = (BI + NW - 1) / NW;
  constexpr int G = AJ + BJ;                 // DMA instructions per wave per stage (uniform: see binst)
  constexpr int SLOT = (AI + BI) * 1024;
  constexpr int EST = WN + 4;
  constexpr int EROWS = (FM >= 2) ? 32 : 16;
  static_assert(AI % NW == 0 && WM % 16 == 0 && WN % 16 == 0, "tile shape");
  static_assert(NW * EROWS * EST * 4 <= R * SLOT, "epilogue staging must fit in the ring");
  FL_STAMP(0);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  const int nt = tiles_m * tiles_n;
  int pid = vpid;
  const int zsplit = pid / nt;
  pid -= zsplit * nt;
  {
    const int q = nt >> 3, r = nt & 7, xcd = pid & 7, idx = pid >> 3;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (pid / tiles_n) * BM, n0 = (pid % tiles_n) * BN;

  const int cpt = p.K1 / KPS;  // stages per tap
  // phase-decomposed UP2 / T2 (gemm.h): this tile's phase, its window of source pixels and its weight block
  int ph_mq = 0, ph = 0, ph_ntx = 3, ph_dy0 = 0, ph_dx0 = 0, ntaps = (MODE == FL_CONV_ANY && p.mode == GEMM_CONV_S2K4) ? 16 : 9;
  long ph_woff = 0;
  if constexpr (MODE == FL_CONV_ANY) {
    if (gemm_phase_mode(p.mode)) {
      ph_mq = p.B * p.Hin * p.Win;
      ph = m0 / ph_mq;
      const int a = ph >> 1, b = ph & 1;
      if (p.mode == GEMM_CONV_UP2P) { ph_ntx = 2; ntaps = 4; ph_dy0 = a - 1; ph_dx0 = b - 1; ph_woff = (long)ph * 4 * p.N * p.K1; }
      else { ph_ntx = 1 + b; ntaps = (1 + a) * (1 + b); ph_woff = (long)p.N * p.K1 * (ph == 0 ? 0 : ph == 1 ? 1 : ph == 2 ? 3 : 5); }
    }
  }
  const long ldw1 = ph_mq ? (long)ntaps * p.K1 : p.ldw1;
  const int ks1 = (MODE == FL_LINEAR ? 1 : ntaps) * cpt;
  const int ks2 = (MODE == FL_LINEAR) ? p.K2 / KPS : 0;
  int kbeg = 0, kend = ks1 + ks2;
  if (p.splitk > 1) {
    const int per = (kend + p.splitk - 1) / p.splitk;
    kbeg = zsplit * per;
    kend = min(kend, kbeg + per);
  }
  const int total = kend - kbeg;

  // ---- per-lane DMA sources: row (lane >> 3) of an 8-row group, source chunk = slot ^ swizzle(row)
  const int lrow = lane >> 3, lslot = lane & 7;
  const char* pa[AJ];            // LINEAR: running pointer.  CONV: centre pixel (S1) / base (ANY)
  const char* a2[AJ];
  uint32_t vmask[AJ];            // CONV_S1: bit t = tap t reads inside the image
  int ab[AJ], ay[AJ], ax[AJ];    // CONV_ANY: output pixel coordinates
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int inst = j * NW + wave;
    const int chunk = (lslot ^ (((inst & 1) << 2) | (lrow >> 1))) * 16;
    int r = m0 + inst * 8 + lrow;
    r = min(r, p.M - 1);
    a2[j] = nullptr; vmask[j] = 0; ab[j] = ay[j] = ax[j] = 0;
    if constexpr (MODE == FL_LINEAR) {
      const bool in2 = kbeg >= ks1;
      a2[j] = p.A2 ? (const char*)p.A2 + ((long)r * p.lda2 + (p.a2_group_n ? (long)(n0 / p.a2_group_n) * p.K2 : 0)) * sizeof(T) + chunk
                   : nullptr;
      pa[j] = in2 ? a2[j] + (long)(kbeg - ks1) * 128
                  : (const char*)p.A1 + ((long)r * p.lda1 + (p.a1_group_n ? (long)(n0 / p.a1_group_n) * p.K1 : 0)) * sizeof(T) +
                        chunk + (long)kbeg * 128;
    } else {
      const int rq = ph_mq ? r - ph * ph_mq : r;            // phase modes: rows live on the SOURCE grid
      const int gw = ph_mq ? p.Win : p.Wout, gh = ph_mq ? p.Hin : p.Hout;
      const int ox = rq % gw; const int t = rq / gw;
      const int oy = t % gh, ob = t / gh;
      if constexpr (MODE == FL_CONV_S1) {
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
  const char* pw[BJ]; const char* w2[BJ];
  int binst[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    // waves past the end re-issue the last group (same bytes, same place): every wave then has exactly
    // G DMA instructions per stage, which the counted vmcnt relies on
    binst[j] = min(j * NW + wave, BI - 1);
    const int chunk = (lslot ^ (((binst[j] & 1) << 2) | (lrow >> 1))) * 16;
    int n = n0 + binst[j] * 8 + lrow;
    n = min(n, p.N - 1);
    w2[j] = (MODE == FL_LINEAR && p.W2) ? (const char*)p.W2 + ((long)n * p.ldw2) * sizeof(T) + chunk : nullptr;
    pw[j] = (MODE == FL_LINEAR && kbeg >= ks1)
                ? w2[j] + (long)(kbeg - ks1) * 128
                : (const char*)p.W1 + (ph_woff + (long)n * ldw1) * sizeof(T) + chunk + (long)kbeg * 128;
  }
  const char* zpage = (const char*)p.zero_page + lslot * 16;

  // wave-uniform walk over (tap, channel chunk) for the conv modes
  int kt_next = kbeg;                                  // next stage to issue
  int tap = (MODE == FL_LINEAR) ? 0 : kbeg / cpt;
  int cc = (MODE == FL_LINEAR) ? 0 : kbeg - tap * cpt;
  const long pixb = (long)p.lda1 * sizeof(T);          // bytes per pixel row of A
  long tapoff = 0;
  const char* cur[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) cur[j] = nullptr;
  if constexpr (MODE == FL_CONV_S1) {
    tapoff = ((long)(tap / 3 - 1) * p.Win + (tap % 3 - 1)) * pixb;
#pragma unroll
    for (int j = 0; j < AJ; ++j)   // a split-K workgroup may start in the middle of a tap
      cur[j] = ((vmask[j] >> tap) & 1u) ? pa[j] + tapoff + (long)cc * 128 : zpage + (long)cc * 128;
  }

  // DMA issue of the next stage, in two parts so that a schedule can put the (expensive) A-operand address
  // generation and the (cheap) B-operand part into different MFMA batches: issue_a first, then issue_b,
  // which also advances the walk.
  auto issue_a = [&](int slot) {
    char* As = smem + slot * SLOT;
    if constexpr (MODE == FL_LINEAR) {
      if (ks2 && kt_next == ks1) {   // switch to the second K segment (LoRA up-projection)
#pragma unroll
        for (int j = 0; j < AJ; ++j) pa[j] = a2[j];
      }
#pragma unroll
      for (int j = 0; j < AJ; ++j) { glds16(pa[j], As + (j * NW + wave) * 1024); pa[j] += 128; }
    } else if constexpr (MODE == FL_CONV_S1) {
      // cur[j] walks the channel chunks of the current tap (+128 B per stage); lanes whose tap falls outside
      // the image walk the zero page instead (it is at least (K1 / 64 + 1) * 128 bytes long), so a DMA costs
      // one 64-bit add.  The select against the 9-bit mask happens only when the tap changes.
      if (cc == 0) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) cur[j] = ((vmask[j] >> tap) & 1u) ? pa[j] + tapoff : zpage;
      }
#pragma unroll
      for (int j = 0; j < AJ; ++j) { glds16(cur[j], As + (j * NW + wave) * 1024); cur[j] += 128; }
    } else {
      const bool k4 = p.mode == GEMM_CONV_S2K4;              // 4 x 4 window
      const int ky = k4 ? tap >> 2 : tap / 3, kx = k4 ? tap & 3 : tap - ky * 3;
      const int sy = (p.mode == GEMM_CONV_S2 || p.mode == GEMM_CONV_S2A || k4) ? 2 : 1;
      const int po = (p.mode == GEMM_CONV_S2A) ? 0 : 1;     // left / top padding
      const bool virt = (p.mode == GEMM_CONV_UP2) | (p.mode == GEMM_CONV_T2);
      const int pty = ph_ntx == 2 ? tap >> 1 : tap, ptx = ph_ntx == 2 ? tap & 1 : 0;   // phase modes: window tap
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int vy = ph_mq ? ay[j] + ph_dy0 + pty : ay[j] * sy + ky - po;
        const int vx = ph_mq ? ax[j] + ph_dx0 + ptx : ax[j] * sy + kx - po;
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
        glds16(src, As + (j * NW + wave) * 1024);
      }
    }
  };
  auto issue_b = [&](int slot) {
    char* Bs = smem + slot * SLOT + AI * 1024;
    if constexpr (MODE == FL_LINEAR) {
      if (ks2 && kt_next == ks1) {
#pragma unroll
        for (int j = 0; j < BJ; ++j) pw[j] = w2[j];
      }
    } else if constexpr (MODE == FL_CONV_S1) {
      const bool wrap = cc + 1 == cpt;          // wave-uniform: scalar selects, no branch
      cc = wrap ? 0 : cc + 1;
      tap += wrap ? 1 : 0;
      const int ky = (tap * 11) >> 5;           // tap / 3 for tap in [0, 9]
      tapoff = ((long)(ky - 1) * p.Win + (tap - 3 * ky - 1)) * pixb;
    } else {
      if (++cc == cpt) { cc = 0; ++tap; }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) { glds16(pw[j], Bs + binst[j] * 1024); pw[j] += 128; }
    ++kt_next;
  };
  auto issue_next = [&](int slot) { issue_a(slot); issue_b(slot); };
  FL_STAMP(1);

  f32x4_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read address: row lr of a 16-row fragment, logical chunk (4*half + g) -> slot ^ (lr >> 1)
  const int lr = lane & 15, g = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)smem;
  const uint32_t frag0 = lr * 128 + ((g ^ (lr >> 1)) * 16);      // k-half 0; k-half 1 = frag0 ^ 64
  const uint32_t a_base = lds_base + (wm * FM) * 2048;
  const uint32_t b_base = lds_base + AI * 1024 + (wn * FN) * 2048;

  auto read_frags = [&](int slot, int half, u32x4_t (&af)[FM], u32x4_t (&bfr)[FN]) {
    const uint32_t off = (half ? (frag0 ^ 64u) : frag0) + slot * SLOT;
    const uint32_t aa = a_base + off, ba = b_base + off;
#pragma unroll
    for (int i = 0; i < FM; ++i)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[i]) : "v"(aa), "i"(i * 2048) : "memory");
#pragma unroll
    for (int j = 0; j < FN; ++j)
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bfr[j]) : "v"(ba), "i"(j * 2048) : "memory");
  };
  auto wait_frags = [&](u32x4_t (&af)[FM], u32x4_t (&bfr)[FN]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(af[i]));
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(bfr[j]));
    __builtin_amdgcn_sched_barrier(0);
  };
  auto mma_all = [&](const u32x4_t (&af)[FM], const u32x4_t (&bfr)[FN]) {
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
  };

  u32x4_t fa[FM], fb[FN], ga[FM], gb[FN];
  int slot = 0;
  if constexpr (PRIO == 4) {
    // ---- coarse ping-pong (R = 3): two sections per stage, L(s) = all fragment reads of stage s + the whole DMA
    // issue of stage s+2 + waits, M(s) = 40 MFMAs; the wave groups one barrier apart as in PRIO 3.
    //   RAW  the vmcnt for stage s (end of L(s-1)) precedes barrier 2s in both groups; L(s) starts after it.
    //   WAR  stage s-1 was last read in L(s-1) (retired before barrier 2s); the DMA into its slot is issued in L(s).
    static_assert(PRIO != 4 || R == 3, "ping-pong needs the 3-slot ring");
    const int grp = wave / (NW / 2);
    issue_next(0);
    if (total > 1) { issue_next(1); wait_vm<G>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    auto stage4 = [&](auto ISSUE) {
      constexpr bool issue = decltype(ISSUE)::value;
      const int slot1 = (slot == R - 1) ? 0 : slot + 1;
      const int slot2 = (slot1 == R - 1) ? 0 : slot1 + 1;
      read_frags(slot, 0, fa, fb);
      read_frags(slot, 1, ga, gb);
      if constexpr (issue) issue_next(slot2);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (issue) wait_vm<G>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i) { asm volatile("" : "+v"(fa[i])); asm volatile("" : "+v"(ga[i])); }
#pragma unroll
      for (int j = 0; j < FN; ++j) { asm volatile("" : "+v"(fb[j])); asm volatile("" : "+v"(gb[j])); }
      __builtin_amdgcn_s_setprio(1);
      mma_all(fa, fb);
      mma_all(ga, gb);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      slot = slot1;
    };
    int s = 0;
    for (; s + 2 < total; ++s) stage4(std::true_type{});
    for (; s < total; ++s) stage4(std::false_type{});
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else if constexpr (PRIO == 3) {
    // ---- ping-pong schedule (R = 3).  Every stage is four barrier-delimited sections per wave,
    //     L0 | M0 | L1 | M1      L = LDS fragment reads + DMA issue + waits, M = 20 bare MFMAs at priority 1
    // and the upper four waves run ONE barrier behind the lower four (each SIMD hosts one wave of either group),
    // so on every SIMD one wave feeds the matrix pipe while the other one loads.
    //   L0(s): read k-half 1 of stage s; issue B-operand DMA of stage s+2; vmcnt: own DMA of stage s+1 landed
    //   L1(s): read k-half 0 of stage s+1; issue A-operand DMA of stage s+3 (into the slot of stage s)
    // Hazards, in barrier generations (group 0 executes L0(s) before barrier 4s+1, group 1 before 4s+2):
    //   RAW  every wave's vmcnt for stage s+1 precedes barrier 4s+2; the earliest L1(s) starts after it.
    //   WAR  the last reads of stage s (L0(s), retired by lgkmcnt(0) before the section's barrier) precede
    //        barrier 4s+2; the earliest DMA into that slot (L1(s) of group 0) is issued after it.
    static_assert(PRIO != 3 || R == 3, "ping-pong needs the 3-slot ring");
    const int grp = wave / (NW / 2);
    issue_next(0);
    if (total > 1) issue_next(1);
    if (total > 2) { issue_a(2); wait_vm<G + AJ>(); } else if (total > 1) { wait_vm<G>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_barrier();
    FL_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0, 0, fa, fb);
    wait_frags(fa, fb);
    if (grp == 1) __builtin_amdgcn_s_barrier();   // the stagger
    __builtin_amdgcn_sched_barrier(0);
    auto stage3 = [&](auto IB_, auto IA_, bool has_next) {
      constexpr bool IB = decltype(IB_)::value, IA = decltype(IA_)::value;
      const int slot1 = (slot == R - 1) ? 0 : slot + 1;
      const int slot2 = (slot1 == R - 1) ? 0 : slot1 + 1;
      // L0
#ifndef FLP_NOLDS
      read_frags(slot, 1, ga, gb);
#endif
      if constexpr (IB) issue_b(slot2);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (IB) wait_vm<G>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // M0
      __builtin_amdgcn_s_setprio(1);
      mma_all(fa, fb);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // L1
#ifndef FLP_NOLDS
      if (has_next) read_frags(slot1, 0, fa, fb);
#endif
      if constexpr (IA) issue_a(slot);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      // M1
#pragma unroll
      for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(ga[i]));
#pragma unroll
      for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(gb[j]));
      __builtin_amdgcn_s_setprio(1);
      mma_all(ga, gb);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(fa[i]));
#pragma unroll
      for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(fb[j]));
      slot = slot1;
    };
    int s = 0;
#ifdef FLP_NODMA
    for (; s + 3 < total; ++s) stage3(std::false_type{}, std::false_type{}, true);
#endif
    for (; s + 3 < total; ++s) stage3(std::true_type{}, std::true_type{}, true);
    for (; s + 2 < total; ++s) stage3(std::true_type{}, std::false_type{}, true);
    for (; s < total; ++s) stage3(std::false_type{}, std::false_type{}, s + 1 < total);
    if (grp == 0) __builtin_amdgcn_s_barrier();   // balance the stagger
  } else if constexpr (PRIO == 2) {
    // ---- split-issue schedule (R = 3): the A-operand DMAs of stage s+2 ride in the first MFMA batch of stage
    // s, the B-operand DMAs in the second one, so both batches carry a similar share of address-generation ALU.
    // (A "late issue" variant -- DMA of stage s+3 issued after the mid-stage barrier, two stages in flight --
    // was measured slower than the baseline schedule: DMA latency is not what limits this kernel.)
    static_assert(PRIO != 2 || R == 3, "split issue needs the 3-slot ring");
    issue_next(0);
    if (total > 1) { issue_next(1); wait_vm<G>(); } else { wait_vm<0>(); }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0, 0, fa, fb);
    wait_frags(fa, fb);
    auto stage2 = [&](auto ISSUE, bool has_next) {
      constexpr bool issue = decltype(ISSUE)::value;
      const int slot1 = (slot == R - 1) ? 0 : slot + 1;
      const int slot2 = (slot1 == R - 1) ? 0 : slot1 + 1;
      read_frags(slot, 1, ga, gb);
      __builtin_amdgcn_sched_barrier(0);
      mma_all(fa, fb);
      if constexpr (issue) issue_a(slot2);                      // slot2 was last read before the previous barrier
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // k-half 1 fragments have landed
      // own DMA of stage s+1 (A and B parts) has landed; only the A part of stage s+2 may still fly
      if constexpr (issue) wait_vm<AJ>(); else wait_vm<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (has_next) read_frags(slot1, 0, fa, fb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(ga[i]));
#pragma unroll
      for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(gb[j]));
      mma_all(ga, gb);
      if constexpr (issue) issue_b(slot2);
      wait_frags(fa, fb);
      slot = slot1;
    };
    int s = 0;
    for (; s + 2 < total; ++s) stage2(std::true_type{}, true);
    for (; s < total; ++s) stage2(std::false_type{}, s + 1 < total);
  } else {
  // ---- prologue: stages 0 and 1 in flight; stage 0 landed, its k-half 0 in registers
  issue_next(0);
  if (total > 1) {
    issue_next(1);
    wait_vm<G>();
  } else {
    wait_vm<0>();
  }
  __builtin_amdgcn_s_barrier();
  FL_STAMP(2);
  __builtin_amdgcn_sched_barrier(0);
  read_frags(0, 0, fa, fb);
  wait_frags(fa, fb);

  // One pipeline stage.  ISSUE (compile time): start the DMA of stage s+2 -- true for all but the last
  // two stages, so that the MFMAs of k-half 0 and the address generation + DMA issue sit in ONE basic
  // block and the scheduler can slot the scalar/vector ALU work into the MFMA issue gaps.
  auto stage = [&](auto ISSUE, bool has_next) {
    constexpr bool issue = decltype(ISSUE)::value;
    const int slot1 = (slot == R - 1) ? 0 : slot + 1;
    const int slot2 = (slot1 == R - 1) ? 0 : slot1 + 1;
    read_frags(slot, 1, ga, gb);
    __builtin_amdgcn_sched_barrier(0);
    mma_all(fa, fb);
    if constexpr (issue && R == 3) issue_next(slot2);          // slot2 was last read before the previous barrier
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // k-half 1 fragments have landed
    if constexpr (issue && R == 3) wait_vm<G>(); else wait_vm<0>();   // own DMA of stage s+1 has landed
    __builtin_amdgcn_s_barrier();                              // ... and everybody else's: stage s+1 visible
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) read_frags(slot1, 0, fa, fb);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FM; ++i) asm volatile("" : "+v"(ga[i]));
#pragma unroll
    for (int j = 0; j < FN; ++j) asm volatile("" : "+v"(gb[j]));
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
    mma_all(ga, gb);
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    // two-slot ring: stage s has been read completely by every wave (barrier above), refill its slot
    if constexpr (issue && R == 2) issue_next(slot);
    wait_frags(fa, fb);
    slot = slot1;
  };
  int s = 0;
  for (; s + 2 < total; ++s) stage(std::true_type{}, true);
  for (; s < total; ++s) stage(std::false_type{}, s + 1 < total);
  }
  FL_STAMP(3);
  __syncthreads();  // all waves done with the operand slots; reuse LDS for the epilogue

  store_tile<T, FM, FN, (WGN == 2 && FN == 5)>(p, acc, reinterpret_cast<float*>(smem) + wave * (EROWS * EST),
                                               m0 + wm * WM, n0 + wn * WN, lane, slab, zsplit, wn,
                                               reinterpret_cast<float*>(smem) + (wave ^ 1) * (EROWS * EST));
  FL_STAMP(4);
#ifdef FL_TIMING
  if (g_fl_timing) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); FL_STAMP(7); }   // (timing builds: the tile's stores have retired)
#endif
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, int R, int PRIO = 0>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void gemm_fl_kernel(GemmParams p, int tiles_m, int tiles_n,
                                                                 float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x >= tiles_m * tiles_n * max(p.splitk, 1)) return;     // launch-tag workgroups (debug_hooks.h)
  fl_tile<T, BM, BN, WGM, WGN, MODE, R, PRIO>(p, tiles_m, tiles_n, slab, (int)blockIdx.x, smem);
}

// Persistent form: gridDim.x workgroups (a multiple of 8, at most what the chip holds at once) walk the virtual
// workgroup ids blockIdx.x, blockIdx.x + gridDim.x, ... -- the XCD of a virtual id is unchanged (id mod 8), and a
// workgroup pays its launch, kernel-argument and first-touch costs once instead of once per tile.  For the
// small-K products (K = 320: five stages per tile) those fixed costs are several times the tile's MFMA time.
template <typename T, int BM, int BN, int WGM, int WGN, int MODE, int R, int PRIO = 0>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void gemm_fl_persist_kernel(GemmParams p, int tiles_m, int tiles_n,
                                                                         float* __restrict__ slab, int nvirt,
                                                                         int stagger) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Equal tiles keep all workgroups in the same phase (everybody loads, then everybody stores).  `stagger` > 0 holds
  // the second half of the grid (with two workgroups per CU: the second resident of every CU) back by that many
  // s_memtime ticks once, so that one resident's stores run under the other's loads from then on.
  if (stagger > 0 && blockIdx.x >= gridDim.x / 2) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)stagger) __builtin_amdgcn_s_sleep(8);
  }
  for (int v = blockIdx.x; v < nvirt; v += gridDim.x) {
    fl_tile<T, BM, BN, WGM, WGN, MODE, R, PRIO>(p, tiles_m, tiles_n, slab, v, smem);
    __syncthreads();   // the epilogue staging aliases the operand ring of the next tile
  }
}

// sum the split-K slabs and apply the epilogue: one lane per 8 output columns
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p, const float* __restrict__ slab, int splits) {
  const EpiArgs e = epi_of(p);
  if (p.act == ACT_GEGLU) {   // value / gate columns interleaved per 160-column tile -> C[M, N/2]
    const int o8 = p.N / 16;
    const long total = (long)p.M * o8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int grow = (int)(i / o8), ocol = (int)(i % o8) * 8;
      const int vcol = (ocol / 80) * 160 + ocol % 80;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int z = 0; z < splits; ++z) {
        float t[8];
        load8(slab + ((long)z * p.M + grow) * p.N + vcol, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
        load8(slab + ((long)z * p.M + grow) * p.N + vcol + 80, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] += t[k];
      }
      if (e.bias) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] += e.bias[vcol + k]; g[k] += e.bias[vcol + 80 + k]; }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= gelu_f(g[k]);
      if (p.out_f32) store8(reinterpret_cast<float*>(e.C) + (long)grow * e.ldc + ocol, v);
      else store8(reinterpret_cast<T*>(e.C) + (long)grow * e.ldc + ocol, v);
    }
    return;
  }
  const int n8 = p.N / 8;
  const long total = (long)p.M * n8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int grow = (int)(i / n8), gcol = (int)(i % n8) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < splits; ++z) {
      float t[8];
      load8(slab + ((long)z * p.M + grow) * p.N + gcol, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    epilogue8<T>(e, v, grow, gcol);
  }
}

// split-K factor imposed on the launch being prepared (0 = the launcher's own rule): set by launch_t from a tuned
// table entry or from the tuner's hook for the duration of one launch (launches are prepared on one host thread at a time)
static thread_local int t_force_sk = 0;

template <typename T, int BM, int BN, int WGM, int WGN, int KSUB, int R>
static int launch_cfg(const GemmParams& p0, hipStream_t stream) {
  constexpr int NW = WGM * WGN;
  constexpr int SLOT = (BM / 16 + BN / 16) * 1024;
  constexpr int WN = BN / WGN, FM = BM / WGM / 16;
  constexpr int EPI = NW * ((FM >= 2) ? 32 : 16) * (WN + 4) * 4;
  constexpr int SMEM = (R * SLOT > EPI) ? R * SLOT : EPI;
  auto kern = &gemm_kernel<T, BM, BN, WGM, WGN, KSUB, R>;
  static bool attr_set = false;
  if (!attr_set) {
    if (SMEM > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return CL_ELAUNCH;
    attr_set = true;
  }
  if ((p0.a1_group_n && p0.a1_group_n % BN) || (p0.a2_group_n && p0.a2_group_n % BN)) return CL_EINVAL;   // a tile would straddle groups
  GemmParams p = p0;
  const int kpb = 64 / (int)sizeof(T);
  const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
  const long tiles = (long)tm * tn;
  const int ksub = ((p.mode == GEMM_LINEAR ? 1 : 9) * p.K1 + p.K2) / kpb;
  float* slab = nullptr;
  if (p.atomic) {
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > ksub) p.splitk = ksub > 0 ? ksub : 1;
  } else {
    // deterministic split-K through the workspace when the tile grid cannot fill the chip
    int sk = 1;
    void* wsp; long wsb;
    ws_for(stream, &wsp, &wsb);
    if (t_force_sk > 0) {   // tuned entry (or the tuner's hook): take the factor as given, within K and the workspace
      sk = wsp ? t_force_sk : 1;
      if (sk > ksub) sk = ksub > 0 ? ksub : 1;
      while (sk > 1 && (long)sk * p.M * p.N * 4 > wsb) --sk;
    } else if (p.splitk <= 1 && tiles < 160 && ksub >= 32 && wsp) {
      sk = (int)((256 + tiles - 1) / tiles);
      const int minsub = p.M <= 64 ? g_tiny_m_minsub : 16;
      if (sk > ksub / minsub) sk = ksub / minsub;
      if (sk > 16) sk = 16;
      while (sk > 1 && (long)sk * p.M * p.N * 4 > wsb) --sk;
      if (sk < 1) sk = 1;
    }
    p.splitk = sk;
    if (sk > 1) slab = reinterpret_cast<float*>(wsp);
  }
  // every split must own at least one substep (the kernel assumes total >= 1 except for empty K)
  if (p.splitk > 1) {
    const int per = (ksub + p.splitk - 1) / p.splitk;
    p.splitk = (ksub + per - 1) / per;
  }
  const long grid = tiles * p.splitk;
  gemm_tag_note(grid, NW * 64);
  hipLaunchKernelGGL(kern, dim3((unsigned)(grid + t_tag)), dim3(NW * 64), SMEM, stream, p, tm, tn, slab);
  if (slab) {
    const long total = (long)p.M * (p.N / 8);
    int rg = (int)((total + 255) / 256); if (rg > 4096) rg = 4096;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(rg), dim3(256), 0, stream, p, slab, p.splitk);
  }
  CL_CHECK_LAUNCH();
  return CL_OK;
}

// choose the split-K factor: `want` workgroups in flight, at least `min_steps` pipeline steps per split
static int pick_splitk(GemmParams& p, long tiles, int steps, int want, int min_steps, float** slab,
                       hipStream_t stream) {
  void* wsp; long wsb;
  ws_for(stream, &wsp, &wsb);
  *slab = nullptr;
  if (p.atomic) {
    if (p.splitk < 1) p.splitk = 1;
    if (p.splitk > steps) p.splitk = steps > 0 ? steps : 1;
  } else {
    int sk = 1;
    if (t_force_sk > 0) {
      sk = wsp ? t_force_sk : 1;
      if (sk > steps) sk = steps > 0 ? steps : 1;
      while (sk > 1 && (long)sk * p.M * p.N * 4 > wsb) --sk;
    } else if (tiles < want && steps >= 2 * min_steps && wsp) {
      sk = (int)((want + tiles - 1) / tiles);
      if (sk > steps / min_steps) sk = steps / min_steps;
      if (sk > 32) sk = 32;
      while (sk > 1 && (long)sk * p.M * p.N * 4 > wsb) --sk;
      if (sk < 1) sk = 1;
    }
    p.splitk = sk;
    if (sk > 1) *slab = reinterpret_cast<float*>(wsp);
  }
  if (p.splitk > 1) {   // every split owns at least one step
    const int per = (steps + p.splitk - 1) / p.splitk;
    p.splitk = (steps + per - 1) / per;
  }
  return p.splitk;
}

// the same choice and the reduce launch for kernels in other translation units (gemm_w4.hip)
int gemm_pick_splitk(GemmParams& p, long tiles, int steps, int want, int min_steps, float** slab, hipStream_t stream) {
  return pick_splitk(p, tiles, steps, want, min_steps, slab, stream);
}
void gemm_launch_splitk_reduce_bf16(const GemmParams& p, const float* slab, hipStream_t stream) {
  const long total = (long)p.M * (p.N / 8);
  int rg = (int)((total + 255) / 256); if (rg > 4096) rg = 4096;
  hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3(rg), dim3(256), 0, stream, p, slab, p.splitk);
}

static int device_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 256;
    n = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  }
  return n;
}

template <typename T, int BM, int BN, int WGM, int WGN, int MODE, int R, int PRIO = 0, bool PERSIST = false>
static int launch_fl_mode(const GemmParams& p0, hipStream_t stream) {
  constexpr int NW = WGM * WGN;
  constexpr int SMEM = R * (BM / 8 + BN / 8) * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    const void* k;
    if constexpr (PERSIST) k = reinterpret_cast<const void*>(&gemm_fl_persist_kernel<T, BM, BN, WGM, WGN, MODE, R, PRIO>);
    else k = reinterpret_cast<const void*>(&gemm_fl_kernel<T, BM, BN, WGM, WGN, MODE, R, PRIO>);
    if (SMEM > 65536 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return CL_ELAUNCH;
    attr_set = true;
  }
  if ((p0.a1_group_n && p0.a1_group_n % BN) || (p0.a2_group_n && p0.a2_group_n % BN)) return CL_EINVAL;   // a tile would straddle groups
  GemmParams p = p0;
  const int kps = 128 / (int)sizeof(T);
  const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
  const long tiles = (long)tm * tn;
  const int steps = ((MODE == FL_LINEAR ? 1 : p.mode == GEMM_CONV_UP2P ? 4 : p.mode == GEMM_CONV_S2K4 ? 16 : 9) * p.K1 + p.K2) / kps;
  float* slab;
  if (gemm_phase_mode(p.mode) && (p.B * p.Hin * p.Win) % BM) return CL_EINVAL;   // a tile lies inside one phase
  if (p.mode == GEMM_CONV_T2P) { p.splitk = 1; slab = nullptr; }   // (its phases are 1 / 2 / 2 / 4 taps deep: no uniform K split)
  else pick_splitk(p, tiles, steps, R == 3 ? (BM == 128 ? g_fl128_split_want : 256) : 512, 4, &slab, stream);
  const long nvirt = tiles * p.splitk;
  if constexpr (PERSIST) {
    // as many workgroups as stay resident together (LDS-bound: one per CU above 80 KB, else two), a multiple of 8
    // so that virtual id mod 8 -- the XCD -- is the same for every tile a workgroup walks
    long grid = (long)device_cus() * (SMEM > 80 * 1024 ? 1 : 2);
    grid -= grid % 8;
    if (grid > nvirt) grid = nvirt;
    if (grid < nvirt) grid -= grid % 8;
    if (grid < 1) grid = nvirt;
    gemm_tag_note(grid, NW * 64);     // (persistent form: the walk tolerates extra workgroups -- they start past nvirt)
    hipLaunchKernelGGL((gemm_fl_persist_kernel<T, BM, BN, WGM, WGN, MODE, R, PRIO>), dim3((unsigned)(grid + t_tag)), dim3(NW * 64),
                       SMEM, stream, p, tm, tn, slab, (int)nvirt, (SMEM <= 80 * 1024 && grid == 2L * device_cus() && nvirt >= 2 * grid) ? g_fl_persist_stagger : 0);
  } else {
    gemm_tag_note(nvirt, NW * 64);
    hipLaunchKernelGGL((gemm_fl_kernel<T, BM, BN, WGM, WGN, MODE, R, PRIO>), dim3((unsigned)(nvirt + t_tag)), dim3(NW * 64), SMEM,
                       stream, p, tm, tn, slab);
  }
  if (slab) {
    const long total = (long)p.M * (p.N / 8);
    int rg = (int)((total + 255) / 256); if (rg > 4096) rg = 4096;
    hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(rg), dim3(256), 0, stream, p, slab, p.splitk);
  }
  CL_CHECK_LAUNCH();
  return CL_OK;
}

template <typename T, int BM, int BN, int WGM, int WGN, int R, int PRIO = 0, bool PERSIST = false>
static int launch_fl(const GemmParams& p, hipStream_t stream) {
  if (p.mode == GEMM_LINEAR) return launch_fl_mode<T, BM, BN, WGM, WGN, FL_LINEAR, R, PRIO, PERSIST>(p, stream);
  if constexpr (PERSIST) return CL_EINVAL;   // persistent form: linear products only (callers check)
  if (p.K2) return CL_EINVAL;   // a second K segment exists for linear operands only
  if (p.mode == GEMM_CONV_S1) return launch_fl_mode<T, BM, BN, WGM, WGN, FL_CONV_S1, R, PRIO>(p, stream);
  return launch_fl_mode<T, BM, BN, WGM, WGN, FL_CONV_ANY, R>(p, stream);   // ping-pong measured slower here (heavy address VALU in the L sections)
}

int g_gemm_force_cfg = -1;   // probe / tuning hook: >= 0 forces a tile configuration
int g_gemm_xs_rules = 1;     // A/B hook (cl_debug_gemm_xs_rules): 0 = the x-stationary kernel only where a table entry names it
int g_fl128_split_want = 128;   // 128-row full-line tiles: split K while the grid is below this many workgroups
int g_tiny_m_minsub = 8;        // fallback kernel, M <= 64: minimum 64-byte substeps per K split

int g_fl_persist_stagger = 0;   // probe hook: s_memtime ticks the second resident workgroup of a CU starts late (persistent 2-per-CU forms)
int g_gemm_force_splitk = 0;    // tuning hook: > 0 imposes the split-K factor (workspace path), 0 = launcher's rule

// Measured launch table (tools/gemm_autotune.py on an MI355X -> ctrlora_amd/gemm_tuned_gfx950.json, loaded by the host
// at start-up): product signature -> (tile configuration, split-K factor).  A signature that is not in the table takes
// the rules below; an entry can only name configurations the switch below accepts, each of which re-checks its own
// preconditions, so a stale or foreign table costs speed, never correctness.
struct TuneKey {
  int v[7];   // dtype, mode, M, N, K1, K2, geglu
  bool operator<(const TuneKey& o) const { return std::lexicographical_compare(v, v + 7, o.v, o.v + 7); }
};
static std::map<TuneKey, std::pair<int, int>> g_tune;

int gemm_tune_set(int dtype, int mode, int M, int N, int K1, int K2, int geglu, int cfg, int splitk) {
  if (cfg < 0 || cfg > 48 || splitk < 0 || splitk > 64) return CL_EINVAL;
  g_tune[TuneKey{{dtype, mode, M, N, K1, K2, geglu ? 1 : 0}}] = std::make_pair(cfg, splitk);
  return CL_OK;
}
void gemm_tune_clear() { g_tune.clear(); }
int gemm_tune_size() { return (int)g_tune.size(); }

template <typename T>
static int launch_t_cfg(const GemmParams& p, hipStream_t stream, int cfg);

template <typename T>
static int launch_t(const GemmParams& p, hipStream_t stream) {
  int cfg = g_gemm_force_cfg, sk = g_gemm_force_splitk;
  if (cfg < 0 && !p.atomic && !g_tune.empty()) {
    const auto it = g_tune.find(TuneKey{{(int)sizeof(T) == 2 ? CL_BF16 : CL_F32, p.mode, p.M, p.N, p.K1, p.K2,
                                         p.act == ACT_GEGLU ? 1 : 0}});
    if (it != g_tune.end()) { cfg = it->second.first; sk = it->second.second; }
  }
  t_force_sk = p.atomic ? 0 : sk;
  const int rc = launch_t_cfg<T>(p, stream, cfg);
  t_force_sk = 0;
  return rc;
}

template <typename T>
static int launch_t_cfg(const GemmParams& p, hipStream_t stream, int cfg) {
  if (p.act == ACT_GEGLU && cfg != 2 && cfg != 12 && cfg != 8 && cfg != 10 && cfg != 14 && cfg != 16 && cfg != 18 && cfg != 20 && cfg != 23 && cfg != 25 && cfg != 27 && cfg != 29 && cfg != 44 && !((cfg == 40 || cfg == 47) && sizeof(T) == 2)) cfg = -2;   // needs a 2 x 80-column wave pair (40: in-register pairing)
  if constexpr (sizeof(T) == 2) {
    // No table entry and no forced configuration: the x-stationary kernel by RULE where the measured table took it at the
    // benchmarked batch sizes (profiles/r05_gemm_xs/autotune_xs.out) -- other batch sizes (pre-training at the reference's
    // batch 4: M = 16384 at the 64x64 level) have no entry of their own.  K = 320: N >= 320 from 16384 rows, N >= 1280 from 8192;
    // K = 640: N >= 5120 from 8192 rows, N >= 640 from 32768.  CL_EINVAL (an epilogue it does not cover) falls through.
    if (cfg == -1 && g_gemm_xs_rules && p.mode == GEMM_LINEAR && p.act == ACT_NONE && !p.rowbias && !p.atomic && !p.out_f32 &&
        !p.a1_group_n && (p.K2 == 0 || p.K2 == 128)) {
      const bool k320 = p.K1 == 320 && p.N >= 320 && (p.M >= 16384 || (p.M >= 8192 && p.N >= 1280));   // (N = 128: the tuner kept the tiles)
      const bool k640 = p.K1 == 640 && ((p.M >= 8192 && p.N >= 5120) || (p.M >= 32768 && p.N >= 640));
      if (k320 || k640) {
        const int rc = launch_gemm_xs(p, stream, 0);
        if (rc != CL_EINVAL) return rc;
      }
    }
  }
  if (cfg < 0) {
    // v2 (64-byte substeps, 4 waves) choices
    if (p.M <= 64 || p.N <= 64) cfg = 0;
    else if (p.N % 160 == 0) cfg = 2;
    else cfg = 1;
    if (cfg != 0 && (long)((p.M + 127) / 128) * ((p.N + 127) / 128) < 48 && p.atomic) cfg = 0;
    // full-line 8-wave kernel: whenever K is whole 128-byte lines and the 256-row tile grid fills the
    // chip, alone or with split-K at >= 8 stages per split (deep-K products of the 8x8 / 16x16 levels)
    const int kps = 128 / (int)sizeof(T);
    const bool fl_ok = p.K1 % kps == 0 && p.K2 % kps == 0 && !(p.mode != GEMM_LINEAR && p.K2) && !p.atomic &&
                       p.M > 128 && p.N >= 96;
    if (fl_ok) {
      const int bn = (p.N % 160 == 0) ? 160 : 128;
      const long t256 = (long)((p.M + 255) / 256) * ((p.N + bn - 1) / bn);
      const int steps = ((p.mode == GEMM_LINEAR ? 1 : 9) * p.K1 + p.K2) / kps;
      const long need = (256 + t256 - 1) / t256;   // split factor that fills 256 CUs
      if (t256 >= 200 || (g_ws && steps >= 8 * need && (p.mode != GEMM_LINEAR || steps > 48)))
        cfg = 16 + (bn == 160 ? 0 : 1);   // ping-pong schedule, 256-row tiles (split-K for the deep-K convs)
      else if (p.mode == GEMM_LINEAR && steps <= 48 && (steps >= 16 || (long)((p.M + 127) / 128) * ((p.N + bn - 1) / bn) >= 128))
        cfg = 20 + (bn == 160 ? 0 : 1);   // mid-size linears (16x16 / 32x32 levels): 128-row tiles fill the chip
    }
    if (p.act == ACT_GEGLU && cfg != 16 && cfg != 20) cfg = 2;
  }
  if (p.a1_group_n || p.a2_group_n) {
    // grouped K segments: the tile width must divide every group width.  BN of the configuration that would actually run
    // (the full-line forms fall back to the generic 128 x 128 tile when K is not whole 128-byte lines)
    const int kps = 128 / (int)sizeof(T);
    const bool lines = p.K1 % kps == 0 && p.K2 % kps == 0;
    auto bn_of = [&](int c) {
      switch (c) {
        case 0: case 24: case 42: case 45: return 64;
        case 43: return 128;
        case 44: return 160;
        case 46: return 128;
        case 1: case 6: case 7: case 22: return 128;
        case 2: case 3: case 4: case 5: case 23: return 160;
        case 40: case 47: return lines ? 160 : 128;
        case 41: case 48: return 128;
        case 31: case 32: case 35: case 36: return (lines && p.N % 80 == 0) ? 80 : 128;
        case 33: return (lines && p.N % 320 == 0) ? 320 : 128;
        case 34: return 32;    // x-stationary kernel: 32-column chunks (its own launcher re-checks the groups)
        default: return lines ? ((c == 8 || c == 12 || c == 14 || c == 16 || c == 18 || c == 20 || c == 10 || c == 25 || c == 27 || c == 29) ? 160 : 128) : 128;
      }
    };
    auto fits = [&](int bn) { return (!p.a1_group_n || p.a1_group_n % bn == 0) && (!p.a2_group_n || p.a2_group_n % bn == 0); };
    if (!fits(bn_of(cfg))) {
      const bool big = p.M > 128 && lines && !p.atomic;
      if (fits(160)) cfg = big ? (p.M >= 8192 ? 16 : 20) : 2;
      else if (fits(128)) cfg = big ? (p.M >= 8192 ? 17 : 21) : 1;
      else if (fits(64)) cfg = 0;
      else return CL_EINVAL;
    }
  }
  switch (cfg) {
    case 0: return launch_cfg<T, 64, 64, 2, 2, 1, 4>(p, stream);
    case 1: return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
    case 2: return launch_cfg<T, 128, 160, 2, 2, 2, 4>(p, stream);
    case 3: return launch_cfg<T, 128, 160, 2, 2, 1, 4>(p, stream);
    case 4: return launch_cfg<T, 128, 160, 2, 2, 2, 6>(p, stream);
    case 5: return launch_cfg<T, 256, 160, 4, 2, 2, 4>(p, stream);
    case 6: return launch_cfg<T, 128, 128, 2, 2, 1, 2>(p, stream);   // the round-0 structure, for A/B
    case 7: return launch_cfg<T, 256, 128, 4, 2, 2, 4>(p, stream);
    case 8: case 9: {
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || (p.mode != GEMM_LINEAR && p.K2))
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 8 ? launch_fl<T, 256, 160, 4, 2, 3>(p, stream) : launch_fl<T, 256, 128, 4, 2, 3>(p, stream);
    }
    case 12: case 13: {   // cfg 8 / 9 with s_setprio(1) around the pure-MFMA half of every stage (+2-3 %)
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || (p.mode != GEMM_LINEAR && p.K2))
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 12 ? launch_fl<T, 256, 160, 4, 2, 3, 1>(p, stream) : launch_fl<T, 256, 128, 4, 2, 3, 1>(p, stream);
    }
    case 14: case 15: {   // split-issue schedule (A-operand DMAs in MFMA batch 1, B-operand DMAs in batch 2)
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || (p.mode != GEMM_LINEAR && p.K2))
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 14 ? launch_fl<T, 256, 160, 4, 2, 3, 2>(p, stream) : launch_fl<T, 256, 128, 4, 2, 3, 2>(p, stream);
    }
    case 16: case 17: {   // production: ping-pong schedule: two wave groups one barrier apart, MFMA sections at priority 1
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || (p.mode != GEMM_LINEAR && p.K2))
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 16 ? launch_fl<T, 256, 160, 4, 2, 3, 3>(p, stream) : launch_fl<T, 256, 128, 4, 2, 3, 3>(p, stream);
    }
    case 18: case 19: {   // coarse ping-pong (one L and one M section per stage)
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || (p.mode != GEMM_LINEAR && p.K2))
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 18 ? launch_fl<T, 256, 160, 4, 2, 3, 4>(p, stream) : launch_fl<T, 256, 128, 4, 2, 3, 4>(p, stream);
    }
    case 20: case 21: {   // ping-pong on 128-row tiles (8 waves of 32 x 80 / 32 x 64): mid-size products
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || (p.mode != GEMM_LINEAR && p.K2))
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 20 ? launch_fl<T, 128, 160, 4, 2, 3, 3>(p, stream) : launch_fl<T, 128, 128, 4, 2, 3, 3>(p, stream);
    }
    case 10: case 11: {   // 128-row tiles, 4 waves, 2-slot ring: two workgroups per CU (small-K / mid-size products)
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || (p.mode != GEMM_LINEAR && p.K2))
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 10 ? launch_fl<T, 128, 160, 2, 2, 2>(p, stream) : launch_fl<T, 128, 128, 2, 2, 2>(p, stream);
    }
    case 25: case 26: case 27: case 28: case 29: case 30: {   // persistent forms of 16 / 17 / 20 / 21 / 10 / 11 (linear products)
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || p.mode != GEMM_LINEAR)
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      switch (cfg) {
        case 25: return launch_fl<T, 256, 160, 4, 2, 3, 3, true>(p, stream);
        case 26: return launch_fl<T, 256, 128, 4, 2, 3, 3, true>(p, stream);
        case 27: return launch_fl<T, 128, 160, 4, 2, 3, 3, true>(p, stream);
        case 28: return launch_fl<T, 128, 128, 4, 2, 3, 3, true>(p, stream);
        case 29: return launch_fl<T, 128, 160, 2, 2, 2, 0, true>(p, stream);
        default: return launch_fl<T, 128, 128, 2, 2, 2, 0, true>(p, stream);
      }
    }
    case 31: case 32: {   // 128 x 80 full-line tiles, 4 waves of 32 x 80 (2- / 3-slot ring): twice the workgroups of the
      // 128 x 160 tile for the M = 2048 / 8192 products of the 16x16 / 32x32 levels (offered to the tuner)
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || p.mode != GEMM_LINEAR || p.N % 80)
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 31 ? launch_fl<T, 128, 80, 4, 1, 2>(p, stream) : launch_fl<T, 128, 80, 4, 1, 3>(p, stream);
    }
    case 35: case 36: {   // 64 x 80 full-line tiles, 4 waves of 16 x 80 (3- / 2-slot ring, 54 / 36 KB: two or three workgroups per CU):
      // the M = 2048 products of the 16x16 level (N = K = 1280) are 256 workgroups of 4 waves as 128 x 80 tiles -- one wave
      // per SIMD, nothing to hide a stage's DMA / barrier latency behind (18 us for 6.7 GFLOP); 512 workgroups here
      // (also the 3x3 convs of the 8x8 level, M = 512: 128 tiles x split-K instead of 16 tiles x 16 splits -- a quarter of the slabs)
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || (p.mode != GEMM_LINEAR && p.K2) || p.N % 80)
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return cfg == 35 ? launch_fl<T, 64, 80, 4, 1, 3>(p, stream) : launch_fl<T, 64, 80, 4, 1, 2>(p, stream);
    }
    case 33: {   // full-N 128 x 320 tiles (8 waves of 64 x 80, 2-slot ring, 112 KB of LDS): the K = 320 products of the 64x64 level
      // (M = 32768: 256 workgroups) read x ONCE instead of once per 160-column tile -- they are bound by bytes, not MFMA
      // (16.3 us against an 8.4 us HBM floor with two tiles per row block); offered to the tuner
      const int kps = 128 / (int)sizeof(T);
      if (p.K1 % kps || p.K2 % kps || p.mode != GEMM_LINEAR || p.N % 320)
        return launch_cfg<T, 128, 128, 2, 2, 2, 4>(p, stream);
      return launch_fl<T, 128, 320, 2, 4, 2>(p, stream);
    }
    case 34: {   // x-stationary streaming kernel (gemm_xs.hip): wide-N / short-K linears, K in {320, 640} (+ 128); the table's
      // split column carries its column-run count.  Anything it does not cover takes the rules above.
      if constexpr (sizeof(T) == 2) {
        const int rc = launch_gemm_xs(p, stream, t_force_sk);
        if (rc != CL_EINVAL) return rc;
      }
      t_force_sk = 0;
      return launch_t_cfg<T>(p, stream, -1);
    }
    case 40: case 41: case 47: case 48: {   // loader / consumer kernel (gemm_w4.hip): 256 x 160 / 256 x 128 tiles, 4 MFMA waves + 4 DMA
      // waves; 47 / 48: persistent (one workgroup per CU walks the tiles: the next tile's loads and this tile's stores overlap)
      if constexpr (sizeof(T) == 2) {
        const int rc = launch_gemm_w4(p, stream, (cfg == 40 || cfg == 47) ? 160 : 128, cfg >= 47);
        if (rc != CL_EINVAL) return rc;
      }
      return launch_t_cfg<T>(p, stream, -1);
    }
    // Round 6: the same small tiles with an 8-slot ring (7 substeps of LDS-DMA in flight instead of 3).  The short-K / small-M
    // launches run a near-constant ~1 us per pipeline step whatever their size: an LDS-DMA round trip under load is ~2000 cycles
    // (profiles/r06_w4/), and a step can only be as short as round trip / (slots - 1).  Offered to the tuner.
    case 42: return launch_cfg<T, 64, 64, 2, 2, 1, 8>(p, stream);
    case 43: return launch_cfg<T, 64, 128, 2, 2, 1, 8>(p, stream);
    case 44: return launch_cfg<T, 64, 160, 2, 2, 1, 8>(p, stream);
    case 45: return launch_cfg<T, 128, 64, 2, 2, 1, 8>(p, stream);
    case 46: return launch_cfg<T, 128, 128, 2, 2, 2, 8>(p, stream);
    // small-M tiles of the generic kernel (8x8 / 16x16 levels, text-context projections): offered to the tuner
    case 22: return launch_cfg<T, 64, 128, 2, 2, 1, 4>(p, stream);
    case 23: return launch_cfg<T, 64, 160, 2, 2, 1, 4>(p, stream);
    case 24: return launch_cfg<T, 128, 64, 2, 2, 1, 4>(p, stream);
    default: return CL_EINVAL;
  }
}

int launch_gemm(const GemmParams& p, int dtype, hipStream_t stream) {
  const int kpb = dtype == CL_BF16 ? 32 : 16;
  if (p.M <= 0 || p.N <= 0) return CL_OK;
  if (p.K1 % kpb || p.K2 % kpb || p.N % 8 || p.ldc % 8) return CL_EINVAL;
  if (p.K1 <= 0) return CL_EINVAL;
  if (p.mode != GEMM_LINEAR && !p.zero_page) return CL_EINVAL;
  if (p.K2 && (!p.A2 || !p.W2)) return CL_EINVAL;
  if (p.atomic == 0 && p.splitk > 1) return CL_EINVAL;
  if (p.rowbias && p.rows_per_batch <= 0) return CL_EINVAL;
  if (p.act == ACT_GEGLU && (p.N % 160 || p.rowbias || p.residual || p.atomic || p.alpha != 1.0f)) return CL_EINVAL;
  t_tag = tag_for(p, dtype);
  if (gemm_phase_mode(p.mode) || p.mode == GEMM_CONV_S2K4) {
    // phase-decomposed UP2 / T2 and the 4x4 stride-2 window (gemm.h): the full-line kernel's generic-conv form only;
    // g_gemm_force_cfg 8 / 9 / 10 / 11 impose 256 x {160, 128} / 128 x {160, 128} tiles.
    const int kps = dtype == CL_BF16 ? 64 : 32;
    const long mq = (long)p.B * p.Hin * p.Win;
    if (p.K1 % kps || p.K2 || p.atomic || p.act == ACT_GEGLU || p.a1_group_n || p.a2_group_n || p.N < 96) return CL_EINVAL;
    if (p.mode == GEMM_CONV_S2K4) {
      if (p.Hin != 2 * p.Hout || p.Win != 2 * p.Wout || p.M != (long)p.B * p.Hout * p.Wout || p.ldw1 != 16L * p.K1) return CL_EINVAL;
    } else if (p.M != 4 * mq || mq % 128 || p.Hout != 2 * p.Hin || p.Wout != 2 * p.Win) {
      return CL_EINVAL;
    }
    const int bn = (p.N % 160 == 0) ? 160 : 128;
    // 128-row tiles (two workgroups per CU, split-K from the launcher's rule) measured 3-12 % ahead of 256-row tiles at every
    // production shape of all three modes (tools/time_conv_phase.py, profiles/r06_phase/time_conv_phase.log)
    bool big = false;
    if (g_gemm_force_cfg == 8 || g_gemm_force_cfg == 9) big = p.mode == GEMM_CONV_S2K4 || mq % 256 == 0;
    if (g_gemm_force_cfg == 10 || g_gemm_force_cfg == 11) big = false;
    const bool n160 = g_gemm_force_cfg >= 8 && g_gemm_force_cfg <= 11 ? (g_gemm_force_cfg % 2 == 0 && p.N % 160 == 0) : bn == 160;
    t_force_sk = g_gemm_force_splitk;
    int rc;
    if (dtype == CL_BF16) {
      rc = big ? (n160 ? launch_fl<bf16_t, 256, 160, 4, 2, 3>(p, stream) : launch_fl<bf16_t, 256, 128, 4, 2, 3>(p, stream))
               : (n160 ? launch_fl<bf16_t, 128, 160, 2, 2, 2>(p, stream) : launch_fl<bf16_t, 128, 128, 2, 2, 2>(p, stream));
    } else {
      rc = big ? (n160 ? launch_fl<float, 256, 160, 4, 2, 3>(p, stream) : launch_fl<float, 256, 128, 4, 2, 3>(p, stream))
               : (n160 ? launch_fl<float, 128, 160, 2, 2, 2>(p, stream) : launch_fl<float, 128, 128, 2, 2, 2>(p, stream));
    }
    t_force_sk = 0;
    return rc;
  }
  if (p.act == ACT_GEGLU_SPLIT || p.ln_gamma)   // natural-order GEGLU rows / LayerNorm prologue: the x-stationary kernel only (gemm_xs.hip)
    return dtype == CL_BF16 ? launch_gemm_xs(p, stream, g_gemm_force_splitk) : CL_EINVAL;
  if (p.a1_group_n < 0 || p.a2_group_n < 0) return CL_EINVAL;
  if (p.alpha_n < 0 || p.alpha_n % 8 || p.alpha_n > p.N || (p.alpha_n && p.act == ACT_GEGLU)) return CL_EINVAL;
  if ((p.a1_group_n || p.a2_group_n) && p.mode != GEMM_LINEAR) return CL_EINVAL;
  if (p.a1_group_n && p.N % p.a1_group_n) return CL_EINVAL;
  if (p.a2_group_n && (p.N % p.a2_group_n || !p.K2)) return CL_EINVAL;
  return dtype == CL_BF16 ? launch_t<bf16_t>(p, stream) : launch_t<float>(p, stream);
}

}  // namespace cl
