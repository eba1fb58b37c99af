// Transpose-free weight gradient for gfx950:
//
//   dW[n, k] (fp32, accumulated) += alpha * sum_m dy[m, n] * x[m, k]
//
// for the optimizer's matrices only -- LoRA down/up (cldm/lora.py:26-80: dA = (dy B)^T x,
// dB = dy^T (x A^T)) and the ControlNet zero convs (cldm/cldm.py:281-282) -- frozen weights never
// get a dW.  Both operands are row-major with the contraction index m as the
// ROW, which is the wrong way round for an MFMA fragment (a lane needs 8 consecutive m of one
// column).  Round 0 materialised dy^T and x^T in HBM first (4 transposes per LoRA linear: 7.6 ms of a
// 74 ms step); here the [32 m][128 col] tiles go HBM->LDS as they are (global_load_lds, whole 256-byte
// rows) and the fragments are built by ds_read_b64_tr_b16, the gfx950 LDS transpose read: within a
// 16-lane group, lane i receives element (i & 3) of the 8 bytes addressed by lane 4j + (i >> 2), for
// j = 0..3 -- i.e. with lanes 4j..4j+3 pointing at 16 consecutive columns of row k0 + j, lane i gets
// column i of rows k0..k0+3.  Two such reads (rows +0, +4) make one 8-deep MFMA operand.
//
// Bank conflicts: a 256-byte row stride puts every row on the same banks; the 16-byte chunk index is
// XOR-swizzled on the DMA source side with f(row) = 2 * ((row & 3) | ((row >> 3) & 1) << 2), which
// spreads the 8 rows x 2 chunks touched by a 32-lane group over all sixteen slots.
//
// The m range is split across workgroups (grid.y); each split stores its fp32 partial tile into a slab
// of the host-provided workspace and a small second kernel adds the slab sum into the flat gradient
// buffer.  (First version: fp32 atomics -- 16K per workgroup on the same few addresses -- measured
// 77 us for a 29 MB problem, time proportional to the number of workgroups.)  Deterministic as a bonus.
#include "gemm.h"
#include "mma.h"
#include <algorithm>

namespace cl {

__device__ __forceinline__ u32x2_t lds_read_tr16(uint32_t addr) {
  u32x2_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// Many weight gradients per launch: the LoRA matrices are small (r x K, N x r), so one problem cannot fill
// 256 CUs without splitting m very finely; instead the engine queues the problems of a transformer block /
// ResBlock and ONE launch covers them all (descriptor table passed by value in the kernel arguments, so the
// launch is hipGraph-replayable without any host->device copy).
struct WgradProb {
  const bf16_t* dy; const bf16_t* x; float* dW; float* slab;   // slab: this problem's partial-tile region or null
  long lddy, ldx, lddw;
  int M, N, K;
  float alpha;
  int tiles_k, tiles, per, splits;   // k-tiles per row of tiles, tiles, 32-row steps per split, splits
  int blk0;                          // first workgroup id of this problem
  int red0;                          // first reduce-kernel workgroup id of this problem
  int tap;                           // >= 0: 3x3-conv tap mode (x rows gathered through the conv geometry)
  short Hin, Win, Hout, Wout;        // conv geometry (<= 32767)
  short stride, pad;
};
constexpr int WGRAD_MAX_PROBS = 24;
struct WgradGroup { int n; int pad; WgradProb p[WGRAD_MAX_PROBS]; };

// 4 waves (2 x 2), 128 (n) x 128 (k) output tile, ROWS (32 or 64) rows of m per pipeline step, R-slot ring
// (R - 1 steps of DMA in flight).  ROWS = 64 (round 5): two 32-deep MFMA batches per barrier -- at 32 rows a step is 16 MFMAs per
// wave (256 matrix-pipe cycles) behind a barrier, a counted vmcnt and the LDS transpose reads' latency.
template <int R, int ROWS = 32>
__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(const WgradGroup grp, const void* __restrict__ zero_page) {
  constexpr int TILE = ROWS * 256;        // one operand tile: ROWS rows x 256 bytes
  constexpr int SLOT = 2 * TILE;
  constexpr int NI = ROWS / 16;           // DMA instructions (4 rows each) per wave and operand tile
  static_assert(R * SLOT >= 4 * 32 * 68 * 4, "epilogue staging must fit in the ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // ---- which problem is this workgroup in (wave-uniform scan over the by-value table)
  int pi = 0;
  while (pi + 1 < grp.n && (int)blockIdx.x >= grp.p[pi + 1].blk0) ++pi;
  const WgradProb& P = grp.p[pi];
  const bf16_t* __restrict__ dy = P.dy; const bf16_t* __restrict__ x = P.x;
  float* __restrict__ dW = P.dW; float* __restrict__ slab = P.slab;
  const long lddy = P.lddy, ldx = P.ldx, lddw = P.lddw;
  const int M = P.M, N = P.N, K = P.K, tiles_k = P.tiles_k, steps_per_split = P.per;
  const float alpha = P.alpha;
  const int local = (int)blockIdx.x - P.blk0;
  const int tile = local % P.tiles, split = local / P.tiles;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
  const int steps_total = (M + ROWS - 1) / ROWS;
  const int sbeg = split * steps_per_split;
  const int send = min(steps_total, sbeg + steps_per_split);
  const int total = send - sbeg;
  if (total <= 0) return;

  // ---- DMA sources: instruction i of a tile covers rows 4i..4i+3; lane -> (row 4i + lane/16, slot lane%16)
  // wave w issues instructions 2w, 2w+1 of each operand tile.
  const int lr4 = lane >> 4, lslot = lane & 15;
  const char* zsrc = (const char*)zero_page + (lslot & 3) * 16;
  int rowi[NI]; const char* sdy[NI]; const char* sx[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int row = (NI * wave + j) * 4 + lr4;                        // 0..ROWS-1 within the step
    const int f = ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
    const int chunk = lslot ^ f;                                      // logical 16-byte chunk of the row
    rowi[j] = row;
    // columns past the matrix edge: any valid bytes do (those outputs are never stored)
    const int cn = min(n0 + chunk * 8, N - 8), ck = min(k0 + chunk * 8, K - 8);
    sdy[j] = (const char*)(dy + cn);
    sx[j] = (const char*)(x + ck);
  }
  auto issue = [&](int step, int slot) {
    char* base = smem + slot * SLOT;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const long m = (long)step * ROWS + rowi[j];
      const bool ok = m < M;                                          // rows past M contribute zeros
      glds16(ok ? sdy[j] + m * lddy * 2 : zsrc, base + (NI * wave + j) * 1024);
      if (P.tap < 0) {                                                 // (wave-uniform)
        glds16(ok ? sx[j] + m * ldx * 2 : zsrc, base + TILE + (NI * wave + j) * 1024);
      } else {   // conv tap: dy row (b, oy, ox) pairs with input pixel (oy s + ky - pad, ox s + kx - pad)
        const int mi = (int)m, ox = mi % P.Wout, t2 = mi / P.Wout, oy = t2 % P.Hout, ob = t2 / P.Hout;
        const int ky = P.tap / 3, kx = P.tap - 3 * ky;
        const int iy = oy * P.stride + ky - P.pad, ix = ox * P.stride + kx - P.pad;
        const bool okx = ok & ((unsigned)iy < (unsigned)P.Hin) & ((unsigned)ix < (unsigned)P.Win);
        const long xr = ((long)ob * P.Hin + iy) * P.Win + ix;
        glds16(okx ? sx[j] + xr * ldx * 2 : zsrc, base + TILE + (NI * wave + j) * 1024);
      }
    }
  };

  // ---- fragment read addresses.  lane = 16g + 4j + q: row 8g + j (+4 for the second read),
  // columns 4q..4q+3 of the fragment's 16 -> chunk (2*frag + q/2) ^ f(row), byte (q & 1) * 8
  const int g = lane >> 4, jj = (lane >> 2) & 3, q = lane & 3;
  const int frow = 8 * g + jj;
  const int fsw = (jj | ((g & 1) << 2)) << 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t aoff[4], boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    aoff[i] = frow * 256 + ((((wn * 8 + 2 * i) + (q >> 1)) ^ fsw) * 16) + (q & 1) * 8;
    boff[i] = TILE + frow * 256 + ((((wk * 8 + 2 * i) + (q >> 1)) ^ fsw) * 16) + (q & 1) * 8;
  }

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < R - 1; ++s)
    if (s < total) issue(sbeg + s, s);
  for (int s = 0; s < total; ++s) {
    // own DMA of step s landed; R-2 newer steps (2 NI instructions per wave each) may stay in flight
    if (s + R - 1 <= total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 2 * NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // step s visible to all; slot (s-1)%R no longer read
    __builtin_amdgcn_sched_barrier(0);
    if (s + R - 1 < total) issue(sbeg + s + R - 1, (s + R - 1) % R);
    const uint32_t base = lds0 + (s % R) * SLOT;
#pragma unroll
    for (int sub = 0; sub < ROWS / 32; ++sub) {      // one 32-deep MFMA batch per 32 rows of the step
      const uint32_t sb = base + sub * (32 * 256);
      u32x4_t af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x2_t lo = lds_read_tr16(sb + aoff[i]), hi = lds_read_tr16(sb + aoff[i] + 1024);
        af[i] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x2_t lo = lds_read_tr16(sb + boff[i]), hi = lds_read_tr16(sb + boff[i] + 1024);
        bfr[i] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(af[i])); asm volatile("" : "+v"(bfr[i])); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<bf16_t>::run(af[i], bfr[j], acc[i][j]);
    }
  }
  __syncthreads();

  // ---- epilogue: stage 32 x 64 fp32 per wave through LDS, accumulate rows with fp32 atomics
  constexpr int EST = 68;
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * EST);
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          stg[(i2 * 16 + (lane >> 4) * 4 + r) * EST + j * 16 + (lane & 15)] = acc[ps * 2 + i2][j][r];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int rr = it * 4 + (lane >> 4);
      const int cg = (lane & 15) * 4;
      const int grow = n0 + wn * 64 + ps * 32 + rr, gcol = k0 + wk * 64 + cg;
      if (grow < N && gcol < K) {
        const float4 v = *reinterpret_cast<const float4*>(&stg[rr * EST + cg]);
        if (slab) {   // one fp32 partial slab per m-split; summed by wgrad_reduce_kernel
          *reinterpret_cast<float4*>(slab + ((long)split * N + grow) * K + gcol) = v;
        } else {      // single split: this workgroup is the only writer of its tile
          float4* dst = reinterpret_cast<float4*>(dW + (long)grow * lddw + gcol);
          float4 o = *dst;
          o.x += v.x * alpha; o.y += v.y * alpha; o.z += v.z * alpha; o.w += v.w * alpha;
          *dst = o;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The three taps of ONE kernel row of a stride-1, pad-1 3x3 conv per workgroup (descriptor tap = 16 + ky; Base-ControlNet
// pre-training, cldm/cldm_ctrlora_pretrain.py:174-182).  As nine single-tap problems the weight gradient of a conv reads dy
// and x nine times (378 MB of L2 -> LDS traffic for 320 -> 320 at 64x64, where one pass is 42 MB).  Here a 32-row step
// loads its dy tile ONCE and ONE x tile that holds the pixels of all three taps kx = 0, 1, 2 -- the step's pixels plus one
// halo pixel on either side of every image-row segment -- and runs three MFMA batches off it, each reading the x
// fragments one LDS row further on: a third of the dy traffic, 1.1x instead of 3x the x traffic, the dy fragment reads
// shared by three batches.  8 waves (2 x 4) own a 128 (n) x 128 (k) tile of each of the three taps (wave: 64 x 32 per tap,
// 96 accumulator registers); x tile = 40 LDS rows of 256 bytes, row of (step row r, tap kx) = (r / seg) (seg + 2) + r % seg + kx
// with seg = min(W, 32) (40 rows at most; a step is 32 / seg whole image-row segments: W a multiple of 32, or 32 a multiple of W).
// dW points at tap (ky, 0) of the [N][3][3][K] gradient: tap kx lies K floats further on.
template <int R>
__global__ __launch_bounds__(512, 2) void wgrad_row3_kernel(const WgradGroup grp, const void* __restrict__ zero_page) {
  constexpr int DYT = 32 * 256;            // dy tile: 32 rows x 256 bytes
  constexpr int XT = 40 * 256;             // x tile: 40 rows (34 / 36 / 40 used at W >= 32 / 16 / 8): ten 4-row DMA instructions
  constexpr int SLOT = DYT + XT;
  static_assert(R * SLOT >= 8 * 32 * 36 * 4, "epilogue staging must fit in the ring");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int pi = 0;
  while (pi + 1 < grp.n && (int)blockIdx.x >= grp.p[pi + 1].blk0) ++pi;
  const WgradProb& P = grp.p[pi];
  const bf16_t* __restrict__ dy = P.dy; const bf16_t* __restrict__ x = P.x;
  float* __restrict__ dW = P.dW; float* __restrict__ slab = P.slab;
  const long lddy = P.lddy, ldx = P.ldx, lddw = P.lddw;
  const int M = P.M, N = P.N, K = P.K, tiles_k = P.tiles_k, steps_per_split = P.per;
  const float alpha = P.alpha;
  const int local = (int)blockIdx.x - P.blk0;
  const int tile = local % P.tiles, split = local / P.tiles;
  const int Wd = P.Wout, Hd = P.Hout, ky = P.tap - 16;
  const int seg = Wd < 32 ? Wd : 32, nseg = 32 / seg, rps = seg + 2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 128;
  const int steps_total = (M + 31) / 32;
  const int sbeg = split * steps_per_split;
  const int send = min(steps_total, sbeg + steps_per_split);
  const int total = send - sbeg;
  if (total <= 0) return;

  auto swz = [](int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; };
  // ---- DMA sources.  dy: instruction `wave` (rows 4 wave .. 4 wave + 3); x: instruction `wave`, and 8 + wave for waves 0, 1
  // (ten instructions = 40 rows; waves 0 and 1 count three DMA instructions per step, the others two)
  const int lr4 = lane >> 4, lslot = lane & 15;
  const char* zsrc = (const char*)zero_page + (lslot & 3) * 16;
  const int dyrow = 4 * wave + lr4;
  const char* sdy = (const char*)(dy + min(n0 + (lslot ^ swz(dyrow)) * 8, N - 8));
  int xs_[2], xj_[2]; bool xq_[2]; const char* sx[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = (wave + 8 * j) * 4 + lr4;                            // x-tile row (j = 1: waves 0, 1 only)
    xs_[j] = q / rps; xj_[j] = q - xs_[j] * rps; xq_[j] = xs_[j] < nseg;
    sx[j] = (const char*)(x + min(k0 + (lslot ^ swz(q)) * 8, K - 8));
  }
  auto issue = [&](int step, int slot) {
    char* base = smem + slot * SLOT;
    const long m = (long)step * 32 + dyrow;
    glds16(m < M ? sdy + m * lddy * 2 : zsrc, base + wave * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && wave >= 2) break;                                   // (wave-uniform)
      const int mc = step * 32 + xs_[j] * seg;                         // first pixel of this row's segment
      const int ox0 = mc % Wd, t2 = mc / Wd, oy = t2 % Hd, ob = t2 / Hd;
      const int iy = oy + ky - 1, ix = ox0 - 1 + xj_[j];
      const bool ok = xq_[j] & (mc < M) & ((unsigned)iy < (unsigned)Hd) & ((unsigned)ix < (unsigned)Wd);
      const long xr = ((long)ob * Hd + iy) * Wd + ix;
      glds16(ok ? sx[j] + xr * ldx * 2 : zsrc, base + DYT + (wave + 8 * j) * 1024);
    }
  };

  // ---- fragment read addresses (as in wgrad_tn_kernel: lane = 16 g + 4 jj + q reads row 8 g + jj, and + 4 for the second half)
  const int g = lane >> 4, jj = (lane >> 2) & 3, q = lane & 3;
  const int frow = 8 * g + jj;
  const int fsw = (jj | ((g & 1) << 2)) << 1;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  uint32_t aoff[4], boff[3][2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = frow * 256 + ((((wn * 8 + 2 * i) + (q >> 1)) ^ fsw) * 16) + (q & 1) * 8;
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = frow + 4 * h;
        const int lrow = (r / seg) * rps + r % seg + t;
        boff[t][i][h] = DYT + lrow * 256 + ((((wk * 4 + 2 * i) + (q >> 1)) ^ swz(lrow)) * 16) + (q & 1) * 8;
      }

  f32x4_t acc[3][4][2];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < R - 1; ++s)
    if (s < total) issue(sbeg + s, s);
  for (int s = 0; s < total; ++s) {
    // own DMA of step s landed; R - 2 newer steps stay in flight: 3 DMA instructions per step in waves 0 and 1, 2 in the others
    if (s + R - 1 <= total) {
      if (wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 3) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 2) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (s + R - 1 < total) issue(sbeg + s + R - 1, (s + R - 1) % R);
    const uint32_t sb = lds0 + (s % R) * SLOT;
    u32x4_t af[4], bfr[3][2];
    auto read_b = [&](int t) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x2_t lo = lds_read_tr16(sb + boff[t][i][0]), hi = lds_read_tr16(sb + boff[t][i][1]);
        bfr[t][i] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
      }
    };
    auto landed_b = [&](int t) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(bfr[t][i]));
    };
    auto mma_tap = [&](int t) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) Mma<bf16_t>::run(af[i], bfr[t][j], acc[t][i][j]);
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x2_t lo = lds_read_tr16(sb + aoff[i]), hi = lds_read_tr16(sb + aoff[i] + 1024);
      af[i] = u32x4_t{lo.x, lo.y, hi.x, hi.y};
    }
    read_b(0);
    landed_b(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(af[i]));
    __builtin_amdgcn_sched_barrier(0);
    read_b(1);                       // the next tap's fragments land under this tap's eight MFMAs
    __builtin_amdgcn_sched_barrier(0);
    mma_tap(0);
    __builtin_amdgcn_sched_barrier(0);
    landed_b(1);
    read_b(2);
    __builtin_amdgcn_sched_barrier(0);
    mma_tap(1);
    __builtin_amdgcn_sched_barrier(0);
    landed_b(2);
    __builtin_amdgcn_sched_barrier(0);
    mma_tap(2);
  }
  __syncthreads();

  // ---- epilogue: per tap, 32 (n) x 32 (k) fp32 per wave through LDS -> slab [split][tap][N][K] or dW (+= alpha .)
  constexpr int EST = 36;
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * EST);
#pragma unroll
  for (int t = 0; t < 3; ++t) {
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stg[(i2 * 16 + (lane >> 4) * 4 + r) * EST + j * 16 + (lane & 15)] = acc[t][ps * 2 + i2][j][r];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int rr = it * 8 + (lane >> 3);
        const int cg = (lane & 7) * 4;
        const int grow = n0 + wn * 64 + ps * 32 + rr, gcol = k0 + wk * 32 + cg;
        if (grow < N && gcol < K) {
          const float4 v = *reinterpret_cast<const float4*>(&stg[rr * EST + cg]);
          if (slab) {
            *reinterpret_cast<float4*>(slab + (((long)split * 3 + t) * N + grow) * K + gcol) = v;
          } else {
            float4* dst = reinterpret_cast<float4*>(dW + (long)grow * lddw + (long)t * K + gcol);
            float4 o = *dst;
            o.x += v.x * alpha; o.y += v.y * alpha; o.z += v.z * alpha; o.w += v.w * alpha;
            *dst = o;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// dW += alpha * sum over splits of the partial slabs (all problems of the group in one launch;
// one workgroup = 256 float4 of one problem)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradGroup grp) {
  int pi = 0;
  while (pi + 1 < grp.n && (int)blockIdx.x >= grp.p[pi + 1].red0) ++pi;
  const WgradProb& P = grp.p[pi];
  if (!P.slab) return;
  const int NT = P.tap >= 16 ? 3 : 1;          // row-of-three-taps problems: slabs are [split][tap][N][K]
  const int N = P.N, K = P.K, k4 = K / 4;
  const long total = (long)NT * N * k4;
  const long i = (long)((int)blockIdx.x - P.red0) * 256 + threadIdx.x;
  if (i >= total) return;
  const int nn = (int)(i / k4), k = (int)(i % k4) * 4;
  const int t = nn / N, n = nn - t * N;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int z = 0; z < P.splits; ++z) {
    const float4 v = *reinterpret_cast<const float4*>(P.slab + ((long)z * NT * N + nn) * K + k);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  float4* dst = reinterpret_cast<float4*>(P.dW + (long)n * P.lddw + (long)t * K + k);
  float4 o = *dst;
  o.x += a.x * P.alpha; o.y += a.y * P.alpha; o.z += a.z * P.alpha; o.w += a.w * P.alpha;
  *dst = o;
}

// tuning knobs (probe): workgroups wanted per GROUP, minimum 32-row steps per split, ring depth, rows of m per step.
// Round 5 (profiles/r05_final/probe_wgrad_forms.log): ring 3 (48 KB: THREE workgroups per CU) instead of 4 (64 KB: two) --
// the nine-tap groups of pre-training 284 -> 239 us at 320 -> 320 / 64x64, 186 -> 152 at 640 -> 640 / 32x32; 64-row steps
// (two MFMA batches per barrier; 2-slot ring to keep two workgroups per CU) measured no better than 32-row steps: 277 / 181 us.
int g_wgrad_blocks = 512, g_wgrad_min_steps = 8, g_wgrad_ring = 3, g_wgrad_rows = 32;
int g_wgrad_row3_blocks = 512;

static int launch_group(WgradGroup& grp, int nblocks, int nred, const void* zero_page, hipStream_t stream, bool row3 = false) {
  if (row3) {   // groups of row-of-three-taps problems (tap = 16 + ky): their own kernel, 8 waves, 72 KB ring
    static bool attr_set = false;
    // one 8-wave workgroup per CU (178 registers): the ring is what keeps DMA in flight -- 8 slots of 18 KB = seven steps ahead
    // (with 3 slots the kernel ran at the L2 latency of two steps: 202 us for the 320 -> 320 conv at 64x64)
    constexpr int RR = 8;
    constexpr int LDSB = RR * (32 * 256 + 40 * 256);
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_row3_kernel<RR>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              LDSB) != hipSuccess)
        return CL_ELAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL((wgrad_row3_kernel<RR>), dim3(nblocks), dim3(512), LDSB, stream, grp, zero_page);
    if (nred > 0) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nred), dim3(256), 0, stream, grp);
    CL_CHECK_LAUNCH();
    return CL_OK;
  }
#define WGRAD_LAUNCH(RR, ROWS)                                                                                          \
  do {                                                                                                                  \
    static bool attr_set = false;                                                                                       \
    constexpr int LDSB = RR * 2 * ROWS * 256;                                                                           \
    if (!attr_set) {                                                                                                    \
      if (LDSB > 65536 &&                                                                                               \
          hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tn_kernel<RR, ROWS>),                                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDSB) != hipSuccess)                          \
        return CL_ELAUNCH;                                                                                              \
      attr_set = true;                                                                                                  \
    }                                                                                                                   \
    hipLaunchKernelGGL((wgrad_tn_kernel<RR, ROWS>), dim3(nblocks), dim3(256), LDSB, stream, grp, zero_page);            \
  } while (0)
  if (g_wgrad_rows == 64) {
    if (g_wgrad_ring == 3) WGRAD_LAUNCH(3, 64);
    else if (g_wgrad_ring == 4) WGRAD_LAUNCH(4, 64);
    else WGRAD_LAUNCH(2, 64);
  } else if (g_wgrad_ring == 3) WGRAD_LAUNCH(3, 32);
  else if (g_wgrad_ring == 6) WGRAD_LAUNCH(6, 32);
  else WGRAD_LAUNCH(4, 32);
#undef WGRAD_LAUNCH
  if (nred > 0) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nred), dim3(256), 0, stream, grp);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

// probs: host array.  Splits m so that the whole group has about g_wgrad_blocks workgroups in flight; the
// partial slabs of all problems are carved out of the registered workspace (group flushed early when it is full).
static int launch_wgrad_group_kind(const WgradDesc* probs, int n, const void* zero_page, hipStream_t stream, bool row3);

int launch_wgrad_tn_group(const WgradDesc* probs, int n, const void* zero_page, hipStream_t stream) {
  if (n <= 0) return CL_OK;
  if (!zero_page) return CL_EINVAL;
  bool any3 = false, any1 = false;
  for (int i = 0; i < n; ++i) (probs[i].tap >= 16 ? any3 : any1) = true;
  if (any1) { const int rc = launch_wgrad_group_kind(probs, n, zero_page, stream, false); if (rc) return rc; }
  if (any3) return launch_wgrad_group_kind(probs, n, zero_page, stream, true);
  return CL_OK;
}

// one kind of problem per launch: single products / single taps (wgrad_tn_kernel) or rows of three taps (wgrad_row3_kernel)
static int launch_wgrad_group_kind(const WgradDesc* probs, int n, const void* zero_page, hipStream_t stream, const bool row3) {
  const int NT = row3 ? 3 : 1;
  void* ws; long ws_bytes;
  gemm_get_workspace_for(stream, &ws, &ws_bytes);
  long tiles_all = 0;
  for (int i = 0; i < n; ++i) {
    const WgradDesc& d = probs[i];
    if ((d.tap >= 16) != row3) continue;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) continue;
    if (d.N % 8 || d.K % 8 || d.lddy % 8 || d.ldx % 8 || d.lddw % 4 || d.N < 8 || d.K < 8 ||
        (reinterpret_cast<uintptr_t>(d.dW) & 15))
      return CL_EINVAL;
    if (row3) {   // tap = 16 + ky: the three taps of kernel row ky; stride 1, pad 1, a 32-row step = whole image-row segments
      if (d.tap > 18 || d.stride != 1 || d.pad != 1 || d.Hin != d.Hout || d.Win != d.Wout || d.Hout <= 0 || d.Wout <= 0 ||
          d.Hin > 32767 || d.Win > 32767 || d.M % (d.Hout * d.Wout) || d.M % 32 ||
          !((d.Wout % 32 == 0) || (d.Wout < 32 && 32 % d.Wout == 0)) || (d.K % 4) || d.lddw < 3L * d.K)
        return CL_EINVAL;
      tiles_all += (long)((d.N + 127) / 128) * ((d.K + 127) / 128);
      continue;
    }
    if (d.tap > 8 || (d.tap >= 0 && (d.Hin <= 0 || d.Win <= 0 || d.Hout <= 0 || d.Wout <= 0 || d.Hin > 32767 || d.Win > 32767 ||
                                     d.stride < 1 || d.stride > 2 || d.M % (d.Hout * d.Wout))))
      return CL_EINVAL;
    tiles_all += (long)((d.N + 127) / 128) * ((d.K + 127) / 128);
  }
  if (tiles_all == 0) return CL_OK;
  // uniform number of m-steps per workgroup across the group (a step = g_wgrad_rows rows of m)
  const int rows = row3 ? 32 : (g_wgrad_rows == 64 ? 64 : 32);
  long steps_all = 0;
  for (int i = 0; i < n; ++i)
    if (probs[i].M > 0 && (probs[i].tap >= 16) == row3)
      steps_all += (long)((probs[i].N + 127) / 128) * ((probs[i].K + 127) / 128) * ((probs[i].M + rows - 1) / rows);
  // (row problems: one 8-wave workgroup per CU -> one workgroup per CU's worth of splits; every split costs 3 N K floats of slab)
  const int want_blocks = row3 ? g_wgrad_row3_blocks : g_wgrad_blocks;
  long per = (steps_all + want_blocks - 1) / want_blocks;
  if (row3 && tiles_all > 0) {
    // one workgroup per CU: never a few workgroups more than a whole number of rounds (27 tiles x 19 splits = 513 workgroups ran
    // as three rounds) -- the largest split count whose grid stays within want_blocks
    long max_steps = 0;
    for (int i = 0; i < n; ++i)
      if (probs[i].M > 0 && probs[i].tap >= 16) max_steps = std::max(max_steps, (long)((probs[i].M + rows - 1) / rows));
    const long smax = std::max(1L, want_blocks / tiles_all);
    per = (max_steps + smax - 1) / smax;
  }
  const long min_steps = std::max(1, g_wgrad_min_steps * 32 / rows);
  if (per < min_steps) per = min_steps;

  WgradGroup grp; grp.n = 0; grp.pad = 0;
  int nblocks = 0, nred = 0; long ws_used = 0;
  for (int i = 0; i < n; ++i) {
    const WgradDesc& d = probs[i];
    if ((d.tap >= 16) != row3) continue;
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) continue;
    const int tn = (d.N + 127) / 128, tk = (d.K + 127) / 128, steps = (d.M + rows - 1) / rows;
    int splits = (int)((steps + per - 1) / per);
    int pp = (steps + splits - 1) / splits;
    splits = (steps + pp - 1) / pp;
    long need = splits > 1 ? (long)splits * NT * d.N * d.K * 4 : 0;
    if (need > ws_bytes) {   // cannot split this far: fewer, longer splits
      splits = (int)(ws_bytes / ((long)NT * d.N * d.K * 4));
      if (splits < 1) splits = 1;
      pp = (steps + splits - 1) / splits; splits = (steps + pp - 1) / pp;
      need = splits > 1 ? (long)splits * NT * d.N * d.K * 4 : 0;
    }
    if (grp.n == WGRAD_MAX_PROBS || ws_used + need > ws_bytes) {   // flush what we have
      const int rc = launch_group(grp, nblocks, nred, zero_page, stream, row3);
      if (rc) return rc;
      grp.n = 0; nblocks = 0; nred = 0; ws_used = 0;
    }
    WgradProb& P = grp.p[grp.n++];
    P.dy = (const bf16_t*)d.dy; P.x = (const bf16_t*)d.x; P.dW = d.dW;
    P.slab = splits > 1 ? reinterpret_cast<float*>((char*)ws + ws_used) : nullptr;
    P.lddy = d.lddy; P.ldx = d.ldx; P.lddw = d.lddw; P.M = d.M; P.N = d.N; P.K = d.K; P.alpha = d.alpha;
    P.tiles_k = tk; P.tiles = tn * tk; P.per = pp; P.splits = splits;
    P.blk0 = nblocks; P.red0 = nred;
    P.tap = d.tap; P.Hin = (short)d.Hin; P.Win = (short)d.Win; P.Hout = (short)d.Hout; P.Wout = (short)d.Wout;
    P.stride = (short)d.stride; P.pad = (short)d.pad;
    nblocks += tn * tk * splits;
    if (splits > 1) nred += (int)(((long)NT * d.N * (d.K / 4) + 255) / 256);
    ws_used += (need + 255) & ~255L;
  }
  if (grp.n) return launch_group(grp, nblocks, nred, zero_page, stream, row3);
  return CL_OK;
}

int launch_wgrad_tn(const void* dy, long lddy, const void* x, long ldx, float* dW, long lddw, int M, int N, int K,
                    float alpha, const void* zero_page, hipStream_t stream) {
  WgradDesc d{}; d.dy = dy; d.lddy = lddy; d.x = x; d.ldx = ldx; d.dW = dW; d.lddw = lddw; d.M = M; d.N = N; d.K = K;
  d.alpha = alpha; d.tap = -1;
  return launch_wgrad_tn_group(&d, 1, zero_page, stream);
}

}  // namespace cl
