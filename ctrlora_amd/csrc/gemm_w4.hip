// Loader / consumer tile kernel for the MFMA-bound contractions (bf16; launch configurations 40 / 41 of gemm.hip, and 47 / 48:
// the same with one PERSISTENT workgroup per CU).
//
//   out[M,N] = epilogue( A1[M,K1].W1[N,K1]^T  (+ A2[M,K2].W2[N,K2]^T) )          linear, or implicit 3x3 conv over NHWC
//
// Which products: the ResBlock / Downsample / Upsample 3x3 convolutions (ldm/modules/diffusionmodules/openaimodel.py:108-118,
// 150,203,229) and the deep-K linears (attention.py:59-76 FeedForward out, 163-170 projections at K >= 1280) -- what the
// 256 x 160 ping-pong tile kernel (gemm_fl_kernel, gemm.hip) serves at 42-44 % of the matrix peak.
//
// What bounds such a product on this chip (profiles/r06_w4/, tools/probe_simd.hip): POWER.  A bare stream of
// v_mfma_f32_32x32x16_bf16 on uniform random operands clocks down to 1.68 GHz = 1.75 PF/s (2.14 PF/s on constant operands); the
// same FLOPs as v_mfma_f32_16x16x32_bf16 hold 1.88 GHz = 1.94 PF/s (a quarter of the accumulator traffic per FLOP); and every
// KB moved beside the MFMAs comes out of the clock: +1 ds_read_b128 per 32-cycle MFMA slot -8.5 %, +1 LDS-DMA (1 KiB) per four
// slots -7.5 %.  Costs ADD -- measured on the first form of this kernel (one 32x32x16 wave per SIMD, 64 x 160 per wave): MFMA
// alone 29.5 us, + fragment reads +10, + DMA +17, + stores +11 = 64 us at 1.56 GHz against the ping-pong kernel's 55.  (s_memtime stamps in both kernels, later: the
// ping-pong kernel's main loop is 1527 cycles per stage at 1.6 GHz, this kernel's final form 1606 at 1.68 GHz -- the same
// 0.95 us per stage, i.e. both sit on the power limit of their operand traffic)
// So the levers are joules, not issue slots: the cheaper MFMA shape, fewer DMA bytes per FLOP, no stall that is not hidden.
//
// Structure: two KINDS of wave, for the whole tile.
//   * waves 0-3, the CONSUMERS, one per SIMD: each owns 64 rows x BN columns of the 256 x BN tile and issues nothing but
//     v_mfma_f32_16x16x32_bf16 (inline asm) and ds_read_b128.  Per 128-byte stage and wave: 2 NW groups of four MFMAs
//     (NW = BN / 16; group k multiplies W fragment k by the four X fragments of its k-half), W fragments through a ring of
//     R = NW / 2 register quads (the read of W[k + R] goes out behind group k), X fragments double-buffered per k-half; every
//     read has >= R groups (320 cycles) to land and is waited for with a COUNTED lgkmcnt (LDS returns in order; the counts come
//     from a constexpr replay of the issue order, w4_sched).  The product is formed TRANSPOSED (D^T[n x m] = W[n x k] . X^T[k x m])
//     so that a lane holds 4 consecutive output columns of one row; v_permlane16_swap pairs neighbouring 16-column blocks and
//     every lane stores 8 consecutive columns, a wave-instruction whole 64-byte row segments, straight from registers;
//   * waves 4-7, the LOADERS, one per SIMD: all address generation and every global_load_lds_dwordx4 (8 rows x 128 B per
//     instruction, XOR-swizzled on the SOURCE side: the fragment reads are bank-conflict-free).
//   * a 3-slot LDS ring and TWO barriers per stage, neither of which makes anybody wait for a read:
//       consumer  stage s:  group 0 | B_war(s-1) | groups 1 .. GB | B_raw(s+1) | groups GB+1 .. (first reads of stage s+1)
//       loader    ... B_war(s-1) | DMA(s+2) -> slot of stage s-1 | vmcnt: own DMA(s+1) landed | B_raw(s+1) ...
//     RAW: every loader's counted vmcnt for stage s+1 precedes B_raw(s+1); the first read of stage s+1 follows it.
//     WAR: the slot of stage s-1 is refilled after B_war(s-1), which a consumer passes behind group 0 of stage s -- that group
//          waited for W[0] of stage s, a read issued after every read of stage s-1 (in-order return: they have all retired).
//     A DMA has ~1.7 stages to land -- about what an LDS-DMA round trip takes under load (~2000-2500 cycles for HBM / Infinity-Cache
//     data, less for the L2-resident weights), and LDS cannot hold a fourth slot: the chain is just-in-time by construction.
//     (Tried: LDS mailboxes -- ds_add counters polled through the consumers' in-order read queue -- instead of barriers, so that
//     one wave's LDS jitter is not everybody's stall.  Correct, and slower: the polls' round trips add straight onto the chain,
//     DMA + synchronisation alone 1263 -> 1650 cycles per stage, profiles/r06_w4/probe_w4_v3_lds_mailboxes_instead_of_barriers.log.)
//   * HALO mode (stride-1 3x3 convolutions whose 256-row tile lies inside one image, H W % 256 == 0, W <= 64): the X operand is
//     not re-fetched per tap.  Per 64-channel chunk the loaders bring the tile's input pixels ONCE, with their one-pixel halo,
//     as an LDS image of (R + 2) x (W + 1) + 1 pixel rows of 128 bytes (one zero column between image rows serves as right AND left
//     padding; zero rows above / below the image: no masks anywhere);
//     the nine taps of the chunk are nine stages whose X fragment rows are the same image read at a wave-uniform row offset
//     (ky (W + 1) + kx), and only the 20 KB weight tile streams per stage.  K is walked chunk-major (chunk, tap) instead of
//     tap-major.  LDS-DMA instructions per stage and SIMD: 13 -> 6.4; two image buffers (the next chunk's lands under this one's
//     nine stages, its pieces riding on taps 2 .. 8: the loaders run two stages ahead, and the buffer they fill was read until the
//     previous chunk's last tap) + the 3-slot weight ring + the epilogue operands' 2 KB = 160 KB.  (At the 32x32 / 16x16 levels the
//     weight matrix outgrows an XCD's L2 and two stages of look-ahead no longer cover its latency: 2290 cycles per stage; the
//     tuner keeps those on the ping-pong kernel.)  The loaders SPECIALISE: waves 4-5 stream nothing but weight
//     tiles (L2-resident: every CU of the grid reads the same 1.8 MB), waves 6-7 nothing but image pieces (HBM / Infinity Cache) --
//     vmcnt retires in order, and a weight tile must not wait behind an image piece nobody needs before the next chunk.
//   * PERSISTENT form (configurations 47 / 48; whole-K tiles of >= 3 stages, more virtual tiles than CUs): grid = CUs rounded down
//     to a multiple of 8, a workgroup walks tile ids id, id + grid, ... (all on its XCD) and the stages of ALL its tiles are one
//     sequence for the ring: the loaders are two stages into the next tile while the consumers store the finished one, the
//     halo image is handed over between tiles (its pieces for the next tile's first chunk ride on the last chunk's taps).
//     Where it wins (tools/gemm_autotune.py, profiles/r06_tune/): launches with several tiles per CU -- the first stage's
//     3x3 convs, DDIM-size linears.  One tile per CU (the training step's 64x64 convs at batch 8): the ping-pong kernel stays.
//   * EPILOGUE: vmcnt retires in order, so ANY load behind a store -- a bias row, a residual vector, a scratch reload of an
//     address the compiler parked before the main loop -- waits for every store before it; measured 12 us of a 59 us tile.
//     Hence: a tile's bias floats / row-bias come to a 1 KB LDS block by one DMA issued with the tile's first stage (double-
//     buffered by tile parity); accumulators leave the AGPR half by explicit v_accvgpr_read; each 16-row x 32-column unit
//     re-derives its coordinates from v_mbcnt; the residual is fetched one row block ahead; no scratch anywhere (a tested
//     property: tests/test_isa_gemm_w4.py).  What remains is the store issue rate of four waves (~190 cycles per
//     global_store_dwordx4 and wave: 16.4 K cycles per tile where the ping-pong kernel's eight waves take 9.5 K).
// All waves of a workgroup share one register allocation: 512 threads = two waves per SIMD = 256 registers; with accumulation
// registers in use hipcc splits them 128 | 128, so 128 accumulators live in "a" registers, 32 in "v" beside 52 fragment registers.
#include <type_traits>
#include "gemm.h"
#include "gemm_epi.h"

namespace cl {
namespace {

template <int I, int N, typename F> __device__ __forceinline__ void w4_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); w4_for<I + 1, N>(f); }
}
template <int OFF> __device__ __forceinline__ void w4_rd128(u32x4_t& v, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
// c += a . b (the compiler pads nothing around an asm MFMA: the stream never reads an accumulator, the epilogue drains first)
template <bool AG> __device__ __forceinline__ void w4_mfma(f32x4_t& c, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ void w4_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// s_waitcnt vmcnt(n) for a run-time n (the count is an instruction immediate)
__device__ __forceinline__ void w4_vm_rt(int n) {
  switch (n) {
#define VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    VMW(0) VMW(1) VMW(2) VMW(3) VMW(4) VMW(5) VMW(6) VMW(7) VMW(8) VMW(9) VMW(10) VMW(11) VMW(12) VMW(13) VMW(14) VMW(15) VMW(16)
    VMW(17) VMW(18) VMW(19) VMW(20) VMW(21) VMW(22) VMW(23) VMW(24) VMW(25) VMW(26) VMW(27) VMW(28) VMW(29) VMW(30) VMW(31) VMW(32)
#undef VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
template <int N> __device__ __forceinline__ void w4_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
// wait states around the asm MFMAs: behind the v_accvgpr_write of the zero fill (SHORT), and before anything reads a result
template <bool SHORT, bool AG> __device__ __forceinline__ void w4_pad(f32x4_t& a, f32x4_t& b, f32x4_t& c, f32x4_t& d) {
  if constexpr (SHORT) {
    if constexpr (AG) asm volatile("s_nop 7" : "+a"(a), "+a"(b), "+a"(c), "+a"(d)); else asm volatile("s_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  } else {
    if constexpr (AG) asm volatile("s_nop 15" : "+a"(a), "+a"(b), "+a"(c), "+a"(d)); else asm volatile("s_nop 15" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  }
}

enum { W4_LINEAR = 0, W4_CONV_S1 = 1, W4_CONV_ANY = 2, W4_CONV_HALO = 3 };

// ---- the consumers' LDS issue order, replayed at compile time.  Per stage: groups g = 0 .. G2-1 (G2 = 2 NW); behind group g go out
//   W[g + R]                 (fragment index continues into the next stage: g + R >= G2 reads stage s+1, k-half 0)
//   X(k-half 1, i = g)       for g < 4
//   X(next stage, k-half 0)  two behind group GB + 1, two behind GB + 2          (GB = G2 - R - 1 carries B_raw)
// Group g needs W[g] (+ the four X of its k-half at g = 0 and g = NW); LDS returns in order, so it may run when at most the
// reads issued BEHIND the last one it needs are outstanding.
template <int NW> struct W4Sched { int allow[2 * NW]; };
template <int NW> constexpr W4Sched<NW> w4_sched() {
  constexpr int G2 = 2 * NW, R = NW / 2, GB = G2 - R - 1;
  int seq[3 * (G2 * 4)] = {};          // > 0: a read id, < 0: group marker -(1000 S + g) - 1
  int n = 0;
  for (int S = 0; S < 3; ++S)
    for (int g = 0; g < G2; ++g) {
      seq[n++] = -(1000 * S + g) - 1;
      const int k = g + R;
      seq[n++] = k < G2 ? 1000 * S + k + 1 : 1000 * (S + 1) + (k - G2) + 1;                 // W id: 1000 S + k + 1
      if (g < 4) seq[n++] = 1000 * S + 200 + g + 1;                                          // X(k-half 1, g)
      if (g == GB + 1) { seq[n++] = 1000 * (S + 1) + 100 + 0 + 1; seq[n++] = 1000 * (S + 1) + 100 + 1 + 1; }
      if (g == GB + 2) { seq[n++] = 1000 * (S + 1) + 100 + 2 + 1; seq[n++] = 1000 * (S + 1) + 100 + 3 + 1; }
    }
  W4Sched<NW> r{};
  for (int g = 0; g < G2; ++g) {       // stage S = 1: its first reads went out in stage 0's tail (the steady state)
    int pg = -1, last = -1;
    for (int i = 0; i < n; ++i) {
      if (seq[i] == -(1000 + g) - 1) pg = i;
      bool need = seq[i] == 1000 + g + 1;
      if (g == 0) for (int x = 0; x < 4; ++x) need = need || seq[i] == 1000 + 100 + x + 1;
      if (g == NW) for (int x = 0; x < 4; ++x) need = need || seq[i] == 1000 + 200 + x + 1;
      if (need && pg < 0) last = i;     // (needed reads sit before the group marker)
    }
    int cnt = 0;
    for (int i = last + 1; i < pg; ++i) cnt += seq[i] > 0 ? 1 : 0;
    r.allow[g] = cnt > 15 ? 15 : cnt;
  }
  return r;
}
template <int NW, int g> constexpr int w4_allow() { return w4_sched<NW>().allow[g]; }

// ABL (template parameter; non-zero instances exist in probe builds only, tools/probe_gemm.hip -DW4_PROBE): bit 0 = consumers skip
// their fragment reads, bit 1 = loaders skip their DMA, bit 2 = no stores, bit 5 = no MFMAs -- wrong results by construction
#define W4_ABL(bit) ((ABL & (bit)) != 0)
#ifdef W4_PROBE
int g_w4_abl_host = 0, g_w4_halo_host = 1;
// wave 0 of every workgroup stores s_memtime at four points of its tile: entry | stage 0 landed | main loop done | stores retired
__device__ unsigned long long* g_w4_timing = nullptr;
#define W4_STAMP(i) do { if (g_w4_timing && threadIdx.x == 0) g_w4_timing[(long)blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4_STAMP(i) do { } while (0)
#endif

constexpr int W4_BM = 256, W4_NC = 4, W4_NL = 4, W4_R = 3;
constexpr int W4_HROWS = 392;                     // HALO: pixel rows of one image buffer (>= (R + 2) (W + 1) + 1, a multiple of 8)
constexpr int W4_XJI = 25;                        // HALO: image DMA instructions per image loader and chunk (2 x 25 >= 392 / 8)
// HALO: an image loader's pieces [lo, hi) of the next chunk ride on the stage with tap t: taps 2 .. 8 carry 4 4 4 4 4 4 1
__host__ __device__ constexpr int w4_pieces_lo(int t) { return t < 2 ? 0 : 4 * (t - 2); }
__host__ __device__ constexpr int w4_pieces_hi(int t) { return t < 2 ? 0 : (4 * (t - 2) + 4 > W4_XJI ? W4_XJI : 4 * (t - 2) + 4); }

// the row-bias of a tile can sit in LDS (one row of it) iff all the tile's rows belong to one batch element
__device__ __forceinline__ bool w4_rb_uniform(const GemmParams& p, int tm0) {
  if (!p.rowbias) return false;
  const int last = min(tm0 + W4_BM, p.M) - 1;
  return tm0 / p.rows_per_batch == last / p.rows_per_batch;
}
// an accumulator register as a VGPR operand.  (Left to the compiler, the copies out of the AGPR half went through SCRATCH: a
// store / reload pair per quad, and every reload's vmcnt(0) also waited for the tile's global stores to drain.)
template <bool AG> __device__ __forceinline__ float w4_get(float a) {
  if constexpr (AG) { float x; asm("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a)); return x; }
  else return a;
}

// NW: 16-column fragment blocks per tile row (BN = 16 NW: 10 -> 160, 8 -> 128).
template <int NW, int MODE, int ABL = 0>
__global__ __launch_bounds__(64 * (W4_NC + W4_NL)) void gemm_w4_kernel(GemmParams p, int tiles_m, int tiles_n,
                                                                      float* __restrict__ slab, int nvirt, int ngrid) {
  typedef bf16_t T;
  constexpr bool HALO = MODE == W4_CONV_HALO;
  constexpr int BN = 16 * NW, G2 = 2 * NW, R = NW / 2, GB = G2 - R - 1;
  constexpr int XI = W4_BM / 8, WI = BN / 8;            // DMA instructions (8 rows x 128 B) per stage: X rows, W rows
  constexpr int XJ = XI / W4_NL, WJ = WI / W4_NL;       // ... per loader wave
  constexpr int G = HALO ? WJ : XJ + WJ;                // per loader and stage (HALO: + the image pieces riding on the stage)
  constexpr int GA = W4_ABL(2) ? 0 : G;
  constexpr int XBUF = W4_HROWS * 128;                  // HALO: one image buffer
  constexpr int SLOT = HALO ? WI * 1024 : (XI + WI) * 1024;
  constexpr int RING0 = HALO ? 2 * XBUF : 0;            // byte offset of the ring
  constexpr int WOFF = HALO ? 0 : XI * 1024;            // W rows inside a slot
  constexpr int EPI0 = RING0 + W4_R * SLOT;             // two 1 KB blocks (tile parity): the epilogue's bias / row-bias operands
  static_assert(WI % W4_NL == 0 && G2 % R == 0 && NW % 2 == 0, "uniform DMA counts; the W ring is periodic per stage");
  static_assert(EPI0 + 2048 <= 160 * 1024 && BN * 4 + BN * 2 <= 1024, "LDS");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- A workgroup walks the virtual tile ids blockIdx.x, blockIdx.x + ngrid, ... < nvirt (ngrid < nvirt: the PERSISTENT form,
  // ngrid a multiple of 8 so that a workgroup's tiles stay on its XCD; ngrid == nvirt: one tile each).  XCD-aware tile id (as
  // gemm_fl_kernel): XCD x (= id mod 8) owns a contiguous range of tiles.  All the stages of all its tiles are ONE sequence for the
  // ring protocol: the loaders run straight into the next tile while the consumers store the finished one, and those stores
  // (straight from registers, asynchronous) drain under the next tile's MFMAs.
  if ((int)blockIdx.x >= ngrid) return;                     // launch-tag workgroups (debug_hooks.h)
  const int nt = tiles_m * tiles_n;
  const int cpt = p.K1 / 64;  // 64-channel chunks (= stages per tap)
  const int ks1 = (MODE == W4_LINEAR ? 1 : 9) * cpt;
  const int ks2 = (MODE == W4_LINEAR) ? p.K2 / 64 : 0;
  struct Tile { int m0, n0, zsplit, kbeg, total; };
  const int splitk = p.splitk, tn_ = tiles_n;
  auto decode = [=](int v) __attribute__((always_inline)) -> Tile {                          // (by VALUE throughout: nothing here may end up address-taken)
    int pid = v;
    const int zs = pid / nt;
    pid -= zs * nt;
    {
      const int q = nt >> 3, r = nt & 7, xcd = pid & 7, idx = pid >> 3;
      pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int kb = 0, ke = ks1 + ks2;
    if (splitk > 1) {
      const int per = (ke + splitk - 1) / splitk;
      kb = zs * per;
      ke = min(ke, kb + per);
    }
    return Tile{(pid / tn_) * W4_BM, (pid % tn_) * BN, zs, kb, ke - kb};      // total >= 1 (launcher)
  };
  const Tile t0 = decode((int)blockIdx.x);                  // the first tile (HALO: the only one)
  const int m0 = t0.m0, n0 = t0.n0, kbeg = t0.kbeg, total = t0.total;
  // HALO: stage index k = 9 chunk + tap; the tile's pixels are rows [py0, py0 + 256 / W) of image pb, all W columns
  const int Wp = HALO ? p.Win + 1 : 0;                      // pixel rows per image row of the LDS image (W pixels + one zero column)
  const int cc0 = HALO ? kbeg / 9 : 0, tap0 = HALO ? kbeg - cc0 * 9 : 0;

  if (wave >= W4_NC) {
    // =================================================================================== loader
    const int lw = wave - W4_NC;
    const int lrow = lane >> 3, lslot = lane & 7;
    const char* zpage = (const char*)p.zero_page + lslot * 16;
    // A tile's epilogue operands -> LDS (1 KB: BN bias floats, then BN row-bias bf16 when the tile's rows share a batch element):
    // ONE DMA instruction by loader 0, issued ahead of the tile's first stage, so that it has landed when that stage has.  The
    // consumers' epilogue then reads LDS only: a global load there would queue behind the stores.  Lanes without a source (no
    // bias / row-bias, columns past N) are masked off -- the epilogue reads only what was brought (linear calls have no zero page).
    auto issue_epi = [&](int tm0, int tn0, int par) __attribute__((always_inline)) {
      if (lw != 0 || slab) return;
      const char* src = nullptr;
      if (lane < BN / 4) {
        const int c = tn0 + 4 * lane;
        if (p.bias && c < p.N) src = (const char*)(p.bias + c);
      } else if (lane < BN / 4 + BN / 8) {
        const int c = tn0 + 8 * (lane - BN / 4);
        if (w4_rb_uniform(p, tm0) && c < p.N)
          src = (const char*)(reinterpret_cast<const T*>(p.rowbias) + (long)(tm0 / p.rows_per_batch) * p.ldrb + c);
      }
      if constexpr (!W4_ABL(2)) { if (src) glds16(src, smem + EPI0 + par * 1024); }
    };
    if constexpr (HALO) {
      // ---- HALO: waves 4-5 stream the weight tiles, waves 6-7 the image pieces (see the top)
      if (lw < 2) {
        constexpr int WJ2 = WI / 2;                           // weight DMA instructions per stage and weight loader
        const char* pw[WJ2];
        auto setup_w = [&](int tn0) __attribute__((always_inline)) {
#pragma unroll
          for (int j = 0; j < WJ2; ++j) {
            const int inst = j * 2 + lw;
            const int chunk = (lslot ^ (((inst & 1) << 2) | (lrow >> 1))) * 16;
            int n = tn0 + inst * 8 + lrow;
            n = min(n, p.N - 1);
            pw[j] = (const char*)p.W1 + ((long)n * p.ldw1) * sizeof(T) + chunk;      // + (tap C + 64 chunk) 2 per stage
          }
        };
        setup_w(n0);
        int tap = tap0, cc = cc0;
        auto issue_w = [&](int slot) __attribute__((always_inline)) {
          char* Ws = smem + RING0 + slot * SLOT;
          const long koff = ((long)tap * p.K1 + (long)cc * 64) * sizeof(T);        // this stage's 64 K columns of W
          if constexpr (!W4_ABL(2)) {
#pragma unroll
            for (int j = 0; j < WJ2; ++j) glds16(pw[j] + koff, Ws + (j * 2 + lw) * 1024);
          }
          if (++tap == 9) { tap = 0; ++cc; }
        };
        // the stages of all this workgroup's tiles in order (a further tile exists in the persistent form only: whole K ranges)
        int v = (int)blockIdx.x, left = total, tpar = 0;
        issue_epi(m0, n0, 0);
        auto next_w = [&](int slot) __attribute__((always_inline)) -> bool {
          if (left == 0) {
            v += ngrid;
            if (v >= nvirt) return false;
            const Tile t = decode(v);
            setup_w(t.n0);
            tap = 0; cc = 0; left = t.total;
            tpar ^= 1;
            issue_epi(t.m0, t.n0, tpar);
          }
          issue_w(slot);
          --left;
          return true;
        };
        constexpr int GW = W4_ABL(2) ? 0 : WJ2;
        next_w(0);
        bool nxt = next_w(1);
        if (nxt) w4_vm<GW>(); else w4_vm<0>();
        __builtin_amdgcn_s_barrier();                         // B_raw(0)
        __builtin_amdgcn_sched_barrier(0);
        int slot2 = 2;
        bool curs = true;
        while (curs) {
          __builtin_amdgcn_s_barrier();                       // B_war(s-1): the slot of stage s-1 is free
          __builtin_amdgcn_sched_barrier(0);
          const bool n2 = nxt ? next_w(slot2) : false;
          if (n2) w4_vm<GW>(); else w4_vm<0>();
          __builtin_amdgcn_s_barrier();                       // B_raw(s+1)
          __builtin_amdgcn_sched_barrier(0);
          curs = nxt; nxt = n2;
          slot2 = (slot2 == W4_R - 1) ? 0 : slot2 + 1;
        }
        return;
      }
      // image loader il = 0 / 1: pieces il, il + 2, ... of every image (W4_XJI each; past the image: the last piece again)
      const int il = lw - 2;
      const int hw = p.Hin * p.Win;
      const int rows = W4_BM / p.Win;
      const int nrow = (rows + 2) * Wp + 1;
      const int ninst = (nrow + 7) / 8;
      // piece j of this loader = image rows 8 (2 j + il) + lrow: (hy, hx) walk on by 16 rows per piece (W + 1 >= 17: one wrap at most).
      const char* pa[W4_XJI];
      auto set_pa = [&](int tm0, int chunk0) __attribute__((always_inline)) {     // pointers of the tile at rows tm0, 64-channel chunk chunk0
        const int pb = tm0 / hw, py0 = (tm0 - pb * hw) / p.Win;
        int hr = il * 8 + lrow;
        int hy = hr / Wp, hx = hr - hy * Wp;
#pragma unroll
        for (int j = 0; j < W4_XJI; ++j) {
          const int inst = j * 2 + il;
          const int chunk = (lslot ^ (((inst & 1) << 2) | (lrow >> 1))) * 16;
          const int y = py0 - 1 + hy, x = hx - 1;        // (hx = 0: the zero column)
          const bool ok = hr < nrow && (unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win;
          pa[j] = ok ? (const char*)p.A1 + ((((long)pb * p.Hin + y) * p.Win + x) * p.lda1) * sizeof(T) + chunk + (long)chunk0 * 128
                     : zpage + (long)chunk0 * 128;
          hr += 16; hx += 16;
          if (hx >= Wp) { hx -= Wp; ++hy; }
        }
      };
      // The FIRST image is brought by the consumer waves (they idle until stage 0 has landed): pointers start at chunk cc0 + 1.
      set_pa(m0, cc0 + 1);
      // pieces [lo, hi) of the NEXT image into buffer `buf` (pieces past the image do not exist: this loader only ever waits for
      // everything); a piece's pointer then moves on a chunk
      auto issue_image = [&](int buf, int lo, int hi) __attribute__((always_inline)) {
        char* Xb = smem + buf * XBUF;
#pragma unroll
        for (int j = 0; j < W4_XJI; ++j)
          if (j >= lo && j < hi) {
            const int inst = j * 2 + il;
            if constexpr (!W4_ABL(2)) { if (inst < ninst) glds16(pa[j], Xb + inst * 1024); }
            pa[j] += 128;
          }
      };
      // The pieces of the next image ride on the steps that issue taps 2 .. 8 of the current chunk (the loaders run two stages
      // ahead; the buffer was read until the previous chunk's last tap); the first step of a split-K range also carries what the
      // taps before it would have (nothing has been read yet).  The next image is the tile's next chunk or -- persistent form,
      // whole K ranges only -- the first chunk of the workgroup's next tile: the pointers are re-made on the step of tap 2, when
      // every piece of the tile's own last image has gone out.  They must have landed at the B_raw of the image's first stage:
      // the step that issues tap 1 of it (two stages ahead) waits for everything.
      int v = (int)blockIdx.x, left = total, tap = tap0, cc = cc0, ibuf = 0;
      int cclast = (kbeg + total - 1) / 9;
      bool first = true;
      auto step_next = [&]() __attribute__((always_inline)) -> bool {   // the step that (on the weight side) issues the stage (v, cc, tap)
        if (left == 0) {
          v += ngrid;
          if (v >= nvirt) return false;
          left = 9 * cpt; tap = 0; cc = 0; cclast = cpt - 1;
        }
        const bool more = cc < cclast, nextt = !more && v + ngrid < nvirt;
        if (nextt && tap == 2) { const Tile t = decode(v + ngrid); set_pa(t.m0, 0); }
        if (more || nextt) {
          const int lo = first ? 0 : w4_pieces_lo(tap), hi = w4_pieces_hi(tap);
          if (hi > lo) issue_image(ibuf ^ 1, lo, hi);
        }
        first = false;
        if (++tap == 9) { tap = 0; ++cc; ibuf ^= 1; }
        --left;
        return true;
      };
      step_next();
      bool nxt = step_next();
      w4_vm_rt(0);                                            // (prologue: everything, the next image's first pieces included)
      __builtin_amdgcn_s_barrier();                           // B_raw(0)
      __builtin_amdgcn_sched_barrier(0);
      bool curs = true;
      while (curs) {
        __builtin_amdgcn_s_barrier();                         // B_war(s-1)
        __builtin_amdgcn_sched_barrier(0);
        // stage s+1 is read after B_raw(s+1): if it is an image's first stage the image must be complete (tap == 1 here <=> the
        // stage this step issues, s+2, has tap 1 <=> stage s+1 has tap 0)
        const bool chunk_start_next = tap == 1;
        const bool n2 = nxt ? step_next() : false;
        if (chunk_start_next) w4_vm<0>();
        __builtin_amdgcn_s_barrier();                         // B_raw(s+1)
        __builtin_amdgcn_sched_barrier(0);
        curs = nxt; nxt = n2;
      }
      return;
    } else {
    const char* pa[XJ];            // LINEAR: running pointer.  CONV: centre pixel (S1) / base (ANY)
    const char* a2[XJ];
    uint32_t vmask[XJ];            // CONV_S1: bit t = tap t reads inside the image
    int ab[XJ], ay[XJ], ax[XJ];    // CONV_ANY: output pixel coordinates
    const char* pw[WJ]; const char* w2[WJ];
    const char* cur[XJ];
    const long pixb = (long)p.lda1 * sizeof(T);
    // wave-uniform walk over (tap, channel chunk) of the tile being issued
    int kt_next = 0, tap = 0, cc = 0;
    long tapoff = 0;
    // per-lane sources of one tile
    auto setup_tile = [&](int tm0, int tn0, int tkbeg) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const int inst = j * W4_NL + lw;
        const int chunk = (lslot ^ (((inst & 1) << 2) | (lrow >> 1))) * 16;
        int r = tm0 + inst * 8 + lrow;
        r = min(r, p.M - 1);
        a2[j] = nullptr; vmask[j] = 0; ab[j] = ay[j] = ax[j] = 0; cur[j] = nullptr;
        if constexpr (MODE == W4_LINEAR) {
          const bool in2 = tkbeg >= ks1;
          a2[j] = p.A2 ? (const char*)p.A2 + ((long)r * p.lda2 + (p.a2_group_n ? (long)(tn0 / p.a2_group_n) * p.K2 : 0)) * sizeof(T) + chunk
                       : nullptr;
          pa[j] = in2 ? a2[j] + (long)(tkbeg - ks1) * 128
                      : (const char*)p.A1 + ((long)r * p.lda1 + (p.a1_group_n ? (long)(tn0 / p.a1_group_n) * p.K1 : 0)) * sizeof(T) +
                            chunk + (long)tkbeg * 128;
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
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const int inst = j * W4_NL + lw;
        const int chunk = (lslot ^ (((inst & 1) << 2) | (lrow >> 1))) * 16;
        int n = tn0 + inst * 8 + lrow;
        n = min(n, p.N - 1);
        w2[j] = (MODE == W4_LINEAR && p.W2) ? (const char*)p.W2 + ((long)n * p.ldw2) * sizeof(T) + chunk : nullptr;
        pw[j] = (MODE == W4_LINEAR && tkbeg >= ks1)
                    ? w2[j] + (long)(tkbeg - ks1) * 128
                    : (const char*)p.W1 + ((long)n * p.ldw1) * sizeof(T) + chunk + (long)tkbeg * 128;
      }
      kt_next = tkbeg;
      tap = (MODE == W4_LINEAR) ? 0 : tkbeg / cpt;
      cc = (MODE == W4_LINEAR) ? 0 : tkbeg - tap * cpt;
      tapoff = 0;
      if constexpr (MODE == W4_CONV_S1) {
        tapoff = ((long)(tap / 3 - 1) * p.Win + (tap % 3 - 1)) * pixb;
#pragma unroll
        for (int j = 0; j < XJ; ++j)   // a split-K workgroup may start in the middle of a tap
          cur[j] = ((vmask[j] >> tap) & 1u) ? pa[j] + tapoff + (long)cc * 128 : zpage + (long)cc * 128;
      }
    };
    auto issue = [&](int slot) __attribute__((always_inline)) {
      char* Xs = smem + RING0 + slot * SLOT;
      char* Ws = Xs + WOFF;
      if constexpr (W4_ABL(2)) { ++kt_next; return; }
      if constexpr (MODE == W4_LINEAR) {
        if (ks2 && kt_next == ks1) {   // switch to the second K segment (LoRA up-projection)
#pragma unroll
          for (int j = 0; j < XJ; ++j) pa[j] = a2[j];
#pragma unroll
          for (int j = 0; j < WJ; ++j) pw[j] = w2[j];
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) { glds16(pa[j], Xs + (j * W4_NL + lw) * 1024); pa[j] += 128; }
      } else if constexpr (MODE == W4_CONV_S1) {
        // cur[j] walks the channel chunks of the current tap (+128 B per stage); lanes whose tap falls outside the image walk
        // the zero page instead (at least (K1 / 64 + 1) * 128 bytes long); the select against the 9-bit mask happens only when
        // the tap changes
        if (cc == 0) {
#pragma unroll
          for (int j = 0; j < XJ; ++j) cur[j] = ((vmask[j] >> tap) & 1u) ? pa[j] + tapoff : zpage;
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) { glds16(cur[j], Xs + (j * W4_NL + lw) * 1024); cur[j] += 128; }
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
      for (int j = 0; j < WJ; ++j) { glds16(pw[j], Ws + (j * W4_NL + lw) * 1024); pw[j] += 128; }
      ++kt_next;
    };

    // the stages of ALL this workgroup's tiles, in order: issue_next() moves on to the next tile when one is exhausted
    int v = (int)blockIdx.x, left = total, tpar = 0;
    setup_tile(m0, n0, kbeg);
    issue_epi(m0, n0, 0);
    auto issue_next = [&](int slot) __attribute__((always_inline)) -> bool {
      if (left == 0) {
        v += ngrid;
        if (v >= nvirt) return false;
        const Tile t = decode(v);
        setup_tile(t.m0, t.n0, t.kbeg);
        left = t.total;
        tpar ^= 1;
        issue_epi(t.m0, t.n0, tpar);
      }
      issue(slot);
      --left;
      return true;
    };
    issue_next(0);
    bool nxt = issue_next(1);                                 // stage s+1 exists
    if (nxt) w4_vm<GA>(); else w4_vm<0>();
    __builtin_amdgcn_s_barrier();                             // B_raw(0)
    __builtin_amdgcn_sched_barrier(0);
    int slot2 = 2;
    bool curs = true;                                         // stage s exists
    while (curs) {
      __builtin_amdgcn_s_barrier();                           // B_war(s-1): the slot of stage s-1 is free
      __builtin_amdgcn_sched_barrier(0);
      const bool n2 = nxt ? issue_next(slot2) : false;
      if (n2) w4_vm<GA>(); else w4_vm<0>();
      __builtin_amdgcn_s_barrier();                           // B_raw(s+1)
      __builtin_amdgcn_sched_barrier(0);
      curs = nxt; nxt = n2;
      slot2 = (slot2 == W4_R - 1) ? 0 : slot2 + 1;
    }
    return;
    }
  }

  // ===================================================================================== consumer
  const int lr = lane & 15, lg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // fragment (16 rows, k-half h of the stage: logical 16-byte chunk lg + 4 h) -> slot (lg ^ swz(row)) ^ 4 h; swz(row) = (row >> 1) & 7
  uint32_t wa[2], xa[HALO ? 4 : 2];
  uint32_t xh0[4];                              // HALO: centre-tap image row of this lane's pixel, per X fragment
  const int wsh = HALO ? 31 - __builtin_clz(p.Win) : 0;
  {
    const uint32_t fo = lr * 128 + ((lg ^ ((lr >> 1) & 7)) * 16);
    wa[0] = lds0 + RING0 + WOFF + fo; wa[1] = wa[0] ^ 64u;
#pragma unroll
    for (int i = 0; i < 4; ++i) xh0[i] = 0;
    if constexpr (!HALO) { xa[0] = lds0 + (wave * 64) * 128 + fo; xa[1] = xa[0] ^ 64u; }
  }
  // (made again at the start of every tile, from a fresh lane id: kept across the epilogue they went to scratch, and the reload's
  // vmcnt(0) at the next tile's start waited for the finished tile's stores)
  auto make_xh0 = [&]() __attribute__((always_inline)) {
    if constexpr (HALO) {
      int l2;
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l2));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int px = wave * 64 + 16 * i + (l2 & 15);                 // tile-local pixel: image row px / W, column px % W
        const int yl = px >> wsh, xl = px & (p.Win - 1);               // (W is a power of two: w4_halo_ok)
        xh0[i] = (yl + 1) * Wp + xl + 1;
      }
    }
  };
  make_xh0();
  int ctap = tap0, cbuf = 0;                    // HALO: tap / image buffer of the stage whose X addresses are in xa
  auto halo_addr = [&](int tp, int buf) __attribute__((always_inline)) {
    if constexpr (HALO) {
      const int ky = (tp * 11) >> 5, kx = tp - 3 * ky;
      const int toff = (ky - 1) * Wp + (kx - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t h = xh0[i] + toff;
        xa[i] = lds0 + buf * XBUF + h * 128 + ((lg ^ ((h >> 1) & 7)) * 16);
      }
    }
  };
  halo_addr(ctap, cbuf);

  f32x4_t acc[NW][4];
  u32x4_t Xf[2][4], Wr[R];

  auto read_w = [&](auto Kc) {                  // W fragment k (0 .. G2-1 of the stage whose addresses are in wa) -> ring slot k % R
    constexpr int k = decltype(Kc)::value;
    if constexpr (W4_ABL(1)) return;
    else w4_rd128<(k % NW) * 2048>(Wr[k % R], wa[k / NW]);
  };
  auto read_x = [&](auto Hc, auto Ic) {         // X fragment i of k-half h
    constexpr int h = decltype(Hc)::value, i = decltype(Ic)::value;
    if constexpr (W4_ABL(1)) return;
    else if constexpr (HALO) {
      if constexpr (h == 0) w4_rd128<0>(Xf[0][i], xa[i]);
      else { const uint32_t a1 = xa[i] ^ 64u; w4_rd128<0>(Xf[1][i], a1); }
    } else {
      w4_rd128<i * 2048>(Xf[h][i], xa[h]);
    }
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;

  W4_STAMP(0);
  if constexpr (HALO) {
    // the FIRST image: the four consumer waves have nothing to do until stage 0 has landed and bring it themselves (pieces
    // wave, wave + 4, ...: 13 each cover the 49) -- the image loaders start with the next chunk's
    const int lrow = lane >> 3, lslot = lane & 7;
    const char* zpage = (const char*)p.zero_page + lslot * 16;
    const int hw = p.Hin * p.Win;
    const int pb = m0 / hw, py0 = (m0 - pb * hw) >> wsh, rows = W4_BM >> wsh;
    const int nrow = (rows + 2) * Wp + 1, ninst = (nrow + 7) / 8;
    int hr = wave * 8 + lrow;
    int hy = hr / Wp, hx = hr - hy * Wp;
#pragma unroll
    for (int j = 0; j < 13; ++j) {
      const int inst = j * 4 + wave;
      const int chunk = (lslot ^ (((inst & 1) << 2) | (lrow >> 1))) * 16;
      const int y = py0 - 1 + hy, x = hx - 1;
      const bool ok = hr < nrow && (unsigned)y < (unsigned)p.Hin && (unsigned)x < (unsigned)p.Win;
      const char* src = ok ? (const char*)p.A1 + ((((long)pb * p.Hin + y) * p.Win + x) * p.lda1) * sizeof(T) + chunk + (long)cc0 * 128
                           : zpage + (long)cc0 * 128;
      if constexpr (!W4_ABL(2)) { if (inst < ninst) glds16(src, smem + inst * 1024); }
      hr += 32; hx += 32;
      if (hx >= Wp) { hx -= Wp; ++hy; }
      if (hx >= Wp) { hx -= Wp; ++hy; }
    }
    w4_vm<0>();
  }
  __builtin_amdgcn_s_barrier();                               // B_raw(0): stage 0 (and the first image) has landed
  __builtin_amdgcn_sched_barrier(0);
  W4_STAMP(1);
  // the first reads of a tile, in the order the steady state issues them at the end of a stage (the counts of w4_sched assume it)
  auto first_reads = [&]() __attribute__((always_inline)) {
    w4_for<0, R>([&](auto Kc) {
      constexpr int k = decltype(Kc)::value;
      read_w(Kc);
      if constexpr (k == 0) { read_x(H0{}, std::integral_constant<int, 0>{}); read_x(H0{}, std::integral_constant<int, 1>{}); }
      if constexpr (k == 1) { read_x(H0{}, std::integral_constant<int, 2>{}); read_x(H0{}, std::integral_constant<int, 3>{}); }
    });
  };
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int f = 0; f < NW; ++f)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (f < 8) acc[f][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        else {
          // (the VGPR quads: left to the compiler they were zeroed ONCE, parked in scratch and reloaded per tile -- a vmcnt(0))
#pragma unroll
          for (int r = 0; r < 4; ++r) { float z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); acc[f][i][r] = z; }
        }
      }
    // (the accumulators were just written by v_accvgpr_write: wait states before the first MFMA reads them as SrcC)
    w4_for<0, NW>([&](auto Fc) { constexpr int f = decltype(Fc)::value; w4_pad<true, (f < 8)>(acc[f][0], acc[f][1], acc[f][2], acc[f][3]); });
  };

  // One stage = 2 NW groups of four MFMAs.  (slot / tap bookkeeping is done by the caller and passed BY VALUE: loop-carried
  // integers captured by reference and updated inside the unrolled body ended up in scratch memory.)  TAIL = false is a tile's
  // LAST stage: the reads behind group GB would be the next tile's first fragments -- they are issued after the epilogue instead
  // (first_reads), so that no fragment register is live while compiler-scheduled code runs.
  auto stage = [&](auto TAILc, int slot, int ntap, int nbuf) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(TAILc)::value;
    w4_for<0, G2>([&](auto Gc) {
      constexpr int g = decltype(Gc)::value, h = g / NW, f = g % NW;
      // (a tile's last stage issues nothing behind group GB: its groups there wait for what the steady state would allow minus
      // those reads -- conservatively, everything)
      if constexpr (!W4_ABL(1)) { if constexpr (TAIL || g <= GB) w4_lgkm<w4_allow<NW, g>()>(); else w4_lgkm<0>(); }
      w4_for<0, 4>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        if constexpr (!W4_ABL(32)) w4_mfma<(f < 8)>(acc[f][i], Wr[g % R], Xf[h][i]);
      });
      if constexpr (g == 0) {
        __builtin_amdgcn_s_barrier();                         // B_war(s-1)
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (g + R < G2) read_w(std::integral_constant<int, g + R>{});
      if constexpr (g == GB) {
        // every read of this stage is out: move to the next slot (and, HALO, to the next tap / image) -- the reads below
        // belong to stage s+1 and follow B_raw(s+1)
        __builtin_amdgcn_sched_barrier(0);
        const int d = (slot == W4_R - 1) ? -(W4_R - 1) * SLOT : SLOT;
        wa[0] += d; wa[1] += d;
        if constexpr (!HALO) { xa[0] += d; xa[1] += d; }
        else halo_addr(ntap, nbuf);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                         // B_raw(s+1)
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (TAIL) {
        if constexpr (g + R >= G2) read_w(std::integral_constant<int, g + R - G2>{});
        if constexpr (g == GB + 1) { read_x(H0{}, std::integral_constant<int, 0>{}); read_x(H0{}, std::integral_constant<int, 1>{}); }
        if constexpr (g == GB + 2) { read_x(H0{}, std::integral_constant<int, 2>{}); read_x(H0{}, std::integral_constant<int, 3>{}); }
      }
      if constexpr (g < 4) read_x(H1{}, std::integral_constant<int, g>{});
    });
  };

  // ---- a finished tile: accumulators -> memory.  Nothing here may wait for a global or scratch LOAD issued behind a store: vmcnt
  // retires in order, so such a wait drains the stores (measured: 12 us of a 59 us tile when every 8-column unit loaded its own
  // bias and the compiler parked addresses in scratch).  Hence: bias and row-bias come from LDS (issue_epi); the residual's 16
  // bytes per unit are fetched four units ahead; every unit re-derives its coordinates from the lane id (a handful of VALU)
  // instead of keeping twenty units' worth of addresses alive; accumulators leave the AGPR half by explicit v_accvgpr_read.
  auto epilogue = [&](int tm0, int tn0, int tz, int par) __attribute__((always_inline)) {
    // an MFMA result may be read 8 passes after its issue; nothing padded that for an asm MFMA
    w4_for<0, NW>([&](auto Fc) { constexpr int f = decltype(Fc)::value; w4_pad<false, (f < 8)>(acc[f][0], acc[f][1], acc[f][2], acc[f][3]); });
    if (W4_ABL(4) || p.act == 77) return;                     // act 77: timing probe only (skip the stores)
    EpiArgs e = epi_of(p);
    const char* eb = smem + EPI0 + par * 1024;
    const bool rb_lds = w4_rb_uniform(p, tm0);
    const int row0 = tm0 + wave * 64;                          // (uniform)
    const unsigned ldc_u = (unsigned)e.ldc, ldr_u = (unsigned)e.ldr, ldn_u = (unsigned)p.N;
    // lane (m = lr, lane row lg) holds columns 4 lg + r of every 16-column block
    auto lane_rc = [&](int& lr_, int& lg_) __attribute__((always_inline)) {
      int l2;                                                  // (volatile: every unit makes its own -- a shared copy went to scratch)
      asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l2));
      lr_ = l2 & 15; lg_ = l2 >> 4;
    };
    // v_permlane16_swap of blocks (fa, fb): lane rows 0 / 2 end up with columns 0-7 / 8-15 of block fa, lane rows 1 / 3 with
    // those of block fb -- 8 consecutive columns per lane
    auto swap8 = [&](const float (&a)[4], const float (&b)[4], float (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[r]), __float_as_uint(b[r]), false, false);
        v[r] = __uint_as_float(sw[0]); v[4 + r] = __uint_as_float(sw[1]);
      }
    };
    if (p.act == ACT_GEGLU && !slab) {
      // value columns [0, 80) of the tile pair with gate columns [80, 160): block f with block f + 5, SAME lane and register --
      // the products are formed in the accumulator layout, then blocks (0,1) (2,3) leave as 16-byte stores and block 4 as 8-byte ones
      if constexpr (NW == 10) {
        w4_for<0, 4>([&](auto Ic) {
          constexpr int i = decltype(Ic)::value;
          int lr, lg;
          lane_rc(lr, lg);
          const int grow = row0 + lr + 16 * i;
          const unsigned o_c = (unsigned)grow * ldc_u + (unsigned)(tn0 / 2);
          // the products of blocks (2 q, 2 q + 1), then block 4
          auto prod = [&](auto Fc, float (&P)[4]) __attribute__((always_inline)) {
            constexpr int f = decltype(Fc)::value;
            float bvv[4] = {0.f, 0.f, 0.f, 0.f}, bgg[4] = {0.f, 0.f, 0.f, 0.f};
            if (e.bias) {
              load4(reinterpret_cast<const float*>(eb + (16 * f + 4 * lg) * 4), bvv);
              load4(reinterpret_cast<const float*>(eb + (80 + 16 * f + 4 * lg) * 4), bgg);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
              P[r] = (w4_get<(f < 8)>(acc[f][i][r]) + bvv[r]) * gelu_f(w4_get<(f + 5 < 8)>(acc[f + 5][i][r]) + bgg[r]);
          };
          w4_for<0, 2>([&](auto Qc) {
            constexpr int q = decltype(Qc)::value;
            float Pa[4], Pb[4], v[8];
            prod(std::integral_constant<int, 2 * q>{}, Pa);
            prod(std::integral_constant<int, 2 * q + 1>{}, Pb);
            swap8(Pa, Pb, v);
            const unsigned c0 = 32 * q + 16 * (lg & 1) + 8 * (lg >> 1);
            if (grow < p.M) {
              if (p.out_f32) store8(reinterpret_cast<float*>(e.C) + (o_c + c0), v);
              else store8(reinterpret_cast<T*>(e.C) + (o_c + c0), v);
            }
            __builtin_amdgcn_sched_barrier(0);
          });
          {
            float P4[4];
            prod(std::integral_constant<int, 4>{}, P4);
            if (grow < p.M) {
              const unsigned c0 = 64 + 4 * lg;
              if (p.out_f32) store4(reinterpret_cast<float*>(e.C) + (o_c + c0), P4);
              else store4(reinterpret_cast<T*>(e.C) + (o_c + c0), P4);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      return;
    }
    constexpr int NQ = NW / 2;
    const bool pre = !slab && e.residual != nullptr;
    // Units in ROW-major order (16 rows x all the tile's column groups, then the next 16 rows): the two 64-byte halves of a
    // 128-byte line leave back to back.  Residual of unit (i, q): fetched when unit (i - 1, q) has consumed its own -- NQ units
    // ahead, one register quad per q.  (Unconditional loads from clamped coordinates: a conditionally written register would be
    // carried around the tile loop.)
    uint4 rs[NQ] = {};
    auto fetch = [&](auto Ic, auto Qc) __attribute__((always_inline)) {
      constexpr int q = decltype(Qc)::value, i = decltype(Ic)::value;
      int lr, lg;
      lane_rc(lr, lg);
      const int grow = row0 + lr + 16 * i, gcol = tn0 + 32 * q + 16 * (lg & 1) + 8 * (lg >> 1);
      const unsigned off = (grow < p.M && gcol < p.N) ? (unsigned)grow * ldr_u + (unsigned)gcol : 0u;
      rs[q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(e.residual) + off);
    };
    if (pre) w4_for<0, NQ>([&](auto Qc) { fetch(std::integral_constant<int, 0>{}, Qc); });
    w4_for<0, 4>([&](auto Ic) {
      constexpr int i = decltype(Ic)::value;
      w4_for<0, NQ>([&](auto Qc) {
        constexpr int q = decltype(Qc)::value;
        float fa[4], fb[4], v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { fa[r] = w4_get<(2 * q < 8)>(acc[2 * q][i][r]); fb[r] = w4_get<(2 * q + 1 < 8)>(acc[2 * q + 1][i][r]); }
        swap8(fa, fb, v);
        int lr, lg;
        lane_rc(lr, lg);
        const int cq = 32 * q + 16 * (lg & 1) + 8 * (lg >> 1);         // this lane's 8 columns inside the tile
        const int grow = row0 + lr + 16 * i, gcol = tn0 + cq;
        const bool in = grow < p.M && gcol < p.N;
        if (slab) {
          if (in) store8(slab + (((unsigned)tz * (unsigned)p.M + (unsigned)grow) * ldn_u + (unsigned)gcol), v);   // split-K partial
        } else {
          float b[8];
          if (e.bias) {
            load8(reinterpret_cast<const float*>(eb + cq * 4), b);     // (LDS: no vmcnt involved)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += b[k];
          }
          if (rb_lds) {
            load8(reinterpret_cast<const T*>(eb + BN * 4 + cq * 2), b);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += b[k];
          } else if (e.rowbias) {                                      // (a tile across batch elements: the slow way)
            if (in) load8(reinterpret_cast<const T*>(e.rowbias) + (long)(grow / e.rows_per_batch) * e.ldrb + gcol, b);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += b[k];
          }
          const uint4 cur = rs[q];
          if constexpr (i + 1 < 4) { if (pre) fetch(std::integral_constant<int, i + 1>{}, Qc); }
          unsigned off_out = (unsigned)grow * ldc_u + (unsigned)gcol;
          if constexpr (W4_ABL(8)) {   // probe only: the same bytes as ONE contiguous KB per store instruction (wrong placement)
            int l3;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l3));
            off_out = (unsigned)tm0 * ldc_u + (unsigned)(tn0 / BN) * (W4_BM * BN) + (unsigned)(((wave * 4 + i) * NQ + q) * 512 + l3 * 8);
          }
          if (in) epilogue8_tail_bf16(e, v, off_out, gcol, cur);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- the tiles of this workgroup
  int slot = 0;                                               // ring slot of the stage being read (runs on across tiles)
  int par = 0;                                                // parity of the tile (the epilogue operands' LDS block)
  for (int v = (int)blockIdx.x; v < nvirt; v += ngrid) {
    const Tile t = decode(v);
    const int tm0 = t.m0, tn0 = t.n0, tz = t.zsplit, tt = t.total;
    make_xh0();
    zero_acc();
    first_reads();
    __builtin_amdgcn_sched_barrier(0);
    auto advance = [&](auto TAILc) __attribute__((always_inline)) {
      const bool twrap = ctap == 8;
      const int ntap = twrap ? 0 : ctap + 1, nbuf = twrap ? (cbuf ^ 1) : cbuf;
      stage(TAILc, slot, ntap, nbuf);
      slot = slot == W4_R - 1 ? 0 : slot + 1; ctap = ntap; cbuf = nbuf;
    };
    for (int s = 0; s + 1 < tt; ++s) advance(std::true_type{});
    advance(std::false_type{});
    w4_lgkm<0>();                                             // (nothing of the consumers' own is in flight past here)
    __builtin_amdgcn_sched_barrier(0);
    if (v == (int)blockIdx.x) W4_STAMP(2);
    epilogue(tm0, tn0, tz, par);
    par ^= 1;
  }
#ifdef W4_PROBE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  W4_STAMP(3);
#endif
}

template <int NW, int MODE, int ABL = 0>
int launch_w4_mode(const GemmParams& p0, hipStream_t stream, bool persist = false) {
  constexpr int BN = 16 * NW;
  constexpr int SMEM = (MODE == W4_CONV_HALO ? 2 * W4_HROWS * 128 + W4_R * (BN / 8) * 1024 : W4_R * (W4_BM / 8 + BN / 8) * 1024) + 2048;
  auto kern = &gemm_w4_kernel<NW, MODE, ABL>;
  static bool attr_set = false;
  static int cus = 256;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess)
      return CL_ELAUNCH;
    int dev = 0; hipDeviceProp_t pr;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
      cus = pr.multiProcessorCount;
    attr_set = true;
  }
  if ((p0.a1_group_n && p0.a1_group_n % BN) || (p0.a2_group_n && p0.a2_group_n % BN)) return CL_EINVAL;   // a tile would straddle groups
  // the epilogue's element offsets are 32-bit
  if ((long)p0.M * p0.ldc >= (1L << 32) || (p0.residual && (long)p0.M * p0.ldr >= (1L << 32))) return CL_EINVAL;
  GemmParams p = p0;
  const int tm = (p.M + W4_BM - 1) / W4_BM, tn = (p.N + BN - 1) / BN;
  const long tiles = (long)tm * tn;
  const int steps = ((MODE == W4_LINEAR ? 1 : 9) * p.K1 + p.K2) / 64;
  float* slab;
  gemm_pick_splitk(p, tiles, steps, 256, 4, &slab, stream);
  const long nvirt = tiles * p.splitk;
  if (slab && (long)p.splitk * p.M * p.N >= (1L << 32)) return CL_EINVAL;
  // persistent form: one workgroup per CU (LDS admits no second one) walks virtual ids id, id + grid, ...; a multiple of 8 so that
  // id mod 8 -- the XCD -- is the same for every tile of a workgroup.  The halo form hands images over between WHOLE K ranges only.
  long grid = nvirt;
  // (>= 3 stages per tile and no split-K: the loaders run two stages ahead, and a tile's epilogue operands / image must not be
  // overwritten by the tile after next while it is still being read)
  if (persist && nvirt > cus && p.splitk == 1 && steps >= 3) grid = cus - cus % 8;
  gemm_tag_note(grid, 64 * (W4_NC + W4_NL));
  hipLaunchKernelGGL(kern, dim3((unsigned)(grid + gemm_cur_tag())), dim3(64 * (W4_NC + W4_NL)), SMEM, stream, p, tm, tn, slab, (int)nvirt,
                     (int)grid);
  if (slab) gemm_launch_splitk_reduce_bf16(p, slab, stream);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

// the halo-resident form covers a stride-1 conv whose every 256-row tile lies inside one image and whose image buffer fits
bool w4_halo_ok(const GemmParams& p) {
  if (p.mode != GEMM_CONV_S1 || p.Hin != p.Hout || p.Win != p.Wout) return false;
  if (p.Win > 64 || p.Win < 16 || (p.Win & (p.Win - 1)) || ((long)p.Hin * p.Win) % W4_BM || p.M % W4_BM) return false;
  return (W4_BM / p.Win + 2) * (p.Win + 2) <= W4_HROWS;
}

template <int NW>
int launch_w4_nw(const GemmParams& p, hipStream_t stream, bool persist) {
  if (p.mode == GEMM_LINEAR) return launch_w4_mode<NW, W4_LINEAR>(p, stream, persist);
  if (p.K2) return CL_EINVAL;   // a second K segment exists for linear operands only
#ifdef W4_PROBE
  if constexpr (NW == 10) {
    if (p.mode == GEMM_CONV_S1 && g_w4_abl_host) {
      const bool halo = g_w4_halo_host && w4_halo_ok(p);
      switch (g_w4_abl_host) {
#define W4_CASE(a) case a: return halo ? launch_w4_mode<NW, W4_CONV_HALO, a>(p, stream, persist) : launch_w4_mode<NW, W4_CONV_S1, a>(p, stream, persist);
        W4_CASE(1) W4_CASE(2) W4_CASE(3) W4_CASE(4) W4_CASE(5) W4_CASE(6) W4_CASE(7) W4_CASE(8) W4_CASE(37)
#undef W4_CASE
        default: break;
      }
    }
  }
  if (p.mode == GEMM_CONV_S1 && !g_w4_halo_host) return launch_w4_mode<NW, W4_CONV_S1>(p, stream, persist);
#endif
  if (p.mode == GEMM_CONV_S1)
    return w4_halo_ok(p) ? launch_w4_mode<NW, W4_CONV_HALO>(p, stream, persist) : launch_w4_mode<NW, W4_CONV_S1>(p, stream, persist);
  return launch_w4_mode<NW, W4_CONV_ANY>(p, stream, persist);
}

}  // namespace

#ifdef W4_PROBE
void w4_abl_set(int v) { g_w4_abl_host = v; }
void w4_halo_set(int v) { g_w4_halo_host = v; }
void w4_timing_set(unsigned long long* buf) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w4_timing), &buf, sizeof(buf)); }
#endif

// CL_EINVAL = "not a product this kernel covers" (the caller falls back to the other tile kernels)
int launch_gemm_w4(const GemmParams& p, hipStream_t stream, int bn, int persist) {
  if (p.atomic || p.K1 % 64 || p.K2 % 64 || p.K1 <= 0) return CL_EINVAL;
  if (p.act == ACT_GEGLU && bn != 160) return CL_EINVAL;
  if (p.act == ACT_GEGLU_SPLIT || p.ln_gamma) return CL_EINVAL;
  if (bn == 160) return launch_w4_nw<10>(p, stream, persist != 0);
  if (bn == 128) return launch_w4_nw<8>(p, stream, persist != 0);
  return CL_EINVAL;
}

}  // namespace cl
