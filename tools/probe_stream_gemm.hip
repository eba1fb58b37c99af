// Probe (not part of the product): how fast can y = x W^T go for the K <= 1 280 products of the 64x64 level if NOTHING runs in
// lock-step?  DESIGN.md 6b.2: the product's kernels run (32768, 320, 320) in 16.7 us -- one wave of 256 workgroups that all load,
// then all compute, then all store -- against ~11 us for a one-wave grid that overlaps its phases and an 8.4 us HBM floor.
//
// Structure tried here ("A-stationary, W streamed"): a workgroup of 4 waves owns 128 rows of x and the WHOLE output width.
//   * every wave loads its 32 rows of x (32 x K bf16) straight into registers ONCE, as the 20 B-operand fragments of
//     v_mfma_f32_32x32x16_bf16 (K = 320: 80 VGPRs) -- x never touches LDS and is read from HBM exactly once;
//   * W streams through LDS in chunks of 32 output columns (32 x K bf16 = 20 KB), LDS-DMA three chunks deep, XOR-swizzled on the
//     SOURCE side so that the fragment reads (ds_read_b128, row n, 16-byte slot) are conflict-free;
//   * per chunk and wave: 20 MFMAs  C^T[32 n x 32 m] += W_chunk[32 n x 16 k] . x^T[16 k x 32 m]  -- the transposed product, so a
//     lane ends up with 4 CONSECUTIVE output columns of one row (8-byte stores) -- and the chunk is stored while the next
//     one is multiplied (two accumulator sets): loads, MFMAs and stores of a workgroup overlap, nothing waits for a grid-wide phase.
// Output: correctness of sampled rows against a double-precision CPU product, then the HIP-event time per launch.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
template <int OFF> __device__ __forceinline__ u32x4_t lds_rd128(uint32_t addr) {
  u32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v; v.x = (__bf16)lo; v.y = (__bf16)hi;
  return __builtin_bit_cast(uint32_t, v);
}

// s_waitcnt vmcnt(n) for a run-time n in [0, 24] (the count is an instruction immediate)
__device__ __forceinline__ void vm_wait(int n) {
  switch (n) {
#define VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    VMW(0) VMW(1) VMW(2) VMW(3) VMW(4) VMW(5) VMW(6) VMW(7) VMW(8) VMW(9) VMW(10) VMW(11) VMW(12) VMW(13) VMW(14) VMW(15) VMW(16)
    VMW(17) VMW(18) VMW(19) VMW(20) VMW(21) VMW(22) VMW(23) VMW(24)
#undef VMW
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}

// K = 16 * KS elements; a workgroup owns 128 rows x (32 * NCH) columns starting at column 32 * NCH * blockIdx.y of an output
// that is N = 32 * NCH * gridDim.y wide; M = 128 * gridDim.x rows.  WIDE: T21 of the CDNA4 guide -- v_permlane32_swap pairs the
// half-waves' 4-column groups so that every lane stores 16 bytes (2 dwordx4 instead of 4 dwordx2 per chunk).  MINW: minimum
// waves per SIMD the register budget is sized for (2 = two workgroups per CU when the ring fits twice in the LDS).
template <int KS, int NCH, bool WIDE, int MINW>
__global__ __launch_bounds__(256, MINW) void stream_gemm_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ W0,
                                                                uint16_t* __restrict__ Y0) {
  constexpr int K = 16 * KS, ROWB = K * 2, CPRW = ROWB / 16;   // bytes / 16-byte slots per W row
  const int N = 32 * NCH * (int)gridDim.y;
  const uint16_t* W = W0 + (long)blockIdx.y * 32 * NCH * K;
  uint16_t* Y = Y0 + (long)blockIdx.y * 32 * NCH;
  constexpr int CHUNK = 32 * ROWB, PIECES = CHUNK / 1024, NJ = (PIECES + 3) / 4, RING = 3;
  static_assert(CPRW % 8 == 0, "the source-side swizzle permutes slots inside aligned groups of 8");
  extern __shared__ __attribute__((aligned(16))) char smem[];          // RING chunks of W
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const long row0 = (long)blockIdx.x * 128 + wave * 32;

  // ---- x rows of this wave -> registers (B operand: col = row l31 of the wave's block, k = 16 j + 8 hi .. + 7)
  u32x4_t xa[KS];
  {
    const char* xp = (const char*)X + (row0 + l31) * ROWB + hi * 16;
#pragma unroll
    for (int j = 0; j < KS; ++j) xa[j] = *reinterpret_cast<const u32x4_t*>(xp + j * 32);
  }
  // ---- W chunk DMA: piece p of a chunk image = 64 slots of 16 bytes; slot q = 64 p + lane sits in row n = q / CPRW at physical
  // slot s = q % CPRW and holds LOGICAL slot s ^ swz(n) of that W row (swz permutes within aligned groups of 8 slots)
  int woff[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int q = (wave + 4 * j) * 64 + lane, n = q / CPRW, s = q - n * CPRW;
    woff[j] = n * ROWB + ((s ^ ((n >> 1) & 7)) * 16);
  }
  auto issue = [&](int c) {
    const char* wb = (const char*)W + (long)c * 32 * ROWB;
    char* dst = smem + (c % RING) * CHUNK;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
      if (wave + 4 * j < PIECES) glds16(wb + woff[j], dst + (wave + 4 * j) * 1024);
  };
  static_assert(PIECES % 4 == 0, "every wave issues the same number of DMA instructions per chunk");
  constexpr int DMA_PER_CHUNK = PIECES / 4;
  // W fragment (A operand: row = output column l31 of the chunk, k = 16 j + 8 hi): logical slot 2 j + hi of row l31
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t sw = (l31 >> 1) & 7;
  uint32_t wrow[KS];          // byte offset of the fragment of k-step j inside a chunk image (loop-invariant)
#pragma unroll
  for (int j = 0; j < KS; ++j) wrow[j] = l31 * ROWB + (((2 * j + hi) ^ sw) * 16);

  f32x16_t acc[2];
  issue(0);
  if (NCH > 1) issue(1);
  uint16_t* yrow = Y + (row0 + l31) * N + 4 * hi;      // lane (m = l31, hi): columns 8 (r / 4) + 4 hi + (r % 4) of the chunk
  uint16_t* yrow_w = Y + (row0 + l31) * N + 8 * hi;   // WIDE: lanes 0-31 store columns 16 p .. + 7, lanes 32-63 columns 16 p + 8 .. + 15
  auto store_chunk = [&](int c, const f32x16_t& a) {
    if constexpr (WIDE) {
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
        uint32_t ax = pack2bf(a[8 * p2], a[8 * p2 + 1]), ay = pack2bf(a[8 * p2 + 2], a[8 * p2 + 3]);          // group k = 2 p2
        uint32_t bx = pack2bf(a[8 * p2 + 4], a[8 * p2 + 5]), by = pack2bf(a[8 * p2 + 6], a[8 * p2 + 7]);      // group k + 1
        auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
        auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
        u32x4_t w = {rx[0], ry[0], rx[1], ry[1]};
        *reinterpret_cast<u32x4_t*>(yrow_w + c * 32 + 16 * p2) = w;
      }
    } else {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        u32x2_t w = {pack2bf(a[4 * q4], a[4 * q4 + 1]), pack2bf(a[4 * q4 + 2], a[4 * q4 + 3])};
        *reinterpret_cast<u32x2_t*>(yrow + c * 32 + 8 * q4) = w;
      }
    }
  };
  constexpr int SPC = WIDE ? 2 : 4;     // store instructions per chunk
  for (int c = 0; c < NCH; ++c) {
    // This wave's requests of chunk c have landed when at most the vector-memory operations issued AFTER them are outstanding
    // (gfx9 family: one vmcnt for loads and stores, retired in issue order -- the compiler's own waitcnt insertion relies on it):
    // the requests of chunk c + 1 (issued one iteration ago) and the 4 stores each of chunks c - 3 and c - 2.
    vm_wait((c + 1 < NCH ? DMA_PER_CHUNK : 0) + (c >= 2 ? SPC : 0) + (c >= 3 ? SPC : 0));
    __builtin_amdgcn_s_barrier();                      // every wave's pieces landed; chunk c - 1's buffer is free
    if (c + 2 < NCH) issue(c + 2);
    const uint32_t base = lds0 + (c % RING) * CHUNK;
    f32x16_t a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // fragment reads four k-steps ahead of their MFMAs
    u32x4_t wf[KS];
    constexpr int AHEAD = 4;
    sfor<0, AHEAD>([&](auto J) { constexpr int j = decltype(J)::value; wf[j] = lds_rd128<0>(base + wrow[j]); });
    sfor<0, KS>([&](auto J) {
      constexpr int j = decltype(J)::value;
      if constexpr (j + AHEAD < KS) { wf[j + AHEAD] = lds_rd128<0>(base + wrow[j + AHEAD]); lgkm_wait<AHEAD>(); }
      else lgkm_wait<KS - 1 - j>();
      a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wf[j]), __builtin_bit_cast(bf16x8_t, xa[j]), a, 0, 0, 0);
    });
    acc[c & 1] = a;
    if (c > 0) store_chunk(c - 1, acc[(c - 1) & 1]);   // the previous chunk's stores go out behind this chunk's MFMAs
  }
  store_chunk(NCH - 1, acc[(NCH - 1) & 1]);
}

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }

template <int KS, int NCH, bool WIDE = false, int MINW = 1> static int run(int M, int nsplit = 1) {
  constexpr int K = 16 * KS;
  const int N = 32 * NCH * nsplit;
  constexpr int LDS = 3 * 32 * K * 2;
  std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K), hy((size_t)M * N);
  uint32_t st = 12345;
  auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hx) v = f2bf(rnd());
  for (auto& v : hw) v = f2bf(rnd() * 0.1f);
  uint16_t *dx, *dw, *dy;
  HIPCHK(hipMalloc(&dx, hx.size() * 2)); HIPCHK(hipMalloc(&dw, hw.size() * 2)); HIPCHK(hipMalloc(&dy, hy.size() * 2));
  HIPCHK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(dy, 0xff, hy.size() * 2));
  auto kern = &stream_gemm_kernel<KS, NCH, WIDE, MINW>;
  if (LDS > 65536) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  hipLaunchKernelGGL(kern, dim3(M / 128, nsplit), dim3(256), LDS, 0, dx, dw, dy);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(hy.data(), dy, hy.size() * 2, hipMemcpyDeviceToHost));
  double num = 0, den = 0; int bad = 0;
  for (int t = 0; t < 24; ++t) {
    const int r = (int)(((long)t * 7919 + 13) % M);
    for (int n = 0; n < N; ++n) {
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)bf2f(hx[(size_t)r * K + k]) * bf2f(hw[(size_t)n * K + k]);
      const double got = bf2f(hy[(size_t)r * N + n]);
      num += (got - ref) * (got - ref); den += ref * ref;
      if (!(std::fabs(got - ref) <= 0.02 * std::fabs(ref) + 0.02)) ++bad;
    }
  }
  const double rel = std::sqrt(num / (den + 1e-30));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(M / 128, nsplit), dim3(256), LDS, 0, dx, dw, dy);
  HIPCHK(hipDeviceSynchronize());
  float best = 1e9f, sum = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(kern, dim3(M / 128, nsplit), dim3(256), LDS, 0, dx, dw, dy);
    hipEventRecord(e1);
    HIPCHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best; sum += ms;
  }
  const double us = best / 50 * 1e3, flops = 2.0 * M * N * K, bytes = 2.0 * ((double)M * K + (double)M * N + (double)N * K);
  printf("[%s] wide=%d minw=%d nsplit=%d (M, N, K) = (%d, %d, %d): rel-L2 %.2e, %d of %d sampled elements off;  %.2f us per launch (mean %.2f), %.0f TF/s, %.2f TB/s algorithmic\n",
         (bad == 0 && rel < 5e-3) ? "PASS" : "FAIL", (int)WIDE, MINW, nsplit, M, N, K, rel, bad, 24 * N, us, sum / 250 * 1e3, flops / us * 1e-6, bytes / us * 1e-6);
  hipFree(dx); hipFree(dw); hipFree(dy);
  return 0;
}

int main() {
  if (run<20, 10>(32768)) return 2;     // (32768, 320, 320): 53 launches per training step at 16.7-17.8 us in the product
  if (run<20, 10, true>(32768)) return 2;
  if (run<20, 10, true, 2>(32768)) return 2;
  // round 5: the wide-N / short-K class: GEGLU projection (32768, 2560, 320) 121 us in the product
  if (run<20, 80>(32768)) return 2;
  if (run<20, 80, true>(32768)) return 2;
  if (run<20, 80, true, 2>(32768)) return 2;
  if (run<20, 40, true>(32768, 2)) return 2;          // the same product, output width split over two workgroups (512 workgroups)
  if (run<20, 40, true, 2>(32768, 2)) return 2;
  if (run<20, 40, true>(32768)) return 2;             // (32768, 1280, 320): 50.5 us
  if (run<20, 30, true>(32768)) return 2;             // (32768, 960, 320): 39.6 us
  if (run<20, 40, true, 2>(32768)) return 2;
  if (run<20, 30, true, 2>(32768)) return 2;
  if (run<40, 40, true>(8192, 4)) return 2;           // (8192, 5120, 640): 91 us; 64 row blocks x 4 column splits
  if (run<40, 20, true>(8192, 8)) return 2;
  if (run<40, 20, true>(8192, 4)) return 2;           // (8192, 2560, 640): 40.5 us
  if (run<40, 20, true>(8192, 1)) return 2;           // (8192, 640, 640)
  if (run<40, 5, true>(8192, 4)) return 2;            // the same, 256 workgroups
  if (run<20, 80, true>(131072)) return 2;            // DDIM batch GEGLU projection
  return 0;
}
