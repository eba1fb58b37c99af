// Consumer-stream microbenchmark (gfx950, not part of the product): ONE wave per SIMD runs the fragment-read + MFMA stream of the
// loader / consumer tile kernel (gemm_w4.hip) with nothing else on the chip -- no DMA, no barriers, no stores -- to price the
// stream itself.  Variants: MFMA shape (16x16x32: 14 reads per 40 MFMAs | 32x32x16: 7 reads per 10), operands taken from the reads
// (dependent, counted lgkmcnt) or from fixed registers (independent), ring depth / prefetch distance, LDS filled with random bf16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
template <int OFF> __device__ __forceinline__ void rd128(u32x4_t& v, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
template <bool AG> __device__ __forceinline__ void mfma16(f32x4_t& c, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// MODE 0: independent (MFMA operands are fixed registers, the reads land in the ring and are never used)
// MODE 1: dependent, counted waits (ring R for W, X double-buffered), exactly the product stream of one k-half = 10 groups
// MODE 2: dependent, but every group waits lgkmcnt(0) (all reads issued so far must land)
// MODE 3: no reads at all, operands random registers
template <int MODE, int R, int WAITK>
__global__ __launch_bounds__(256) void dep_kernel(unsigned long long* __restrict__ out, int iters, uint32_t seed) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // random bf16 pairs in [-1, 1) all over LDS
  for (int i = tid; i < 160 * 1024 / 4; i += 256) {
    uint32_t h = (uint32_t)(i * 2654435761u) ^ seed ^ (blockIdx.x * 40503u);
    h = h * 1664525u + 1013904223u; const float f = ((h >> 8) & 0xffff) / 32768.0f - 1.0f;
    h = h * 1664525u + 1013904223u; const float g = ((h >> 8) & 0xffff) / 32768.0f - 1.0f;
    reinterpret_cast<uint32_t*>(smem)[i] = (__float_as_uint(f) >> 16) | (__float_as_uint(g) & 0xffff0000u);
  }
  __syncthreads();
  const int lr = lane & 15, lg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t fo = lr * 128 + ((lg ^ ((lr >> 1) & 7)) * 16);
  uint32_t wa = lds0 + 32768 + fo, xa = lds0 + wave * 8192 + fo;
  f32x4_t acc[10][4];
#pragma unroll
  for (int f = 0; f < 10; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[f][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  u32x4_t Xf[2][4], Wr[R];
  // prologue: everything the first groups need
  sfor<0, R>([&](auto Kc) { rd128<decltype(Kc)::value * 2048>(Wr[decltype(Kc)::value], wa); });
  sfor<0, 4>([&](auto Ic) { rd128<decltype(Ic)::value * 2048>(Xf[0][decltype(Ic)::value], xa); rd128<decltype(Ic)::value * 2048>(Xf[1][decltype(Ic)::value], xa ^ 64u); });
  lgkm<0>();
  u32x4_t fa = Wr[0], fb = Xf[0][0];
  asm volatile("" : "+v"(fa), "+v"(fb));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    sfor<0, 20>([&](auto Gc) {                  // one stage = 20 groups (two k-halves of 10)
      constexpr int g = decltype(Gc)::value, h = g / 10, f = g % 10;
      if constexpr (MODE == 1) lgkm<(WAITK < 15 ? WAITK : 15)>();
      if constexpr (MODE == 2) lgkm<0>();
      sfor<0, 4>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        if constexpr (MODE == 0 || MODE == 3) mfma16<(f < 8)>(acc[f][i], fa, fb);
        else mfma16<(f < 8)>(acc[f][i], Wr[g % R], Xf[h][i]);
      });
      if constexpr (MODE != 3) {
        rd128<((g + R) % 10) * 2048>(Wr[g % R], ((g + R) % 20) < 10 ? wa : (wa ^ 64u));
        if constexpr (g < 4) rd128<g * 2048>(Xf[1][g], xa ^ 64u);
        if constexpr (g >= 10 && g < 14) rd128<(g - 10) * 2048>(Xf[0][g - 10], xa);
      }
    });
  }
  lgkm<0>();
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float s = 0.f;
#pragma unroll
  for (int f = 0; f < 10; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[f][i][0];
  if (lane == 0) { out[((long)blockIdx.x * 4 + wave) * 2] = t1 - t0; out[((long)blockIdx.x * 4 + wave) * 2 + 1] = (unsigned long long)(s + Wr[0].x + Xf[0][0].x + Xf[1][1].y); }
}

static unsigned long long* g_out;
template <int MODE, int R, int WAITK>
static void run(const char* tag) {
  auto kern = &dep_kernel<MODE, R, WAITK>;
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int iters = 400, grid = 256;
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  double cyc = 0; float wall = 0;
  for (int rep = 0; rep < 3; ++rep) {
    HIPCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 160 * 1024, 0, g_out, iters, 12345u + rep);
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); wall = ms;
    std::vector<unsigned long long> hbuf(grid * 8);
    HIPCHK(hipMemcpy(hbuf.data(), g_out, hbuf.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> t;
    for (int i = 0; i < grid * 4; ++i) t.push_back((double)hbuf[i * 2] / iters);
    std::sort(t.begin(), t.end());
    cyc = t[t.size() / 2];
  }
  const double flop = 4.0 * 256 * iters * 80 * 16384.0;
  printf("%-58s %8.1f cycles / stage (80 MFMAs: 1280 bare)  wall %7.1f us  %5.2f GHz  %7.1f TF/s\n", tag, cyc, wall * 1e3, cyc * iters / (wall * 1e3) * 1e-3,
         flop / (wall * 1e-3) * 1e-12);
}

int main() {
  HIPCHK(hipMalloc(&g_out, 256 * 8 * 8 + 64));
  run<3, 5, 4>("no reads, operands = two random fragments");
  run<0, 5, 4>("28 reads / stage, independent of the MFMAs");
  run<1, 5, 4>("dependent, ring 5, lgkmcnt(4)   (the product stream)");
  run<1, 5, 2>("dependent, ring 5, lgkmcnt(2)");
  run<1, 5, 8>("dependent, ring 5, lgkmcnt(8)   (too lax: WRONG data, timing only)");
  run<2, 5, 0>("dependent, ring 5, lgkmcnt(0) before every group");
  run<1, 8, 7>("dependent, ring 8, lgkmcnt(7)");
  run<1, 10, 9>("dependent, ring 10, lgkmcnt(9)");
  run<1, 3, 2>("dependent, ring 3, lgkmcnt(2)");
  return 0;
}
