// What does a SIMD's matrix pipe lose to the data movement beside it?  (gfx950 microbenchmark, not part of the product.)
//
// A workgroup of 8 waves, one per CU: waves 0-3 (one per SIMD) issue a stream of independent v_mfma_f32_32x32x16_bf16 and,
// optionally, one "self" operation behind every SELF_EVERY-th MFMA; waves 4-7 (their SIMD partners) issue "aux" operations at a
// chosen rate (one per AUX_GAP cycles of s_sleep-free spinning is not controllable, so: AUX_N operations per MFMA batch, the
// batches delimited by a barrier).  The MFMA waves time themselves with s_memtime.  Every combination is a template instance.
//
//   op 0 none | 1 ds_read_b128 -> VGPR | 2 ds_read_b128 -> AGPR | 3 global_load_lds_dwordx4 (LDS-DMA, 1 KiB) | 4 global_load_dwordx4
//   -> VGPR | 5 ds_write_b128 | 6 global_load_dwordx4 + ds_write_b128 of the previous one | 7 global_load_lds_dword (256 B)
//   | 8 ds_read_b64 | 9 v_mov x4 (VALU) | 10 buffer_load_dwordx4 ... lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}

template <int OFF> __device__ __forceinline__ void ds_w128(uint32_t addr, const u32x4_t& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void gload128(u32x4_t& v, const char* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
__device__ __forceinline__ void pinv(u32x4_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void mfma_a(f32x16_t& c, const u32x4_t& a, const u32x4_t& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma16(f32x4_t& c, const u32x4_t& a, const u32x4_t& b) { asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
__device__ __forceinline__ void mfma_v(f32x16_t& c, const u32x4_t& a, const u32x4_t& b) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }

constexpr int NMF = 40;          // MFMAs per batch (one "stage")

template <int OP> __device__ __forceinline__ void do_op(u32x4_t& r, u32x4_t& r2, uint32_t lds_addr, const char* gsrc, char* lds_wave_base) {
  if constexpr (OP == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(lds_addr) : "memory");
  else if constexpr (OP == 2) asm volatile("ds_read_b128 %0, %1" : "=a"(r) : "v"(lds_addr) : "memory");
  else if constexpr (OP == 3) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
  else if constexpr (OP == 4) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r) : "v"(gsrc) : "memory");
  else if constexpr (OP == 5) asm volatile("ds_write_b128 %0, %1" :: "v"(lds_addr), "v"(r) : "memory");
  else if constexpr (OP == 6) {
    asm volatile("ds_write_b128 %0, %1" :: "v"(lds_addr), "v"(r2) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(r) : "v"(gsrc) : "memory");
  }
  else if constexpr (OP == 7) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
  else if constexpr (OP == 8) { u32x2_t t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"(lds_addr) : "memory"); r.x = t.x; r.y = t.y; }
  else if constexpr (OP == 9) asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1" : "=v"(r.x) : "v"(lds_addr));
}

// SELF: op the MFMA waves issue themselves behind every SELF_EVERY-th MFMA.  AUX: op of the partner waves, AUX_N per batch.
// ACCV: accumulators in architectural registers ("v") instead of accumulation registers ("a").
template <int SELF, int SELF_EVERY, int AUX, int AUX_N, bool ACCV, int MF16 = 0>
__global__ __launch_bounds__(512) void simd_kernel(const char* __restrict__ gsrc, unsigned long long* __restrict__ out, int iters, int rnd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // conflict-free 32-row x 128-byte fragment address (source-swizzled image, as the product kernels use)
  const int l31 = lane & 31, hi = lane >> 5;
  const uint32_t faddr = lds0 + (wave & 3) * 8192 + l31 * 128 + ((hi ^ ((l31 >> 1) & 7)) * 16);
  const uint32_t waddr = lds0 + 65536 + (wave & 3) * 16384 + lane * 16;     // lane-linear store image
  const char* src = gsrc + ((long)blockIdx.x * 8 + wave) * 65536 + lane * 16;
  char* dma_base = smem + 65536 + (wave & 3) * 16384;
  if (wave < 4) {
    f32x16_t acc[8];
    f32x4_t acc16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc16[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4_t a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a, r = a, r2 = a;
    u32x4_t av[4], bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      av[q] = a; bv[q] = b;
      if (rnd) {     // bf16 pairs uniform in [-1, 1): random sign, exponent 0x3e..0x3f-ish, random mantissa
        uint32_t h = (uint32_t)(tid * 2654435761u) ^ (uint32_t)(blockIdx.x * 40503u) ^ (q * 0x9e3779b9u);
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          h = h * 1664525u + 1013904223u;
          const float f = ((h >> 8) & 0xffff) / 32768.0f - 1.0f;
          const uint32_t lo = __float_as_uint(f) >> 16;
          h = h * 1664525u + 1013904223u;
          const float g = ((h >> 8) & 0xffff) / 32768.0f - 1.0f;
          w[e] = lo | (__float_as_uint(g) & 0xffff0000u);
        }
        av[q] = u32x4_t{w[0], w[1], w[2], w[3]}; bv[q] = u32x4_t{w[4], w[5], w[6], w[7]};
      }
      asm volatile("" : "+v"(av[q]), "+v"(bv[q]));
    }
    asm volatile("" : "+v"(a), "+v"(b));
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      sfor<0, NMF>([&](auto Mc) {
        constexpr int m = decltype(Mc)::value;
        if constexpr (MF16) { mfma16(acc16[(2 * m) & 15], av[m & 3], bv[(m >> 2) & 3]); mfma16(acc16[(2 * m + 1) & 15], av[(m + 1) & 3], bv[(m >> 2) & 3]); }
        else if constexpr (ACCV) mfma_v(acc[m & 7], av[m & 3], bv[(m >> 2) & 3]); else mfma_a(acc[m & 7], av[m & 3], bv[(m >> 2) & 3]);
        if constexpr (SELF == 20) asm volatile("s_nop 7" ::: "memory");
        if constexpr (SELF == 21) asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
        if constexpr (SELF == 22) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
        if constexpr (SELF != 0 && SELF < 20 && (m % SELF_EVERY) == 0) do_op<SELF>(r, r2, faddr, src + (m & 15) * 1024, dma_base + (m & 15) * 1024);
      });
      if constexpr (SELF == 1 || SELF == 2 || SELF == 5 || SELF == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (SELF == 3 || SELF == 4 || SELF == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (SELF == 2) asm volatile("" : "+a"(r)); else asm volatile("" : "+v"(r));
      __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc16[i][0];
    if (lane == 0) { out[((long)blockIdx.x * 4 + wave) * 2] = t1 - t0; out[((long)blockIdx.x * 4 + wave) * 2 + 1] = (unsigned long long)(s + r.x); }
  } else {
    u32x4_t r = {1u, 2u, 3u, 4u}, r2 = r;
    u32x4_t buf[AUX_N];
#pragma unroll
    for (int k = 0; k < AUX_N; ++k) buf[k] = r;
    asm volatile("" : "+v"(r), "+v"(r2));
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
      if constexpr (AUX == 6) {
        // register-staged fill: the loads of this batch go out, then the previous batch's registers are stored to LDS
        sfor<0, AUX_N>([&](auto Kc) {
          constexpr int k = decltype(Kc)::value;
          ds_w128<(k & 3) * 1024>(waddr, buf[k]);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the stores have read their registers before the loads overwrite them)
        sfor<0, AUX_N>([&](auto Kc) {
          constexpr int k = decltype(Kc)::value;
          gload128(buf[k], src + (k & 15) * 1024);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sfor<0, AUX_N>([&](auto Kc) { pinv(buf[decltype(Kc)::value]); });
      } else {
        sfor<0, AUX_N>([&](auto Kc) {
          constexpr int k = decltype(Kc)::value;
          if constexpr (AUX != 0) do_op<AUX>(r, r2, AUX == 5 ? waddr : faddr, src + (k & 15) * 1024, dma_base + (k & 15) * 1024);
        });
      }
      if constexpr (AUX == 1 || AUX == 2 || AUX == 5 || AUX == 6 || AUX == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (AUX == 3 || AUX == 4 || AUX == 6 || AUX == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if constexpr (AUX == 2) asm volatile("" : "+a"(r)); else asm volatile("" : "+v"(r));
      __builtin_amdgcn_s_barrier();
    }
    if (lane == 0 && r.x == 0xdeadbeefu) out[0] = r.x + r2.y;
  }
}

static const char* opname(int op) {
  static const char* n[] = {"none", "ds_read_b128->v", "ds_read_b128->a", "LDS-DMA dwordx4", "global_load_dwordx4", "ds_write_b128",
                            "global_load_x4 + ds_write_b128", "LDS-DMA dword", "ds_read_b64", "4 x v_mov"};
  return op >= 0 && op < 10 ? n[op] : "s_nop pad";
}

static char* g_src; static unsigned long long* g_out; static double g_base = 0; static int g_rnd = 0;

template <int SELF, int SELF_EVERY, int AUX, int AUX_N, bool ACCV, int MF16 = 0>
static void run(const char* tag) {
  auto kern = &simd_kernel<SELF, SELF_EVERY, AUX, AUX_N, ACCV, MF16>;
  HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int iters = 400, grid = 256;
  std::vector<double> med; float wall_ms = 0;
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    HIPCHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 160 * 1024, 0, g_src, g_out, iters, g_rnd);
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); if (rep == 1 || wall_ms == 0) wall_ms = ms;
    HIPCHK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(grid * 8);
    HIPCHK(hipMemcpy(h.data(), g_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> t;
    for (int i = 0; i < grid * 4; ++i) t.push_back((double)h[i * 2] / iters);
    std::sort(t.begin(), t.end());
    med.push_back(t[t.size() / 2]);
  }
  std::sort(med.begin(), med.end());
  const double cyc = med[1];       // s_memtime ticks per batch of NMF MFMAs (100 MHz constant clock or shader clock: compare to the base line)
  const int nself = SELF ? (NMF + SELF_EVERY - 1) / SELF_EVERY : 0;
  if (g_base == 0) g_base = cyc;
  const int nops = nself + (AUX ? AUX_N : 0);
  printf("%-34s self %-22s x%2d | aux %-30s x%2d | %8.1f ticks / %d MFMAs = %6.2f per MFMA | + %7.1f over bare = %6.2f per op (%s)\n", tag,
         opname(SELF), nself, opname(AUX), AUX ? AUX_N : 0, cyc, NMF, cyc / NMF, cyc - g_base, nops ? (cyc - g_base) / nops : 0.0,
         ACCV ? "acc in v" : "acc in a");
  const double flop = 4.0 * 256 * iters * NMF * 32768.0;
  printf("%-34s      wall %8.1f us  -> %5.2f GHz effective, %7.1f TF/s\n", "", wall_ms * 1e3, cyc * iters / (wall_ms * 1e3) * 1e-3, flop / (wall_ms * 1e-3) * 1e-12);
}

int main() {
  HIPCHK(hipMalloc(&g_src, 256L * 8 * 65536 + 65536)); HIPCHK(hipMemset(g_src, 0x3f, 256L * 8 * 65536 + 65536));
  HIPCHK(hipMalloc(&g_out, 256 * 8 * 8 + 64));
  printf("ticks are s_memtime units; 'bare' = the first line\n");
  for (g_rnd = 0; g_rnd < 2; ++g_rnd) {
    printf("======== operands: %s\n", g_rnd ? "uniform random bf16 in [-1, 1)" : "constant 1.0");
    run<0, 1, 0, 1, false>("bare MFMA stream");
    run<0, 1, 0, 1, false>("bare MFMA stream again");
    run<0, 1, 0, 1, true>("bare, accumulators in VGPRs");
    run<0, 1, 0, 1, true, 1>("bare 16x16x32 (2 per slot)");
    run<1, 1, 0, 1, false>("self ds_read_b128 every MFMA");
    run<1, 2, 0, 1, false>("self ds_read_b128 every 2nd");
    run<3, 4, 0, 1, false>("self LDS-DMA x4 every 4th");
    run<3, 3, 0, 1, false>("self LDS-DMA x4 every 3rd");
    run<4, 4, 0, 1, false>("self global_load_x4 every 4th");
  }
  return 0;
}
