// How much of the attention forward's softmax VALU can hide under its MFMAs on one gfx950 SIMD -- and in which structure?
// One "step" is the instruction mix of attn_fwd_hyb_kernel per wave and 64-key tile at d_head 40:
//   18 MFMAs   6 x v_mfma_f32_32x32x16_bf16 (two chains of 3) + 12 x v_mfma_f32_16x16x32_bf16 (6 accumulators, twice each)
//   104 VALU   32 v_max_f32 (one chain), 16 packed fma, 32 v_exp_f32, 16 v_cvt_pk_bf16_f32, 8 v_permlane16_swap
// with NO data dependence between the two streams (the ceiling: two independent query tiles per wave would give that).
// Modes: 0 MFMA only | 1 VALU only | 2 per wave 18 MFMAs then 104 VALU, no barrier | 3 per wave fine interleave (8 VALU
// behind every 32x32 MFMA, 4-5 behind every 16x16 MFMA) | 4 two wave groups one phase apart with s_barrier (the product's
// structure; needs >= 2 waves per SIMD).  Block sizes 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD, one block per CU.
// Output: ns and (at the clock calibrated by mode 0, which is 384 matrix cycles per step and wave) cycles per step and SIMD.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <type_traits>
#include <utility>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); sfor<I + 1, N>(f); }
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) { bf2 v; v.x = (__bf16)lo; v.y = (__bf16)hi; return __builtin_bit_cast(uint32_t, v); }

template <int ACC> __device__ __forceinline__ void mma32(f16v& c, const bf8& a, const bf8& b) {
  if constexpr (ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int ACC> __device__ __forceinline__ void mma16(f4& c, const bf8& a, const bf8& b) {
  if constexpr (ACC) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int ACC, typename V> __device__ __forceinline__ float first_of(V t) {
  if constexpr (ACC) asm volatile("" : "+a"(t));
  return t[0];
}

// VALU op k of the flattened softmax stream ends before op vend(i) once MFMA i has been issued (mode 3)
__host__ __device__ constexpr int vend(int i) { return i < 6 ? 8 * (i + 1) : 48 + ((i - 5) * 56) / 12; }

template <int MODE, int THREADS, int ACC = 0, int VMIX = 0, int MMIX = 0, int LMIX = 0> __global__ __launch_bounds__(THREADS < 512 ? 512 : THREADS) void k(float* out, int iters) {   // (a 256-thread bound makes the compiler park accumulators in AGPRs: v_accvgpr moves in the loop)
  extern __shared__ char smem[];
  bf8 a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(i * 0.5f); }
  f16v c32[2] = {}; f4 c16[6] = {};
  float s[32];
  for (int r = 0; r < 32; ++r) s[r] = (float)((threadIdx.x * 7 + r * 13) & 63) * 0.01f;
  float mx = 0.f, m_run = 0.1f; const float sl2 = 0.23f;
  f2 xg[16]; float e0g[16], e1g[16]; uint32_t pk[16]; uint32_t acc = 0;
  for (int i = 0; i < 16; ++i) { xg[i] = f2{0.f, 0.f}; e0g[i] = e1g[i] = 0.f; pk[i] = 0; }
  const int wave = threadIdx.x >> 6, grp = (wave >> 2) & 1;

  constexpr int NMF = MMIX == 0 ? 18 : MMIX == 1 ? 24 : 12;
  // LMIX 1: the A operand of every MFMA comes from LDS, read two MFMAs ahead (6 x ds_read_b128 for the 32x32 products, 12 x
  // ds_read_b64 pairs for the 16x16 ones: the product's K fragment and V^T transpose reads per step)
  // (LMIX 2: the same reads issued four MFMAs ahead instead of two)
  constexpr int RING = LMIX == 2 ? 6 : 3, AHEAD = LMIX == 2 ? 4 : 2;
  bf8 ring[RING];
  for (int i = 0; i < RING; ++i) ring[i] = a8;
  int itv = 0;
  auto lds_read = [&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    if constexpr (LMIX != 0) {
      const char* src = smem + (((itv * 3 + i) * 1024 + (int)(threadIdx.x & 63) * 16) & 0xfff0);
      constexpr int im = i % NMF;
      if constexpr (im < 6) ring[i % RING] = *reinterpret_cast<const bf8*>(src);          // K fragment: ds_read_b128
      else if constexpr ((im - 6) % 2 == 0) {                                               // V^T fragment: 2 x b64, feeds two MFMAs
        typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
        const bf4 lo = *reinterpret_cast<const bf4*>(src), hi = *reinterpret_cast<const bf4*>(src + 2048);
        ring[i % RING] = bf8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      } else ring[i % RING] = ring[(i - 1) % RING];
    }
  };
  auto mf = [&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    if constexpr (LMIX != 0) { lds_read(std::integral_constant<int, i + AHEAD>{}); a8 = ring[i % RING]; }
    constexpr bool big = MMIX == 0 ? (i < 6) : (MMIX == 2);
    if constexpr (big) {
      constexpr int j = MMIX == 0 ? i / 3 : i % 2;
      mma32<ACC>(c32[j], a8, b8);
    } else {
      constexpr int j = (MMIX == 0 ? i - 6 : i) % 6;
      mma16<ACC>(c16[j], a8, b8);
    }
  };
  auto vop = [&](auto Kc) {
    constexpr int kk = decltype(Kc)::value;
    if constexpr (kk < 32) mx = fmaxf(mx, s[kk]);
    else if constexpr (kk < 96) {
      constexpr int gi = (kk - 32) / 4, sub = (kk - 32) % 4;
      if constexpr (sub == 0) {
        if constexpr (VMIX == 3) xg[gi] = f2{s[2 * gi], s[2 * gi + 1]};
        else if constexpr (VMIX == 2) {
          float x0 = s[2 * gi] * sl2 - m_run, x1 = s[2 * gi + 1] * sl2 - m_run;
          asm volatile("" : "+v"(x0)); asm volatile("" : "+v"(x1));          // keep them from being re-packed
          xg[gi] = f2{x0, x1};
        } else xg[gi] = f2{s[2 * gi], s[2 * gi + 1]} * sl2 - m_run;
      }
      if constexpr (sub == 1) e0g[gi] = VMIX == 1 ? xg[gi].x * 1.0001f : __builtin_amdgcn_exp2f(xg[gi].x);
      if constexpr (sub == 2) e1g[gi] = VMIX == 1 ? xg[gi].y * 0.9999f : __builtin_amdgcn_exp2f(xg[gi].y);
      if constexpr (sub == 3) pk[gi] = pack2bf(e0g[gi], e1g[gi]);
    } else {
      constexpr int j = kk - 96, t = j / 4, q = j % 4;
      constexpr int a = 8 * t + (q < 2 ? q : q + 2), b = a + 2;            // (0,2) (1,3) (4,6) (5,7)
      const auto sw = __builtin_amdgcn_permlane16_swap(pk[a], pk[b], false, false);
      acc ^= sw[0] + sw[1];
    }
  };
  auto opaque = [&]() {            // the scores change every step as far as the compiler knows (no instructions)
#pragma unroll
    for (int r = 0; r < 32; ++r) asm volatile("" : "+v"(s[r]));
    asm volatile("" : "+v"(m_run));
  };
  auto phaseM = [&]() { sfor<0, NMF>([&](auto Ic) { mf(Ic); }); };
  auto phaseV = [&]() { sfor<0, 104>([&](auto Kc) { vop(Kc); }); };

  if constexpr (LMIX != 0) {
    for (int i = threadIdx.x; i < 100 * 1024 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0x3c003c00u + (i & 255);
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    itv = it;
    opaque();
    if constexpr (MODE == 0) phaseM();
    if constexpr (MODE == 1) phaseV();
    if constexpr (MODE == 2) { phaseM(); __builtin_amdgcn_sched_barrier(0); phaseV(); }
    if constexpr (MODE == 3) {
      sfor<0, NMF>([&](auto Ic) {
        constexpr int i = decltype(Ic)::value;
        mf(Ic);
        constexpr int lo = MMIX == 0 ? (i == 0 ? 0 : vend(i - 1)) : (104 * i) / NMF, hi = MMIX == 0 ? vend(i) : (104 * (i + 1)) / NMF;
        sfor<lo, hi>([&](auto Kc) { vop(Kc); });
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    if constexpr (MODE == 4) {
      if (grp == 0) { phaseM(); __builtin_amdgcn_s_barrier(); phaseV(); __builtin_amdgcn_s_barrier(); }
      else { phaseV(); __builtin_amdgcn_s_barrier(); phaseM(); __builtin_amdgcn_s_barrier(); }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float r = mx + (float)acc;
  for (int j = 0; j < 2; ++j) r += first_of<ACC>(c32[j]);
  for (int j = 0; j < 6; ++j) r += first_of<ACC>(c16[j]);
  out[blockIdx.x * 1024 + threadIdx.x] = r + (float)smem[0];
}

template <int MODE, int THREADS, int ACC = 0, int VMIX = 0, int MMIX = 0, int LMIX = 0> static int run(float* out, int iters, float* ms_out) {
  const int lds = 100 * 1024;      // one block per CU
  HIPCHK(hipFuncSetAttribute((const void*)k<MODE, THREADS, ACC, VMIX, MMIX, LMIX>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, THREADS, ACC, VMIX, MMIX, LMIX>), dim3(256), dim3(THREADS), lds, 0, out, iters);
    hipEventRecord(e1);
    HIPCHK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  *ms_out = best;
  return 0;
}

static float g_ref = 0.f;
static void report(int w, const char* name, float ms, int iters) {
  const double ns = ms * 1e6 / iters;
  if (g_ref == 0.f) g_ref = (float)ns;          // first call: MFMA only, one wave per SIMD = 384 matrix cycles
  const double cyc = ns / g_ref * 384.0;
  printf("%d waves/SIMD  %-46s %8.1f ns per step = %7.0f cycles per SIMD (%6.0f per wave-step)\n", w, name, ns, cyc, cyc / w);
}

int main() {
  float* out; HIPCHK(hipMalloc(&out, 256 * 1024 * 4)); HIPCHK(hipMemset(out, 0, 256 * 1024 * 4));
  const int iters = 3000;
  float ms;
  // ---- part 1: the product's mix, accumulators in VGPRs (compiler's choice at this occupancy)
  if (run<0, 256>(out, iters, &ms)) return 2; report(1, "MFMA only (18)", ms, iters);
  if (run<1, 256>(out, iters, &ms)) return 2; report(1, "VALU only (104)", ms, iters);
  if (run<2, 256>(out, iters, &ms)) return 2; report(1, "per wave: M then V", ms, iters);
  if (run<3, 256>(out, iters, &ms)) return 2; report(1, "per wave: interleaved", ms, iters);
  if (run<0, 1024>(out, iters, &ms)) return 2; report(4, "MFMA only (18)", ms, iters);
  if (run<1, 1024>(out, iters, &ms)) return 2; report(4, "VALU only (104)", ms, iters);
  if (run<2, 1024>(out, iters, &ms)) return 2; report(4, "per wave: M then V", ms, iters);
  if (run<3, 1024>(out, iters, &ms)) return 2; report(4, "per wave: interleaved", ms, iters);
  if (run<4, 1024>(out, iters, &ms)) return 2; report(4, "two groups, barriers (product structure)", ms, iters);
  // ---- part 5: scale and -max folded into the matrix product (no packed fma in the softmax): 88 VALU per step
  if (run<1, 1024, 0, 3>(out, iters, &ms)) return 2; report(4, "VALU only, no fma (88)", ms, iters);
  if (run<3, 1024, 0, 3>(out, iters, &ms)) return 2; report(4, "interleaved, no fma", ms, iters);
  if (run<3, 256, 0, 3>(out, iters, &ms)) return 2; report(1, "interleaved, no fma", ms, iters);
  // ---- part 6: with the LDS operand reads of the step (6 x b128 + 12 x 2 b64 per wave)
  if (run<0, 1024, 0, 0, 0, 1>(out, iters, &ms)) return 2; report(4, "MFMA + LDS reads", ms, iters);
  if (run<3, 1024, 0, 0, 0, 1>(out, iters, &ms)) return 2; report(4, "interleaved + LDS reads", ms, iters);
  if (run<4, 1024, 0, 0, 0, 1>(out, iters, &ms)) return 2; report(4, "two groups + barriers + LDS reads", ms, iters);
  if (run<3, 1024, 0, 3, 0, 1>(out, iters, &ms)) return 2; report(4, "interleaved, no fma + LDS reads", ms, iters);
  if (run<0, 1024, 0, 0, 0, 2>(out, iters, &ms)) return 2; report(4, "MFMA + LDS reads 4 ahead", ms, iters);
  if (run<3, 1024, 0, 0, 0, 2>(out, iters, &ms)) return 2; report(4, "interleaved + LDS reads 4 ahead", ms, iters);
  if (run<0, 512, 0, 0, 0, 1>(out, iters, &ms)) return 2; report(2, "MFMA + LDS reads", ms, iters);
  if (run<3, 512, 0, 0, 0, 1>(out, iters, &ms)) return 2; report(2, "interleaved + LDS reads", ms, iters);
  if (run<3, 512, 0, 3, 0, 2>(out, iters, &ms)) return 2; report(2, "interleaved, no fma + LDS reads 4 ahead", ms, iters);
  if (run<3, 256, 0, 3, 0, 2>(out, iters, &ms)) return 2; report(1, "interleaved, no fma + LDS reads 4 ahead", ms, iters);
  return 0;
}
