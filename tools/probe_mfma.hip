// Issue-rate probe of the bf16 MFMA shapes on gfx950 (one wave per SIMD, 4 independent accumulators):
// v_mfma_f32_16x16x32_bf16 (K = 32), the carried-forward v_mfma_f32_16x16x16_bf16 (K = 16), 32x32x16 and 32x32x8.
// Question it answers: does the K = 16 form cost half the K = 32 form?  (d_head 40 = 32 + 8: a 48-deep walk instead of 64.)
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;

template <int MODE> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  bf8 a8, b8; s4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(threadIdx.x * 0.001f + i); b8[i] = (__bf16)(i * 0.5f); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(threadIdx.x + i); b4[i] = (short)(0x3f80 + i); }
  f4 c[4] = {}; f16v d[4] = {};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if constexpr (MODE == 0) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, c[j], 0, 0, 0);
      if constexpr (MODE == 1) c[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, c[j], 0, 0, 0);
      if constexpr (MODE == 2) d[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, d[j], 0, 0, 0);
      if constexpr (MODE == 3) d[j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, d[j], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int j = 0; j < 4; ++j) { s += c[j][0] + d[j][0]; }
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; long long* cyc; HIPCHK(hipMalloc(&out, 256 * 256 * 4)); HIPCHK(hipMalloc(&cyc, 8));
  const int iters = 4096;
  const char* names[4] = {"16x16x32_bf16", "16x16x16_bf16_1k", "32x32x16_bf16", "32x32x8_bf16_1k"};
  for (int m = 0; m < 4; ++m) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
      hipEventRecord(e1);
      HIPCHK(hipDeviceSynchronize());
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; HIPCHK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double n = (double)iters * 4;
    printf("%-20s %8.2f clk-counter ticks per MFMA per wave, %7.3f ms for %d x 4 MFMAs per wave (256 WGs x 4 waves)\n", names[m],
           (double)c / n, ms, iters);
  }
  return 0;
}
