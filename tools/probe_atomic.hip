// Throughput probe: fp32 global atomics in the access pattern of a fused attention backward -- every workgroup adds
// [64 rows x 40 floats] tiles (16 lanes = 64 contiguous bytes per row piece, 4 rows per instruction) into a [B*N, 320] fp32
// matrix, each matrix element being hit `passes` times by different workgroups.  Compares against plain stores of the
// same pattern.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

// grid: (key blocks kb, heads h, batch b); loop over the N / 64 query tiles, start skewed by kb
template <int MODE> __global__ __launch_bounds__(256) void k(float* acc, int N, int H, int nkb) {
  const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, lq = lane & 15;
  const int nt = N / 64;
  const long ld = (long)H * 40;
  for (int tt = 0; tt < nt; ++tt) {
    const int t = (tt + kb * (nt / nkb)) % nt;
    float* base = acc + ((long)b * N + t * 64 + wave * 16 + 4 * g) * ld + h * 40 + lq;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (16 * i + lq < 40) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = (float)(kb + r) * 1e-3f;
          if (MODE == 0) unsafeAtomicAdd(base + r * ld + 16 * i, v);
          else if (MODE == 1) atomicAdd(base + r * ld + 16 * i, v);
          else base[r * ld + 16 * i] = v;
        }
      }
    }
  }
}

int main() {
  const int B = 8, H = 8, N = 4096, nkb = 32;
  float* acc; const size_t bytes = (size_t)B * N * H * 40 * 4;
  HIPCHK(hipMalloc(&acc, bytes)); HIPCHK(hipMemset(acc, 0, bytes));
  const char* names[3] = {"unsafeAtomicAdd(f32)", "atomicAdd(f32)", "plain store"};
  for (int m = 0; m < 3; ++m) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(nkb, H, B), dim3(256), 0, 0, acc, N, H, nkb);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(nkb, H, B), dim3(256), 0, 0, acc, N, H, nkb);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(nkb, H, B), dim3(256), 0, 0, acc, N, H, nkb);
      hipEventRecord(e1);
      HIPCHK(hipDeviceSynchronize());
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double elems = (double)B * H * N * 40 * nkb;
    printf("%-22s %8.1f us for %.0f M element updates (%d passes over %.1f MB): %.1f G updates/s\n", names[m], ms * 1e3,
           elems * 1e-6, nkb, bytes * 1e-6, elems / ms * 1e-6);
  }
  float h0; HIPCHK(hipMemcpy(&h0, acc, 4, hipMemcpyDeviceToHost)); printf("acc[0] = %g\n", h0);
  return 0;
}
