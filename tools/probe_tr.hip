// Probe of ds_read_b64_tr_b16 semantics on gfx950: LDS holds short[i] = i; every lane reads
// from a chosen byte address and dumps its 4 result shorts.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

__global__ void k(const int* addr, uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)lds;
  uint32_t a = base + addr[threadIdx.x];
  typedef __attribute__((ext_vector_type(2))) uint32_t u2;
  u2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = v.x & 0xffff; out[threadIdx.x * 4 + 1] = v.x >> 16;
  out[threadIdx.x * 4 + 2] = v.y & 0xffff; out[threadIdx.x * 4 + 3] = v.y >> 16;
}

int main() {
  int* da; uint16_t* dout; HIPCHK(hipMalloc(&da, 64 * 4)); HIPCHK(hipMalloc(&dout, 64 * 4 * 2));
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) {
      if (mode == 0) a[l] = l * 8;                                   // linear: lane l -> shorts 4l..4l+3
      else if (mode == 1) a[l] = ((l & 15) * 64 + (l >> 4) * 8);     // row-major [16 rows][32 shorts]: row = l&15, col4 = l>>4
      else a[l] = ((l & 15) * 2 + (l >> 4) * 4 * 64 * 2) ;           // guide's V-subtile style: col (l&15), row block (l>>4)*4 of a [rows][64-short] matrix? (8B aligned only if even)
    }
    if (mode == 2) for (int l = 0; l < 64; ++l) a[l] = (((l >> 4) * 4) * 128 + ((l & 15) >> 2) * 8) ;  // rows 4*(l>>4), 8B piece index (l&15)>>2 : see what comes back
    HIPCHK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, dout, mode);
    HIPCHK(hipDeviceSynchronize());
    std::vector<uint16_t> o(256); HIPCHK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d addr %5d (short %4d): %4d %4d %4d %4d\n", l, a[l], a[l] / 2, o[l * 4], o[l * 4 + 1], o[l * 4 + 2], o[l * 4 + 3]);
  }
  return 0;
}
