// Semantics probe of v_permlane16_swap_b32 / v_permlane32_swap_b32 on gfx950: lane l passes a = l, b = 100 + l and prints both
// results.  Expected for permlane16_swap (rows = 16 lanes): rows 1 / 3 of the first operand <-> rows 0 / 2 of the second.
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  const auto q = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; HIPCHK(hipMalloc(&d, 256 * 4));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  HIPCHK(hipDeviceSynchronize());
  unsigned h[256]; HIPCHK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
  const char* names[4] = {"permlane16_swap r[0]", "permlane16_swap r[1]", "permlane32_swap r[0]", "permlane32_swap r[1]"};
  for (int v = 0; v < 4; ++v) {
    printf("%s:", names[v]);
    for (int l = 0; l < 64; ++l) printf("%s%u", (l % 16 == 0) ? " | " : " ", h[v * 64 + l]);
    printf("\n");
  }
  return 0;
}
