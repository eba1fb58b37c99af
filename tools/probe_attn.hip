// Standalone GPU probe: fused attention forward (+ transpose kernel) vs a CPU double
// reference on sampled query rows; timing of the hot self-attention shapes.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../ctrlora_amd/csrc/attention.h"
#include "../ctrlora_amd/csrc/elementwise.h"

using namespace cl;
#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

static uint32_t rng_state = 777;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float h_bf2f(uint16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }
static int g_fail = 0;
static bool g_tr = false;   // --tr: transpose-free bf16 kernel (attention_tr.hip)

struct Buf {
  std::vector<float> h; void* d = nullptr; size_t n = 0; int dtype = 0;
  void init(size_t n_, int dt, float scale, bool zero = false) {
    n = n_; dtype = dt; h.resize(n);
    for (size_t i = 0; i < n; ++i) { float v = zero ? 0.f : frand() * scale; h[i] = dt == CL_BF16 ? h_bf2f(h_f2bf(v)) : v; }
    HIPCHK(hipMalloc(&d, n * (dt == CL_BF16 ? 2 : 4) + 256));
    if (dt == CL_BF16) { std::vector<uint16_t> t(n); for (size_t i = 0; i < n; ++i) t[i] = h_f2bf(h[i]); HIPCHK(hipMemcpy(d, t.data(), n * 2, hipMemcpyHostToDevice)); }
    else HIPCHK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  }
  void download() {
    if (dtype == CL_BF16) { std::vector<uint16_t> t(n); HIPCHK(hipMemcpy(t.data(), d, n * 2, hipMemcpyDeviceToHost)); for (size_t i = 0; i < n; ++i) h[i] = h_bf2f(t[i]); }
    else HIPCHK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  }
  void free_() { if (d) HIPCHK(hipFree(d)); d = nullptr; }
};

static void run_case(const char* name, int dtype, int B, int H, int N, int Nkv, int DH, int nsample, bool timeit) {
  const int inner = H * DH, pad = (Nkv + 63) / 64 * 64;
  Buf Q, K, V, Vt, O; std::vector<float> lse((size_t)B * H * N);
  // Q/K/V live as column slices of wider buffers (ld > inner) to exercise the strides
  const int ldq = inner + 16, ldk = inner + 8, ldv = inner;
  Q.init((size_t)B * N * ldq, dtype, 1.5f); K.init((size_t)B * Nkv * ldk, dtype, 1.5f); V.init((size_t)B * Nkv * ldv, dtype, 1.0f);
  Vt.init((size_t)B * inner * pad, dtype, 1.0f, true); O.init((size_t)B * N * inner, dtype, 1.f, true);
  float* dlse; HIPCHK(hipMalloc(&dlse, lse.size() * 4));
  int rc = transpose(dtype == CL_BF16 ? CL_BF16 : CL_F32, dtype, V.d, ldv, (long)Nkv * ldv, Vt.d, pad, (long)inner * pad, B, Nkv, inner, pad, 0);
  if (dtype == CL_F32 && rc == 0) {}  // f32->f32 path
  AttnFwdArgs a{}; a.Q = Q.d; a.ldq = ldq; a.K = K.d; a.ldk = ldk; a.Vt = Vt.d; a.nkv_pad = pad; a.O = O.d; a.ldo = inner;
  a.LSE = dlse; a.lse_stride = N; a.B = B; a.H = H; a.N = N; a.Nkv = Nkv; a.DH = DH; a.scale = 1.0f / std::sqrt((float)DH);
  const bool tr = g_tr && dtype == CL_BF16;
  if (!rc) rc = tr ? attn_fwd_tr(a, V.d, ldv, 0) : attn_fwd(a, dtype, 0);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("[FAIL] %s rc=%d\n", name, rc); g_fail++; return; }
  O.download(); HIPCHK(hipMemcpy(lse.data(), dlse, lse.size() * 4, hipMemcpyDeviceToHost));
  double num = 0, den = 0, lse_err = 0;
  std::vector<double> s(Nkv);
  for (int it = 0; it < nsample; ++it) {
    const int b = it % B, h = (it / B) % H;
    int q = (int)((frand() * 0.5f + 0.5f) * N); if (q >= N) q = N - 1; if (it < 4) q = (it & 1) ? N - 1 : 0;
    double mx = -1e300;
    for (int j = 0; j < Nkv; ++j) {
      double d = 0;
      for (int e = 0; e < DH; ++e) d += (double)Q.h[((size_t)b * N + q) * ldq + h * DH + e] * K.h[((size_t)b * Nkv + j) * ldk + h * DH + e];
      s[j] = d * a.scale; mx = std::max(mx, s[j]);
    }
    double l = 0; for (int j = 0; j < Nkv; ++j) { s[j] = std::exp(s[j] - mx); l += s[j]; }
    for (int e = 0; e < DH; ++e) {
      double o = 0;
      for (int j = 0; j < Nkv; ++j) o += s[j] * V.h[((size_t)b * Nkv + j) * ldv + h * DH + e];
      o /= l;
      const double d = O.h[((size_t)b * N + q) * inner + h * DH + e] - o; num += d * d; den += o * o;
    }
    const double ref_lse2 = (mx + std::log(l)) * 1.4426950408889634;
    lse_err = std::max(lse_err, std::fabs(ref_lse2 - lse[((size_t)b * H + h) * N + q]));
  }
  const double rel = std::sqrt(num / (den + 1e-30));
  const double tol = dtype == CL_BF16 ? 8e-3 : 2e-5, ltol = dtype == CL_BF16 ? 2e-2 : 1e-4;
  const bool ok = rel <= tol && lse_err <= ltol && std::isfinite(rel);
  printf("[%s] %-46s rel_l2=%.3e lse_err=%.2e\n", ok ? "PASS" : "FAIL", name, rel, lse_err);
  if (!ok) g_fail++;
  if (timeit) {
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) { if (tr) attn_fwd_tr(a, V.d, ldv, 0); else attn_fwd(a, dtype, 0); }
    HIPCHK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) { if (tr) attn_fwd_tr(a, V.d, ldv, 0); else attn_fwd(a, dtype, 0); }
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 20;
    printf("[TIME] %-46s %8.3f ms  %8.1f TFLOP/s\n", name, ms, 4.0 * B * H * (double)N * Nkv * DH / ms * 1e-9);
  }
  Q.free_(); K.free_(); V.free_(); Vt.free_(); O.free_(); HIPCHK(hipFree(dlse));
}

int main(int argc, char** argv) {
  const bool timeit = argc > 1 && !strcmp(argv[1], "--time");
  g_tr = argc > 2 && !strcmp(argv[2], "--tr");
  printf("kernels: %s\n", g_tr ? "transpose-free (tr)" : "round-0");
  run_case("bf16 d40 N200 self", CL_BF16, 2, 3, 200, 200, 40, 96, false);
  run_case("bf16 d40 N130 cross 77", CL_BF16, 2, 8, 130, 77, 40, 96, false);
  run_case("bf16 d80 N256 self", CL_BF16, 2, 8, 256, 256, 80, 96, false);
  run_case("bf16 d160 N64 self", CL_BF16, 2, 8, 64, 64, 160, 96, false);
  run_case("bf16 d160 N300 cross 77", CL_BF16, 1, 8, 300, 77, 160, 96, false);
  run_case("bf16 d8 N16 self (tiny)", CL_BF16, 2, 8, 16, 16, 8, 64, false);
  run_case("bf16 d32 N4 self (tiny)", CL_BF16, 2, 8, 4, 4, 32, 32, false);
  run_case("f32 d40 N200 self", CL_F32, 2, 3, 200, 200, 40, 96, false);
  run_case("f32 d160 N100 cross 77", CL_F32, 1, 8, 100, 77, 160, 96, false);
  run_case("f32 d80 N256 self", CL_F32, 1, 8, 256, 256, 80, 96, false);
  run_case("f32 d16 N64 self (tiny)", CL_F32, 2, 8, 64, 64, 16, 64, false);
  run_case("bf16 d40 N4096 self B8 (QW=2)", CL_BF16, 8, 8, 4096, 4096, 40, 128, timeit);
  run_case("bf16 d80 N1024 self B8", CL_BF16, 8, 8, 1024, 1024, 80, 128, timeit);
  run_case("bf16 d160 N256 self B8", CL_BF16, 8, 8, 256, 256, 160, 128, timeit);
  run_case("bf16 d40 N4096 cross 77 B8", CL_BF16, 8, 8, 4096, 77, 40, 128, timeit);
  printf("probe_attn: %s (%d failures)\n", g_fail ? "FAILED" : "ALL PASS", g_fail);
  return g_fail ? 1 : 0;
}
