// Standalone GPU probe: fused attention backward vs a full CPU double reference.
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

static uint32_t rng_state = 4242;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float h_bf2f(uint16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }
static int g_fail = 0;
static bool g_tr = false;   // --tr: transpose-free bf16 kernels (attention_tr.hip)

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

static double relerr(const std::vector<float>& got, const std::vector<double>& ref) {
  double num = 0, den = 0;
  for (size_t i = 0; i < ref.size(); ++i) { double d = got[i] - ref[i]; num += d * d; den += ref[i] * ref[i]; }
  return std::sqrt(num / (den + 1e-30));
}

static void run_case(const char* name, int dtype, int B, int H, int N, int Nkv, int DH, bool want_dkv, bool timeit) {
  const int inner = H * DH, npad = (N + 63) / 64 * 64, kpad = (Nkv + 63) / 64 * 64;
  Buf Q, K, V, dO, O, Vt, Qt, dOt, Kt, dQ, dK, dV;
  Q.init((size_t)B * N * inner, dtype, 1.2f); K.init((size_t)B * Nkv * inner, dtype, 1.2f); V.init((size_t)B * Nkv * inner, dtype, 1.f);
  dO.init((size_t)B * N * inner, dtype, 1.f); O.init((size_t)B * N * inner, dtype, 1.f, true);
  Vt.init((size_t)B * inner * kpad, dtype, 1.f, true); Kt.init((size_t)B * inner * kpad, dtype, 1.f, true);
  Qt.init((size_t)B * inner * npad, dtype, 1.f, true); dOt.init((size_t)B * inner * npad, dtype, 1.f, true);
  dQ.init((size_t)B * N * inner, dtype, 1.f, true); dK.init((size_t)B * Nkv * inner, dtype, 1.f, true); dV.init((size_t)B * Nkv * inner, dtype, 1.f, true);
  float *lse, *delta; HIPCHK(hipMalloc(&lse, (size_t)B * H * npad * 4)); HIPCHK(hipMalloc(&delta, (size_t)B * H * npad * 4));
  HIPCHK(hipMemset(lse, 0xff, (size_t)B * H * npad * 4));  // NaN-fill the pad to catch unmasked reads
  int rc = 0;
  rc |= transpose(dtype, dtype, V.d, inner, (long)Nkv * inner, Vt.d, kpad, (long)inner * kpad, B, Nkv, inner, kpad, 0);
  rc |= transpose(dtype, dtype, K.d, inner, (long)Nkv * inner, Kt.d, kpad, (long)inner * kpad, B, Nkv, inner, kpad, 0);
  rc |= transpose(dtype, dtype, Q.d, inner, (long)N * inner, Qt.d, npad, (long)inner * npad, B, N, inner, npad, 0);
  rc |= transpose(dtype, dtype, dO.d, inner, (long)N * inner, dOt.d, npad, (long)inner * npad, B, N, inner, npad, 0);
  const float scale = 1.0f / std::sqrt((float)DH);
  AttnFwdArgs f{}; f.Q = Q.d; f.ldq = inner; f.K = K.d; f.ldk = inner; f.Vt = Vt.d; f.nkv_pad = kpad; f.O = O.d; f.ldo = inner;
  f.LSE = lse; f.lse_stride = npad; f.B = B; f.H = H; f.N = N; f.Nkv = Nkv; f.DH = DH; f.scale = scale;
  const bool tr = g_tr && dtype == CL_BF16;
  rc |= tr ? attn_fwd_tr(f, V.d, inner, 0) : attn_fwd(f, dtype, 0);
  AttnBwdArgs a{}; a.Q = Q.d; a.ldq = inner; a.K = K.d; a.ldk = inner; a.V = V.d; a.ldv = inner; a.O = O.d; a.ldo = inner;
  a.dO = dO.d; a.lddo = inner; a.Qt = Qt.d; a.dOt = dOt.d; a.n_pad = npad; a.Kt = Kt.d; a.nkv_pad = kpad; a.LSE = lse; a.Delta = delta;
  a.lse_stride = npad; a.dQ = dQ.d; a.lddq = inner; a.dK = want_dkv ? dK.d : nullptr; a.lddk = inner; a.dV = want_dkv ? dV.d : nullptr; a.lddv = inner;
  a.B = B; a.H = H; a.N = N; a.Nkv = Nkv; a.DH = DH; a.scale = scale;
  rc |= tr ? attn_bwd_tr(a, 0) : attn_bwd(a, dtype, 0);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("[FAIL] %s rc=%d\n", name, rc); g_fail++; return; }
  if ((long)B * H * N * Nkv <= 4000000L) {
    dQ.download(); dK.download(); dV.download();
    std::vector<double> rQ(dQ.n, 0.0), rK(dK.n, 0.0), rV(dV.n, 0.0), P((size_t)N * Nkv), dP((size_t)N * Nkv);
    for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) {
      auto q = [&](int i, int e) { return (double)Q.h[((size_t)b * N + i) * inner + h * DH + e]; };
      auto k = [&](int j, int e) { return (double)K.h[((size_t)b * Nkv + j) * inner + h * DH + e]; };
      auto v = [&](int j, int e) { return (double)V.h[((size_t)b * Nkv + j) * inner + h * DH + e]; };
      auto go = [&](int i, int e) { return (double)dO.h[((size_t)b * N + i) * inner + h * DH + e]; };
      for (int i = 0; i < N; ++i) {
        double mx = -1e300;
        for (int j = 0; j < Nkv; ++j) { double s = 0; for (int e = 0; e < DH; ++e) s += q(i, e) * k(j, e); P[(size_t)i * Nkv + j] = s * scale; mx = std::max(mx, s * scale); }
        double l = 0; for (int j = 0; j < Nkv; ++j) { P[(size_t)i * Nkv + j] = std::exp(P[(size_t)i * Nkv + j] - mx); l += P[(size_t)i * Nkv + j]; }
        double dl = 0;
        for (int j = 0; j < Nkv; ++j) {
          P[(size_t)i * Nkv + j] /= l;
          double d = 0; for (int e = 0; e < DH; ++e) d += go(i, e) * v(j, e);
          dP[(size_t)i * Nkv + j] = d; dl += d * P[(size_t)i * Nkv + j];
        }
        for (int j = 0; j < Nkv; ++j) {
          const double p = P[(size_t)i * Nkv + j], ds = p * (dP[(size_t)i * Nkv + j] - dl) * scale;
          for (int e = 0; e < DH; ++e) {
            rV[((size_t)b * Nkv + j) * inner + h * DH + e] += p * go(i, e);
            rQ[((size_t)b * N + i) * inner + h * DH + e] += ds * k(j, e);
            rK[((size_t)b * Nkv + j) * inner + h * DH + e] += ds * q(i, e);
          }
        }
      }
    }
    const double eq = relerr(dQ.h, rQ), ek = want_dkv ? relerr(dK.h, rK) : 0, ev = want_dkv ? relerr(dV.h, rV) : 0;
    const double tol = dtype == CL_BF16 ? 1.2e-2 : 3e-5;
    const bool ok = eq <= tol && ek <= tol && ev <= tol && std::isfinite(eq + ek + ev);
    printf("[%s] %-40s dQ=%.3e dK=%.3e dV=%.3e\n", ok ? "PASS" : "FAIL", name, eq, ek, ev);
    if (!ok) g_fail++;
  }
  if (timeit) {
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) { if (tr) attn_bwd_tr(a, 0); else attn_bwd(a, dtype, 0); }
    HIPCHK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) { if (tr) attn_bwd_tr(a, 0); else attn_bwd(a, dtype, 0); }
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("[TIME] %-40s %8.3f ms  %8.1f TFLOP/s (10 N Nkv d flops)\n", name, ms, (want_dkv ? 10.0 : 6.0) * B * H * (double)N * Nkv * DH / ms * 1e-9);
  }
  Buf* all[] = {&Q, &K, &V, &dO, &O, &Vt, &Qt, &dOt, &Kt, &dQ, &dK, &dV};
  for (auto* x : all) x->free_();
  HIPCHK(hipFree(lse)); HIPCHK(hipFree(delta));
}

int main(int argc, char** argv) {
  const bool timeit = argc > 1 && !strcmp(argv[1], "--time");
  g_tr = argc > 2 && !strcmp(argv[2], "--tr");
  printf("kernels: %s\n", g_tr ? "transpose-free (tr)" : "round-0");
  run_case("bf16 d40 N200 self", CL_BF16, 2, 3, 200, 200, 40, true, false);
  run_case("bf16 d40 N130 cross 77", CL_BF16, 2, 4, 130, 77, 40, true, false);
  run_case("bf16 d40 N130 cross 77 (dQ only)", CL_BF16, 2, 4, 130, 77, 40, false, false);
  run_case("bf16 d80 N128 self", CL_BF16, 1, 8, 128, 128, 80, true, false);
  run_case("bf16 d160 N64 self", CL_BF16, 2, 8, 64, 64, 160, true, false);
  run_case("bf16 d160 N100 cross 77", CL_BF16, 1, 8, 100, 77, 160, true, false);
  run_case("bf16 d8 N16 self (tiny)", CL_BF16, 2, 8, 16, 16, 8, true, false);
  run_case("bf16 d32 N4 self (tiny)", CL_BF16, 2, 8, 4, 4, 32, true, false);
  run_case("bf16 d16 N256 cross 77 (tiny)", CL_BF16, 2, 8, 256, 77, 16, true, false);
  run_case("f32 d40 N200 self", CL_F32, 1, 3, 200, 200, 40, true, false);
  run_case("f32 d160 N100 cross 77", CL_F32, 1, 4, 100, 77, 160, true, false);
  run_case("f32 d80 N96 self", CL_F32, 1, 4, 96, 96, 80, true, false);
  run_case("f32 d16 N64 self (tiny)", CL_F32, 2, 8, 64, 64, 16, true, false);
  run_case("f32 d8 N300 cross 77 (tiny)", CL_F32, 1, 8, 300, 77, 8, true, false);
  if (timeit) {
    run_case("bf16 d40 N4096 self B8", CL_BF16, 8, 8, 4096, 4096, 40, true, true);
    run_case("bf16 d80 N1024 self B8", CL_BF16, 8, 8, 1024, 1024, 80, true, true);
    run_case("bf16 d160 N256 self B8", CL_BF16, 8, 8, 256, 256, 160, true, true);
    run_case("bf16 d40 N4096 cross 77 B8", CL_BF16, 8, 8, 4096, 77, 40, true, true);
  }
  printf("probe_attn_bwd: %s (%d failures)\n", g_fail ? "FAILED" : "ALL PASS", g_fail);
  return g_fail ? 1 : 0;
}
