// Standalone GPU probe for the GEMM / implicit-conv core: checks every mode
// against a CPU double reference on asymmetric random data and times a few
// hot-path shapes.  Build: see tools/build_probes.sh.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <map>
#include <vector>
#include "../ctrlora_amd/csrc/gemm.h"

using namespace cl;

#define HIPCHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(2); } } while (0)

static uint32_t rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xffff) / 32768.0f - 1.0f; }
static uint16_t h_f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float h_bf2f(uint16_t v) { uint32_t u = ((uint32_t)v) << 16; float f; memcpy(&f, &u, 4); return f; }

struct Buf {
  std::vector<float> h;  // values as float (already rounded to storage precision)
  void* d = nullptr; size_t n = 0; int dtype = 0;
  void init(size_t n_, int dtype_, float scale = 1.0f, bool zero = false) {
    n = n_; dtype = dtype_; h.resize(n);
    for (size_t i = 0; i < n; ++i) { float v = zero ? 0.f : frand() * scale; h[i] = dtype == CL_BF16 ? h_bf2f(h_f2bf(v)) : v; }
    upload();
  }
  void upload() {
    if (!d) HIPCHK(hipMalloc(&d, n * (dtype == CL_BF16 ? 2 : 4) + 256));
    if (dtype == CL_BF16) { std::vector<uint16_t> t(n); for (size_t i = 0; i < n; ++i) t[i] = h_f2bf(h[i]); HIPCHK(hipMemcpy(d, t.data(), n * 2, hipMemcpyHostToDevice)); }
    else HIPCHK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  }
  void download(int dt) {
    if (dt == CL_BF16) { std::vector<uint16_t> t(n); HIPCHK(hipMemcpy(t.data(), d, n * 2, hipMemcpyDeviceToHost)); for (size_t i = 0; i < n; ++i) h[i] = h_bf2f(t[i]); }
    else HIPCHK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  }
};

static void* g_zero;
static int g_fail = 0;

static void report(const char* name, double num, double den, double tol) {
  double rel = std::sqrt(num / (den + 1e-30));
  bool ok = rel <= tol && std::isfinite(rel);
  printf("[%s] %-44s rel_l2=%.3e (tol %.1e)\n", ok ? "PASS" : "FAIL", name, rel, tol);
  if (!ok) g_fail++;
}

// ---------------- linear cases ----------------
static void case_linear(const char* name, int dtype, int M, int N, int K1, int K2, bool bias, bool resid, bool rowb,
                        int act, float alpha, float beta, bool out_f32, int splitk) {
  Buf A1, W1, A2, W2, R, RB, C; std::vector<float> hb(N);
  A1.init((size_t)M * K1, dtype); W1.init((size_t)N * K1, dtype, 0.1f);
  if (K2) { A2.init((size_t)M * K2, dtype); W2.init((size_t)N * K2, dtype, 0.1f); }
  const int rpb = 7; const int nb = (M + rpb - 1) / rpb;
  if (resid) R.init((size_t)M * N, dtype);
  if (rowb) RB.init((size_t)nb * N, dtype);
  float* dbias = nullptr;
  if (bias) { for (auto& v : hb) v = frand(); HIPCHK(hipMalloc(&dbias, N * 4)); HIPCHK(hipMemcpy(dbias, hb.data(), N * 4, hipMemcpyHostToDevice)); }
  const bool atomic = splitk > 1;
  const int odt = (out_f32 || atomic) ? CL_F32 : dtype;
  C.init((size_t)M * N, odt, 1.0f, true);
  GemmParams p{}; p.A1 = A1.d; p.lda1 = K1; p.K1 = K1; p.W1 = W1.d; p.ldw1 = K1;
  if (K2) { p.A2 = A2.d; p.lda2 = K2; p.K2 = K2; p.W2 = W2.d; p.ldw2 = K2; }
  p.M = M; p.N = N; p.mode = GEMM_LINEAR; p.zero_page = nullptr;  /* (as the product: linear calls carry no zero page) */ p.bias = dbias;
  if (rowb) { p.rowbias = RB.d; p.ldrb = N; p.rows_per_batch = rpb; }
  if (resid) { p.residual = R.d; p.ldr = N; }
  p.alpha = alpha; p.beta = beta; p.act = act; p.C = C.d; p.ldc = N; p.out_f32 = out_f32; p.atomic = atomic; p.splitk = splitk;
  int rc = launch_gemm(p, dtype, 0);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("[FAIL] %s rc=%d\n", name, rc); g_fail++; return; }
  C.download(odt);
  double num = 0, den = 0;
  for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
    double s = 0;
    for (int k = 0; k < K1; ++k) s += (double)A1.h[(size_t)m * K1 + k] * W1.h[(size_t)n * K1 + k];
    for (int k = 0; k < K2; ++k) s += (double)A2.h[(size_t)m * K2 + k] * W2.h[(size_t)n * K2 + k];
    if (bias) s += hb[n];
    if (rowb) s += RB.h[(size_t)(m / rpb) * N + n];
    if (act == ACT_SILU) s = s / (1.0 + std::exp(-s));
    s *= alpha;
    if (resid) s += beta * R.h[(size_t)m * N + n];
    double d = C.h[(size_t)m * N + n] - s; num += d * d; den += s * s;
  }
  report(name, num, den, odt == CL_BF16 ? 4e-3 : 2e-5);
}

// ---------------- x-stationary streaming kernel (configuration 34, gemm_xs.hip) ----------------
// groups > 1: output columns [g N / groups, (g + 1) N / groups) take their second K segment from columns [g K2, (g + 1) K2) of A2
static float g_xs_beta = 0.f;      // != 0: the next case_xs adds beta * residual
static bool g_xs_geglu = false;    // the next case_xs is a fused GEGLU projection (N = value | gate rows, output N / 2 columns)
static bool g_xs_ln = false;       // the next case_xs normalises its rows first (LayerNorm prologue; K2 must be 0)
static void case_xs(const char* name, int M, int N, int K1, int K2, bool bias, int groups, float alpha, int alpha_n, int nsplit,
                    bool timeit = false) {
  const int dtype = CL_BF16;
  const float beta = g_xs_beta; const bool geglu = g_xs_geglu, ln = g_xs_ln;
  const int NO = geglu ? N / 2 : N;     // output columns
  Buf A1, W1, A2, W2, C, C2, R; std::vector<float> hb(N);
  if (beta != 0.f) R.init((size_t)M * NO, dtype);
  A1.init((size_t)M * K1, dtype); W1.init((size_t)N * K1, dtype, 0.1f);
  if (K2) { A2.init((size_t)M * K2 * groups, dtype); W2.init((size_t)N * K2, dtype, 0.1f); }
  float* dbias = nullptr;
  if (bias) { for (auto& v : hb) v = frand(); HIPCHK(hipMalloc(&dbias, N * 4)); HIPCHK(hipMemcpy(dbias, hb.data(), N * 4, hipMemcpyHostToDevice)); }
  C.init((size_t)M * NO, dtype, 1.0f, true); C2.init((size_t)M * NO, dtype, 1.0f, true);
  GemmParams p{}; p.A1 = A1.d; p.lda1 = K1; p.K1 = K1; p.W1 = W1.d; p.ldw1 = K1;
  if (K2) { p.A2 = A2.d; p.lda2 = K2 * groups; p.K2 = K2; p.W2 = W2.d; p.ldw2 = K2; if (groups > 1) p.a2_group_n = N / groups; }
  p.M = M; p.N = N; p.mode = GEMM_LINEAR; p.zero_page = nullptr; p.bias = dbias;
  p.alpha = alpha; p.alpha_n = alpha_n; p.C = C.d; p.ldc = NO; p.splitk = 1;
  if (beta != 0.f) { p.residual = R.d; p.ldr = NO; p.beta = beta; }
  if (geglu) p.act = ACT_GEGLU_SPLIT;
  std::vector<float> lg(K1), lb(K1), xn;      // LayerNorm prologue: the CPU reference normalises in double and rounds to bf16
  float *dlg = nullptr, *dlb = nullptr, *dstats = nullptr;
  if (ln) {
    for (size_t i = 0; i < A1.h.size(); ++i) A1.h[i] = h_bf2f(h_f2bf(A1.h[i] * 1.7f + 0.6f));
    A1.upload();
    for (int k = 0; k < K1; ++k) { lg[k] = 1.0f + 0.3f * frand(); lb[k] = 0.3f * frand(); }
    HIPCHK(hipMalloc(&dlg, K1 * 4)); HIPCHK(hipMalloc(&dlb, K1 * 4)); HIPCHK(hipMalloc(&dstats, (size_t)M * 8));
    HIPCHK(hipMemcpy(dlg, lg.data(), K1 * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dlb, lb.data(), K1 * 4, hipMemcpyHostToDevice));
    p.ln_gamma = dlg; p.ln_beta = dlb; p.ln_eps = 1e-5f; p.ln_stats = dstats;
    xn.resize(A1.h.size());
    for (int m = 0; m < M; ++m) {
      double mu = 0, var = 0;
      for (int k = 0; k < K1; ++k) mu += A1.h[(size_t)m * K1 + k];
      mu /= K1;
      for (int k = 0; k < K1; ++k) { const double d = A1.h[(size_t)m * K1 + k] - mu; var += d * d; }
      const double rstd = 1.0 / std::sqrt(var / K1 + 1e-5);
      for (int k = 0; k < K1; ++k) xn[(size_t)m * K1 + k] = h_bf2f(h_f2bf((float)((A1.h[(size_t)m * K1 + k] - mu) * rstd * lg[k] + lb[k])));
    }
  }
  const std::vector<float>& XA = ln ? xn : A1.h;
  HIPCHK(hipMemset(C.d, 0xff, (size_t)M * NO * 2));
  const int rc = launch_gemm_xs(p, 0, nsplit);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("[FAIL] %s rc=%d\n", name, rc); g_fail++; return; }
  C.download(dtype);
  double num = 0, den = 0;
  const int rows_checked = M <= 2048 ? M : 96;
  for (int t = 0; t < rows_checked; ++t) {
    const int m = M <= 2048 ? t : (int)(((long)t * 7919 + (t % 3 == 0 ? M - 1 - t : 13)) % M);
    auto dotrow = [&](int n) {
      double s = 0;
      const int g = groups > 1 ? n / (N / groups) : 0;
      for (int k = 0; k < K1; ++k) s += (double)XA[(size_t)m * K1 + k] * W1.h[(size_t)n * K1 + k];
      for (int k = 0; k < K2; ++k) s += (double)A2.h[(size_t)m * K2 * groups + g * K2 + k] * W2.h[(size_t)n * K2 + k];
      if (bias) s += hb[n];
      return s;
    };
    for (int n = 0; n < NO; ++n) {
      double s = dotrow(n);
      if (geglu) { const double g = dotrow(NO + n); s *= 0.5 * g * (1.0 + std::erf(g * 0.70710678118654752440)); }
      else if (!(alpha_n > 0 && n >= alpha_n)) s *= alpha;
      if (beta != 0.f) s += (double)beta * R.h[(size_t)m * NO + n];
      const double d = C.h[(size_t)m * NO + n] - s; num += d * d; den += s * s;
    }
  }
  report(name, num, den, 4e-3);
  if (ln) {   // the statistics the kernel hands to the backward pass
    std::vector<float> st((size_t)M * 2);
    HIPCHK(hipMemcpy(st.data(), dstats, st.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int m = 0; m < M; ++m) {
      double mu = 0, var = 0;
      for (int k = 0; k < K1; ++k) mu += A1.h[(size_t)m * K1 + k];
      mu /= K1;
      for (int k = 0; k < K1; ++k) { const double d = A1.h[(size_t)m * K1 + k] - mu; var += d * d; }
      const double rstd = 1.0 / std::sqrt(var / K1 + 1e-5);
      worst = std::max(worst, std::max(std::fabs(st[2 * m] - mu) / (std::fabs(mu) + 1e-3), std::fabs(st[2 * m + 1] - rstd) / rstd));
    }
    printf("       LayerNorm statistics (mean, rstd) worst relative error %.2e%s\n", worst, worst > 1e-5 ? "  <-- FAIL" : "");
    if (worst > 1e-5) g_fail++;
  }
  // bitwise repeatability + agreement with the tile kernels on the same parameters
  std::vector<uint16_t> r0((size_t)M * NO), r1((size_t)M * NO);
  HIPCHK(hipMemcpy(r0.data(), C.d, r0.size() * 2, hipMemcpyDeviceToHost));
  int diff = 0;
  for (int rep = 0; rep < 3; ++rep) {
    HIPCHK(hipMemset(C.d, 0xff, (size_t)M * NO * 2));
    launch_gemm_xs(p, 0, nsplit);
    HIPCHK(hipMemcpy(r1.data(), C.d, r1.size() * 2, hipMemcpyDeviceToHost));
    diff += memcmp(r0.data(), r1.data(), r0.size() * 2) != 0;
  }
  p.C = C2.d;
  const int keep = g_gemm_force_cfg; g_gemm_force_cfg = -1;
  if (!geglu && !ln) launch_gemm(p, dtype, 0);       // (the tile kernels' GEGLU wants permuted rows, and they have no LayerNorm prologue: no cross-check there)
  g_gemm_force_cfg = keep;
  HIPCHK(hipDeviceSynchronize());
  C2.download(dtype);
  double n2 = 0, d2 = 0;
  if (!geglu && !ln) for (size_t i = 0; i < (size_t)M * NO; ++i) { const double d = C.h[i] - C2.h[i]; n2 += d * d; d2 += (double)C2.h[i] * C2.h[i]; }
  printf("       repeat launches differing: %d of 3; vs tile kernels rel_l2 %.2e%s\n", diff, std::sqrt(n2 / (d2 + 1e-30)),
         (diff || std::sqrt(n2 / (d2 + 1e-30)) > 4e-3) ? "  <-- FAIL" : "");
  if (diff || std::sqrt(n2 / (d2 + 1e-30)) > 4e-3) g_fail++;
  if (timeit) {
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto timed = [&](int cfg) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        HIPCHK(hipEventRecord(e0, 0));
        for (int i = 0; i < 20; ++i) { if (cfg == 34) launch_gemm_xs(p, 0, nsplit); else launch_gemm(p, dtype, 0); }
        HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 20);
      }
      return best * 1e3;
    };
    const double fl = 2.0 * M * N * (K1 + K2);
    const float t_xs = timed(34), t_tile = (geglu || ln) ? 0.f : timed(-1);
    printf("[TIME] %-44s xs %7.2f us (%6.1f TF/s)   tile kernels (rules) %7.2f us\n", name, t_xs, fl / t_xs * 1e-6, t_tile);
  }
  hipFree(A1.d); hipFree(W1.d); hipFree(C.d); hipFree(C2.d); if (K2) { hipFree(A2.d); hipFree(W2.d); } if (dbias) hipFree(dbias);
  if (R.d) hipFree(R.d);
  if (dlg) { hipFree(dlg); hipFree(dlb); hipFree(dstats); }
  g_xs_beta = 0.f; g_xs_geglu = false; g_xs_ln = false;
}

// ---------------- GEGLU-fused projection ----------------
static void case_geglu(const char* name, int dtype, int M, int half, int K1, int K2, bool out_f32) {
  const int N = 2 * half;
  Buf A1, W1, A2, W2, C; std::vector<float> hb(N);
  A1.init((size_t)M * K1, dtype); W1.init((size_t)N * K1, dtype, 0.1f);
  if (K2) { A2.init((size_t)M * K2, dtype); W2.init((size_t)N * K2, dtype, 0.1f); }
  for (auto& v : hb) v = frand();
  // device copies with the rows permuted: tile j = [80 value rows 80j.. | 80 gate rows half+80j..]
  auto perm = [&](int r) { const int j = r / 160, c = r % 160; return c < 80 ? j * 80 + c : half + j * 80 + (c - 80); };
  Buf Wp, W2p; Wp.init((size_t)N * K1, dtype, 1.f, true); if (K2) W2p.init((size_t)N * K2, dtype, 1.f, true);
  std::vector<float> hbp(N);
  for (int r = 0; r < N; ++r) {
    memcpy(&Wp.h[(size_t)r * K1], &W1.h[(size_t)perm(r) * K1], K1 * 4);
    if (K2) memcpy(&W2p.h[(size_t)r * K2], &W2.h[(size_t)perm(r) * K2], K2 * 4);
    hbp[r] = hb[perm(r)];
  }
  Wp.upload(); if (K2) W2p.upload();
  float* dbias; HIPCHK(hipMalloc(&dbias, N * 4)); HIPCHK(hipMemcpy(dbias, hbp.data(), N * 4, hipMemcpyHostToDevice));
  const int odt = out_f32 ? CL_F32 : dtype;
  C.init((size_t)M * half, odt, 1.0f, true);
  GemmParams p{}; p.A1 = A1.d; p.lda1 = K1; p.K1 = K1; p.W1 = Wp.d; p.ldw1 = K1;
  if (K2) { p.A2 = A2.d; p.lda2 = K2; p.K2 = K2; p.W2 = W2p.d; p.ldw2 = K2; }
  p.M = M; p.N = N; p.mode = GEMM_LINEAR; p.zero_page = nullptr; p.bias = dbias; p.alpha = 1.f; p.act = ACT_GEGLU;
  p.C = C.d; p.ldc = half; p.out_f32 = out_f32; p.splitk = 1;
  int rc = launch_gemm(p, dtype, 0);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("[FAIL] %s rc=%d\n", name, rc); g_fail++; return; }
  C.download(odt);
  double num = 0, den = 0;
  for (int m = 0; m < M; ++m) for (int o = 0; o < half; ++o) {
    double v = hb[o], g = hb[half + o];
    for (int k = 0; k < K1; ++k) { v += (double)A1.h[(size_t)m * K1 + k] * W1.h[(size_t)o * K1 + k]; g += (double)A1.h[(size_t)m * K1 + k] * W1.h[(size_t)(half + o) * K1 + k]; }
    for (int k = 0; k < K2; ++k) { v += (double)A2.h[(size_t)m * K2 + k] * W2.h[(size_t)o * K2 + k]; g += (double)A2.h[(size_t)m * K2 + k] * W2.h[(size_t)(half + o) * K2 + k]; }
    const double r = v * 0.5 * g * (1.0 + std::erf(g * 0.70710678118654752440));
    const double d = C.h[(size_t)m * half + o] - r; num += d * d; den += r * r;
  }
  report(name, num, den, odt == CL_BF16 ? 4e-3 : 2e-5);
}

// ---------------- conv cases ----------------
static void case_conv(const char* name, int dtype, int mode, int B, int Hin, int Win, int C, int N) {
  int Hout, Wout;
  if (mode == GEMM_CONV_S1) { Hout = Hin; Wout = Win; }
  else if (mode == GEMM_CONV_S2) { Hout = Hin / 2; Wout = Win / 2; }
  else { Hout = 2 * Hin; Wout = 2 * Win; }
  const int M = B * Hout * Wout;
  Buf X, W, Cb; std::vector<float> hb(N);
  X.init((size_t)B * Hin * Win * C, dtype); W.init((size_t)N * 9 * C, dtype, 0.1f);
  for (auto& v : hb) v = frand();
  float* dbias; HIPCHK(hipMalloc(&dbias, N * 4)); HIPCHK(hipMemcpy(dbias, hb.data(), N * 4, hipMemcpyHostToDevice));
  Cb.init((size_t)M * N, CL_F32, 1.0f, true);
  GemmParams p{}; p.A1 = X.d; p.lda1 = C; p.K1 = C; p.W1 = W.d; p.ldw1 = 9 * C; p.M = M; p.N = N; p.mode = mode;
  p.B = B; p.Hin = Hin; p.Win = Win; p.Hout = Hout; p.Wout = Wout; p.zero_page = g_zero; p.bias = dbias;
  p.alpha = 1.f; p.beta = 0.f; p.C = Cb.d; p.ldc = N; p.out_f32 = 1; p.splitk = 1;
  int rc = launch_gemm(p, dtype, 0);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("[FAIL] %s rc=%d\n", name, rc); g_fail++; return; }
  Cb.download(CL_F32);
  double num = 0, den = 0;
  for (int b = 0; b < B; ++b) for (int oy = 0; oy < Hout; ++oy) for (int ox = 0; ox < Wout; ++ox) for (int n = 0; n < N; ++n) {
    double s = hb[n];
    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
      int iy, ix; bool ok;
      if (mode == GEMM_CONV_S1) { iy = oy + ky - 1; ix = ox + kx - 1; ok = iy >= 0 && iy < Hin && ix >= 0 && ix < Win; }
      else if (mode == GEMM_CONV_S2) { iy = 2 * oy + ky - 1; ix = 2 * ox + kx - 1; ok = iy >= 0 && iy < Hin && ix >= 0 && ix < Win; }
      else {
        int vy = oy + ky - 1, vx = ox + kx - 1; ok = vy >= 0 && vy < 2 * Hin && vx >= 0 && vx < 2 * Win;
        if (mode == GEMM_CONV_T2) ok = ok && !(vy & 1) && !(vx & 1);
        iy = vy >> 1; ix = vx >> 1;
      }
      if (!ok) continue;
      const float* xp = &X.h[(((size_t)b * Hin + iy) * Win + ix) * C];
      const float* wp = &W.h[((size_t)n * 9 + ky * 3 + kx) * C];
      for (int c = 0; c < C; ++c) s += (double)xp[c] * wp[c];
    }
    double d = Cb.h[(((size_t)b * Hout + oy) * Wout + ox) * N + n] - s; num += d * d; den += s * s;
  }
  report(name, num, den, 2e-5);
}

// ---------------- timing ----------------
static int g_probe_act = 0;
static void time_case(const char* name, int dtype, int mode, int M, int N, int K1, int B, int H, int W, int K2 = 0) {
  Buf A, Wt, C, A2, W2;
  const int taps = mode == GEMM_LINEAR ? 1 : 9;
  A.init(mode == GEMM_LINEAR ? (size_t)M * K1 : (size_t)B * H * W * K1, dtype);
  Wt.init((size_t)N * taps * K1, dtype, 0.05f);
  if (K2) { A2.init((size_t)M * K2, dtype); W2.init((size_t)N * K2, dtype, 0.05f); }
  C.init((size_t)M * N, dtype, 1.f, true);
  GemmParams p{}; p.A1 = A.d; p.lda1 = K1; p.K1 = K1; p.W1 = Wt.d; p.ldw1 = taps * K1; p.M = M; p.N = N; p.mode = mode;
  if (K2) { p.A2 = A2.d; p.lda2 = K2; p.K2 = K2; p.W2 = W2.d; p.ldw2 = K2; }
  p.B = B; p.Hin = H; p.Win = W; p.Hout = H; p.Wout = W; p.zero_page = g_zero; p.alpha = 1.f; p.C = C.d; p.ldc = N; p.splitk = 1; p.act = g_probe_act;
  if (mode == GEMM_CONV_UP2) { p.Hout = 2 * H; p.Wout = 2 * W; }
  if (mode == GEMM_CONV_S2) { p.Hout = H / 2; p.Wout = W / 2; }
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch_gemm(p, dtype, 0);
  HIPCHK(hipDeviceSynchronize());
  const int iters = 20;
  HIPCHK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch_gemm(p, dtype, 0);
  HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
  double fl = 2.0 * M * N * ((double)taps * K1 + K2);
  printf("[TIME] %-44s %8.3f ms  %8.1f TFLOP/s\n", name, ms, fl / ms * 1e-9);
  hipFree(A.d); hipFree(Wt.d); hipFree(C.d); if (K2) { hipFree(A2.d); hipFree(W2.d); }
}

// ---------------- transpose-free weight gradient ----------------
static void case_wgrad(const char* name, int M, int N, int K, float alpha, bool timeit = false) {
  Buf DY, X; DY.init((size_t)M * N, CL_BF16); X.init((size_t)M * K, CL_BF16);
  std::vector<float> init((size_t)N * K);
  for (auto& v : init) v = frand();
  float* dW; HIPCHK(hipMalloc(&dW, (size_t)N * K * 4));
  HIPCHK(hipMemcpy(dW, init.data(), (size_t)N * K * 4, hipMemcpyHostToDevice));
  int rc = launch_wgrad_tn(DY.d, N, X.d, K, dW, K, M, N, K, alpha, g_zero, 0);
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("[FAIL] %s rc=%d\n", name, rc); g_fail++; return; }
  if (!timeit) {
    std::vector<float> out((size_t)N * K);
    HIPCHK(hipMemcpy(out.data(), dW, (size_t)N * K * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) {
      double s = 0;
      for (int m = 0; m < M; ++m) s += (double)DY.h[(size_t)m * N + n] * X.h[(size_t)m * K + k];
      s = init[(size_t)n * K + k] + alpha * s;
      double d = out[(size_t)n * K + k] - s; num += d * d; den += s * s;
    }
    report(name, num, den, 2e-5);
  } else {
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    const int iters = 20;
    HIPCHK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) launch_wgrad_tn(DY.d, N, X.d, K, dW, K, M, N, K, alpha, g_zero, 0);
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= iters;
    printf("[TIME] %-44s %8.3f ms  %8.1f GB/s operand reads\n", name, ms, ((double)M * (N + K) * 2) / ms * 1e-6);
  }
  hipFree(dW); hipFree(DY.d); hipFree(X.d);
}

static void wgrad_suite(bool timeit) {
  printf("---- weight gradient (LDS transpose reads)\n");
  case_wgrad("wgrad 300x200x136", 300, 200, 136, 1.0f);
  case_wgrad("wgrad 616x320x128 alpha 0.5", 616, 320, 128, 0.5f);
  case_wgrad("wgrad M=8 1280x128", 8, 1280, 128, 1.0f);
  case_wgrad("wgrad 4096x128x320", 4096, 128, 320, 1.0f);
  case_wgrad("wgrad 1000x8x40 (tiny)", 1000, 8, 40, 1.0f);
  if (timeit) {
    const int rings[] = {4}, mins[] = {16, 8}, blks[] = {256, 512};
    for (int r : rings) for (int mn : mins) for (int b : blks) {
      g_wgrad_ring = r; g_wgrad_min_steps = mn; g_wgrad_blocks = b;
      char nm[96]; snprintf(nm, 96, "wgrad 32768x320x128 R%d min%d blk%d", r, mn, b);
      case_wgrad(nm, 32768, 320, 128, 1.0f, true);
      snprintf(nm, 96, "wgrad 8192x640x128 R%d min%d blk%d", r, mn, b);
      case_wgrad(nm, 8192, 640, 128, 1.0f, true);
    }
    g_wgrad_ring = 4; g_wgrad_min_steps = 16; g_wgrad_blocks = 512;
    case_wgrad("wgrad 32768x320x128 (dB @64^2)", 32768, 320, 128, 1.0f, true);
    case_wgrad("wgrad 32768x128x320 (dA @64^2)", 32768, 128, 320, 1.0f, true);
    case_wgrad("wgrad 32768x2560x128 (dB GEGLU)", 32768, 2560, 128, 1.0f, true);
    case_wgrad("wgrad 32768x320x320 (zero conv)", 32768, 320, 320, 1.0f, true);
    case_wgrad("wgrad 512x1280x1280 (zero conv 8^2)", 512, 1280, 1280, 1.0f, true);
  }
}

// round 5: the nine tap problems of one 3x3 conv's weight gradient + the block's LoRA problems as ONE grouped launch (what a
// pre-training backward stage flushes), for every (rows per step, ring depth) form of wgrad_tn_kernel
static void wgrad_tap_group(int B, int H, int C, int O) {
  const int M = B * H * H;
  Buf DY, X; DY.init((size_t)M * O, CL_BF16); X.init((size_t)M * C, CL_BF16);
  float* dW; HIPCHK(hipMalloc(&dW, (size_t)O * 9 * C * 4)); HIPCHK(hipMemset(dW, 0, (size_t)O * 9 * C * 4));
  WgradDesc d[9];
  for (int t = 0; t < 9; ++t) {
    d[t] = WgradDesc{}; d[t].dy = DY.d; d[t].lddy = O; d[t].x = X.d; d[t].ldx = C; d[t].dW = dW + (size_t)t * C; d[t].lddw = 9 * C;
    d[t].M = M; d[t].N = O; d[t].K = C; d[t].alpha = 1.f; d[t].tap = t; d[t].Hin = H; d[t].Win = H; d[t].Hout = H; d[t].Wout = H;
    d[t].stride = 1; d[t].pad = 1;
  }
  std::vector<float> ref;
  const int forms[][2] = {{32, 4}, {32, 3}, {64, 2}, {64, 3}, {64, 4}};
  for (auto& f : forms) {
    g_wgrad_rows = f[0]; g_wgrad_ring = f[1];
    HIPCHK(hipMemset(dW, 0, (size_t)O * 9 * C * 4));
    int rc = launch_wgrad_tn_group(d, 9, g_zero, 0);
    HIPCHK(hipDeviceSynchronize());
    if (rc) { printf("[FAIL] tap group rows %d ring %d rc=%d\n", f[0], f[1], rc); g_fail++; continue; }
    std::vector<float> out((size_t)O * 9 * C);
    HIPCHK(hipMemcpy(out.data(), dW, out.size() * 4, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    if (ref.empty()) ref = out;       // the 32-row / 4-slot form is the product's (tests/test_gpu_parity_r3.py checks it against fp64)
    for (size_t i = 0; i < out.size(); ++i) { const double dd = out[i] - ref[i]; num += dd * dd; den += (double)ref[i] * ref[i]; }
    hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      HIPCHK(hipEventRecord(e0, 0));
      for (int i = 0; i < 10; ++i) launch_wgrad_tn_group(d, 9, g_zero, 0);
      HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
      float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms / 10);
    }
    const double fl = 2.0 * M * O * 9.0 * C;
    printf("[TIME] conv dW %dx%dx%d %d->%d, 9 taps: rows %2d ring %d  %8.1f us  %7.1f TF/s   vs 32/4 form rel_l2 %.2e%s\n", B, H, H, C, O,
           f[0], f[1], best * 1e3, fl / best * 1e-9, std::sqrt(num / (den + 1e-30)), std::sqrt(num / (den + 1e-30)) > 1e-5 ? "  <-- FAIL" : "");
    if (std::sqrt(num / (den + 1e-30)) > 1e-5) g_fail++;
  }
  g_wgrad_rows = 32; g_wgrad_ring = 4;
  hipFree(dW); hipFree(DY.d); hipFree(X.d);
}

static void correctness_suite(const char* tag) {
  printf("---- correctness: %s\n", tag);
  case_linear("linear bf16 300x200x96 plain", CL_BF16, 300, 200, 96, 0, false, false, false, 0, 1.f, 0.f, true, 1);
  case_linear("linear bf16 300x200x96 bf16-out", CL_BF16, 300, 200, 96, 0, false, false, false, 0, 1.f, 0.f, false, 1);
  case_linear("linear bf16 bias+resid alpha/beta", CL_BF16, 257, 136, 160, 0, true, true, false, 0, 0.5f, 2.0f, true, 1);
  case_linear("linear bf16 lora K2=64", CL_BF16, 300, 200, 96, 64, true, false, false, 0, 1.f, 0.f, true, 1);
  case_linear("linear bf16 rowbias+silu", CL_BF16, 70, 72, 64, 0, true, false, true, ACT_SILU, 1.f, 0.f, true, 1);
  case_linear("linear bf16 splitK=3 atomic", CL_BF16, 100, 64, 32 * 9, 0, false, false, false, 0, 1.f, 0.f, true, 3);
  case_linear("linear bf16 big-tile 1500x520x256", CL_BF16, 1500, 520, 256, 0, true, true, false, 0, 1.f, 1.f, true, 1);
  case_linear("linear bf16 N=320 (2x160) K odd substeps", CL_BF16, 1000, 320, 32 * 7, 32, true, true, false, 0, 1.f, 1.f, true, 1);
  case_linear("linear bf16 ws split-K 300x640x2048", CL_BF16, 300, 640, 2048, 64, true, true, true, ACT_SILU, 0.5f, 1.f, false, 1);
  case_linear("linear bf16 ws split-K 200x320x4096 f32out", CL_BF16, 200, 320, 4096, 0, true, false, false, 0, 1.f, 0.f, true, 1);
  case_linear("linear f32 130x72x48", CL_F32, 130, 72, 48, 0, true, true, false, 0, 1.f, 1.f, true, 1);
  case_linear("linear f32 lora K2=16 big", CL_F32, 2100, 1032, 64, 16, true, false, false, 0, 1.f, 0.f, true, 1);
  case_linear("linear bf16 K%64 1000x320x448+64 bias+resid", CL_BF16, 1000, 320, 448, 64, true, true, false, 0, 1.f, 1.f, true, 1);
  case_linear("linear bf16 K%64 300x200x128 rowbias+silu bf16out", CL_BF16, 300, 200, 128, 0, true, false, true, ACT_SILU, 1.f, 0.f, false, 1);
  case_linear("linear bf16 K%64 one stage 520x136x64", CL_BF16, 520, 136, 64, 0, false, false, false, 0, 1.f, 0.f, true, 1);
  case_linear("linear bf16 K%64 two stages 70x72x128", CL_BF16, 70, 72, 128, 0, true, false, false, 0, 1.f, 0.f, true, 1);
  case_linear("linear f32 K%32 600x320x96+32", CL_F32, 600, 320, 96, 32, true, true, false, 0, 1.f, 1.f, true, 1);
  case_geglu("geglu bf16 700x320(out)x128 f32out", CL_BF16, 700, 320, 128, 0, true);
  case_geglu("geglu bf16 300x160x192+64 bf16out", CL_BF16, 300, 160, 192, 64, false);
  case_geglu("geglu bf16 130x240x2048 (split-K)", CL_BF16, 130, 240, 2048, 0, true);
  case_geglu("geglu f32 200x160x96", CL_F32, 200, 160, 96, 0, true);
  case_linear("linear bf16 M=8 (emb)", CL_BF16, 8, 1280, 320, 32, true, false, false, ACT_SILU, 1.f, 0.f, false, 1);
  case_conv("conv3x3 s1 bf16 2x12x12x32->64", CL_BF16, GEMM_CONV_S1, 2, 12, 12, 32, 64);
  case_conv("conv3x3 s1 bf16 big 2x40x40x64->136", CL_BF16, GEMM_CONV_S1, 2, 40, 40, 64, 136);
  case_conv("conv3x3 s1 bf16 2x16x16x96->320", CL_BF16, GEMM_CONV_S1, 2, 16, 16, 96, 320);
  case_conv("conv3x3 s1 bf16 deep-K 2x8x8x512->320 (ws split)", CL_BF16, GEMM_CONV_S1, 2, 8, 8, 512, 320);
  case_conv("conv3x3 s2 bf16 2x12x12x32->64", CL_BF16, GEMM_CONV_S2, 2, 12, 12, 32, 64);
  case_conv("conv3x3 up2 bf16 2x6x6x32->64", CL_BF16, GEMM_CONV_UP2, 2, 6, 6, 32, 64);
  case_conv("conv3x3 t2 bf16 2x6x6x32->64", CL_BF16, GEMM_CONV_T2, 2, 6, 6, 32, 64);
  case_conv("conv3x3 s2 bf16 2x12x12x64->160", CL_BF16, GEMM_CONV_S2, 2, 12, 12, 64, 160);
  case_conv("conv3x3 up2 bf16 2x6x6x64->72", CL_BF16, GEMM_CONV_UP2, 2, 6, 6, 64, 72);
  case_conv("conv3x3 t2 bf16 2x6x6x128->64", CL_BF16, GEMM_CONV_T2, 2, 6, 6, 128, 64);
  case_conv("conv3x3 s1 f32 2x9x7x32->40", CL_F32, GEMM_CONV_S1, 2, 9, 7, 32, 40);
  case_conv("conv3x3 s1 f32 1x9x7x16->24", CL_F32, GEMM_CONV_S1, 1, 9, 7, 16, 24);
}


// ---------------- loader / consumer kernel (configurations 40 / 41, gemm_w4.hip) ----------------
#ifdef W4_PROBE
namespace cl { void w4_abl_set(int v); void w4_halo_set(int v); void w4_timing_set(unsigned long long* buf); }
#endif
// interleaved A/B of tile configurations on ONE set of operands: median / min of `rounds` timed batches each, the outputs compared with
// configuration cfgs[0]'s (different summation order: rel-L2 at the bf16 rounding level), and every configuration's repeat launches
// compared BITWISE (an LDS-ring race shows up as a run-to-run difference long before it shows up in an error norm)
static void ab_case(const char* name, int mode, int M, int N, int K1, int B, int H, int W, int K2, const int* cfgs, int ncfg, int rounds = 5,
                    bool resid = false, bool rowb_silu = false) {
  const int dtype = CL_BF16;
  Buf A, Wt, A2, W2, R, RB;
  const int taps = mode == GEMM_LINEAR ? 1 : 9;
  A.init(mode == GEMM_LINEAR ? (size_t)M * K1 : (size_t)B * H * W * K1, dtype);
  Wt.init((size_t)N * taps * K1, dtype, 0.05f);
  if (K2) { A2.init((size_t)M * K2, dtype); W2.init((size_t)N * K2, dtype, 0.05f); }
  if (resid) R.init((size_t)M * N, dtype);
  const int rpb = mode == GEMM_LINEAR ? 4096 : H * W;
  if (rowb_silu) RB.init((size_t)((M + rpb - 1) / rpb) * N, dtype);
  std::vector<float> hb(N); for (auto& v : hb) v = frand();
  float* dbias; HIPCHK(hipMalloc(&dbias, N * 4)); HIPCHK(hipMemcpy(dbias, hb.data(), N * 4, hipMemcpyHostToDevice));
  GemmParams p{}; p.A1 = A.d; p.lda1 = K1; p.K1 = K1; p.W1 = Wt.d; p.ldw1 = taps * K1; p.M = M; p.N = N; p.mode = mode;
  if (K2) { p.A2 = A2.d; p.lda2 = K2; p.K2 = K2; p.W2 = W2.d; p.ldw2 = K2; }
  p.B = B; p.Hin = H; p.Win = W; p.Hout = H; p.Wout = W; p.zero_page = mode == GEMM_LINEAR ? nullptr : g_zero; p.alpha = 1.f; p.ldc = N; p.splitk = 1; p.bias = dbias;
  if (resid) { p.residual = R.d; p.ldr = N; p.beta = 1.f; }
  if (rowb_silu) { p.rowbias = RB.d; p.ldrb = N; p.rows_per_batch = rpb; p.act = ACT_SILU; }
  std::vector<void*> C(ncfg);
  std::vector<std::vector<uint16_t>> h(ncfg, std::vector<uint16_t>((size_t)M * N)), h2(1, std::vector<uint16_t>((size_t)M * N));
  std::vector<std::vector<float>> t(ncfg);
  std::vector<int> rcs(ncfg, 0);
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int c = 0; c < ncfg; ++c) { HIPCHK(hipMalloc(&C[c], (size_t)M * N * 2)); HIPCHK(hipMemset(C[c], 0, (size_t)M * N * 2)); }
  const int iters = 10;
  for (int r = 0; r < rounds + 1; ++r)
    for (int c = 0; c < ncfg; ++c) {
      g_gemm_force_cfg = cfgs[c]; p.C = C[c];
      HIPCHK(hipEventRecord(e0, 0));
      for (int i = 0; i < iters; ++i) rcs[c] |= launch_gemm(p, dtype, 0);
      HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
      float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) t[c].push_back(ms / iters * 1e3f);      // round 0 = warm-up (clock ramp, first-touch)
    }
  g_gemm_force_cfg = -1;
  const double fl = 2.0 * M * N * ((double)taps * K1 + K2);
  for (int c = 0; c < ncfg; ++c) HIPCHK(hipMemcpy(h[c].data(), C[c], h[c].size() * 2, hipMemcpyDeviceToHost));
  for (int c = 0; c < ncfg; ++c) {
    // bitwise repeat: three more single launches into a scrubbed buffer
    size_t rep_diff = 0;
    for (int k = 0; k < 3; ++k) {
      HIPCHK(hipMemset(C[c], 0xff, (size_t)M * N * 2));
      g_gemm_force_cfg = cfgs[c]; p.C = C[c];
      launch_gemm(p, dtype, 0); HIPCHK(hipDeviceSynchronize());
      HIPCHK(hipMemcpy(h2[0].data(), C[c], h2[0].size() * 2, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < h2[0].size(); ++i) rep_diff += h2[0][i] != h[c][i];
    }
    g_gemm_force_cfg = -1;
    double num = 0, den = 0; size_t nz = 0;
    for (size_t i = 0; i < h[c].size(); ++i) {
      const double a = h_bf2f(h[c][i]), b = h_bf2f(h[0][i]); num += (a - b) * (a - b); den += b * b; nz += h[c][i] != 0;
    }
    std::sort(t[c].begin(), t[c].end());
    const float med = t[c][t[c].size() / 2], mn = t[c][0];
    const double rel = std::sqrt(num / (den + 1e-30));
    const bool ok = rcs[c] == 0 && rep_diff == 0 && rel < 4e-3 && nz > h[c].size() / 2;
    printf("[%s] %-40s cfg %2d: median %8.1f us (%7.1f TF/s)  min %8.1f us  | vs cfg %2d rel_l2 %.2e | repeat diffs %zu\n", ok ? "PASS" : "FAIL",
           name, cfgs[c], med, fl / med * 1e-6, mn, cfgs[0], rel, rep_diff);
    if (!ok) g_fail++;
  }
  for (int c = 0; c < ncfg; ++c) hipFree(C[c]);
  hipFree(A.d); hipFree(Wt.d); hipFree(dbias); if (K2) { hipFree(A2.d); hipFree(W2.d); } if (resid) hipFree(R.d); if (rowb_silu) hipFree(RB.d);
}

// one big conv against the CPU double reference on sampled output elements (full tensors are too slow on the host)
static void conv_sampled(const char* name, int cfg, int B, int H, int W, int C, int N, int nsamp) {
  const int M = B * H * W;
  Buf X, Wt, Cb; std::vector<float> hb(N);
  X.init((size_t)M * C, CL_BF16); Wt.init((size_t)N * 9 * C, CL_BF16, 0.05f);
  for (auto& v : hb) v = frand();
  float* dbias; HIPCHK(hipMalloc(&dbias, N * 4)); HIPCHK(hipMemcpy(dbias, hb.data(), N * 4, hipMemcpyHostToDevice));
  Cb.init((size_t)M * N, CL_F32, 1.0f, true);
  GemmParams p{}; p.A1 = X.d; p.lda1 = C; p.K1 = C; p.W1 = Wt.d; p.ldw1 = 9 * C; p.M = M; p.N = N; p.mode = GEMM_CONV_S1;
  p.B = B; p.Hin = H; p.Win = W; p.Hout = H; p.Wout = W; p.zero_page = g_zero; p.bias = dbias;
  p.alpha = 1.f; p.C = Cb.d; p.ldc = N; p.out_f32 = 1; p.splitk = 1;
  g_gemm_force_cfg = cfg;
  int rc = launch_gemm(p, CL_BF16, 0);
  g_gemm_force_cfg = -1;
  HIPCHK(hipDeviceSynchronize());
  if (rc) { printf("[FAIL] %s rc=%d\n", name, rc); g_fail++; return; }
  Cb.download(CL_F32);
  double num = 0, den = 0;
  for (int sidx = 0; sidx < nsamp; ++sidx) {
    rng_state = rng_state * 1664525u + 1013904223u; const int m = (int)((rng_state >> 4) % (unsigned)M);
    rng_state = rng_state * 1664525u + 1013904223u; const int n = (int)((rng_state >> 4) % (unsigned)N);
    // corners and edges first: they exercise the tap mask
    const int mm = sidx < 8 ? ((sidx & 1) ? M - 1 - (sidx >> 1) * (W - 1) : (sidx >> 1) * (W - 1)) : m;
    const int ox = mm % W, oy = (mm / W) % H, b = mm / (W * H);
    double sacc = hb[n];
    for (int ky = 0; ky < 3; ++ky) for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy + ky - 1, ix = ox + kx - 1;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const float* xp = &X.h[(((size_t)b * H + iy) * W + ix) * C];
      const float* wp = &Wt.h[((size_t)n * 9 + ky * 3 + kx) * C];
      for (int c = 0; c < C; ++c) sacc += (double)xp[c] * wp[c];
    }
    const double d = Cb.h[(size_t)mm * N + n] - sacc; num += d * d; den += sacc * sacc;
  }
  report(name, num, den, 2e-5);
  hipFree(X.d); hipFree(Wt.d); hipFree(Cb.d); hipFree(dbias);
}


#ifdef W4_PROBE
// s_memtime stamps of every tile (wave 0): entry | stage 0 landed | main loop done | stores retired, next to the wall time of the
// same launch: cycles / wall = the clock the chip actually held
static void w4_timing_case(const char* tag, int B_ = 8, int H_ = 64, int N_ = 320, int C_ = 320) {
  const int M = B_ * H_ * H_, N = N_, K1 = C_;
  const int nst = 9 * C_ / 64;
  Buf A, Wt, Cb; A.init((size_t)M * K1, CL_BF16); Wt.init((size_t)N * 9 * K1, CL_BF16, 0.05f); Cb.init((size_t)M * N, CL_BF16, 1.f, true);
  GemmParams p{}; p.A1 = A.d; p.lda1 = K1; p.K1 = K1; p.W1 = Wt.d; p.ldw1 = 9 * K1; p.M = M; p.N = N; p.mode = GEMM_CONV_S1;
  p.B = B_; p.Hin = H_; p.Win = H_; p.Hout = H_; p.Wout = H_; p.zero_page = g_zero; p.alpha = 1.f; p.C = Cb.d; p.ldc = N; p.splitk = 1;
  const long nv = 256;
  unsigned long long* tb; HIPCHK(hipMalloc(&tb, nv * 32)); HIPCHK(hipMemset(tb, 0, nv * 32));
  g_gemm_force_cfg = 40;
  for (int i = 0; i < 10; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipDeviceSynchronize());
  cl::w4_timing_set(tb);
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipEventRecord(e0, 0));
  for (int i = 0; i < 10; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  HIPCHK(hipDeviceSynchronize());
  cl::w4_timing_set(nullptr);
  g_gemm_force_cfg = -1;
  std::vector<unsigned long long> tt(nv * 4);
  HIPCHK(hipMemcpy(tt.data(), tb, nv * 32, hipMemcpyDeviceToHost));
  double ph[3] = {0, 0, 0};
  for (long i = 0; i < nv; ++i)
    for (int k = 0; k < 3; ++k) ph[k] += (double)(tt[i * 4 + k + 1] - tt[i * 4 + k]);
  const double tot = (ph[0] + ph[1] + ph[2]) / nv;
  printf("[TIMING] %-52s wall %6.1f us | per tile: entry -> stage 0 landed %6.0f | main loop (%d stages) %6.0f = %5.0f per stage | epilogue %6.0f | "
         "sum %6.0f cycles -> %.2f GHz\n", tag, ms * 1e3, ph[0] / nv, nst, ph[1] / nv, ph[1] / nv / nst, ph[2] / nv, tot, tot / (ms * 1e3) * 1e-3);
  hipFree(tb); hipFree(A.d); hipFree(Wt.d); hipFree(Cb.d);
}
#endif


#if defined(W4_PROBE) && defined(FL_TIMING)
namespace cl { void fl_timing_set(unsigned long long* buf); }
// the same cycles-against-wall accounting for the ping-pong tile kernel (configuration 16): stamps 0 entry | 2 first stage landed |
// 3 main loop done | 4 epilogue issued
static void fl_timing_case16(int B_ = 8, int H_ = 64, int N_ = 320, int C_ = 320) {
  const int M = B_ * H_ * H_, N = N_, K1 = C_;
  const int nst = 9 * C_ / 64;
  Buf A, Wt, Cb; A.init((size_t)M * K1, CL_BF16); Wt.init((size_t)N * 9 * K1, CL_BF16, 0.05f); Cb.init((size_t)M * N, CL_BF16, 1.f, true);
  GemmParams p{}; p.A1 = A.d; p.lda1 = K1; p.K1 = K1; p.W1 = Wt.d; p.ldw1 = 9 * K1; p.M = M; p.N = N; p.mode = GEMM_CONV_S1;
  p.B = B_; p.Hin = H_; p.Win = H_; p.Hout = H_; p.Wout = H_; p.zero_page = g_zero; p.alpha = 1.f; p.C = Cb.d; p.ldc = N; p.splitk = 1;
  const long nv = 256;
  unsigned long long* tb; HIPCHK(hipMalloc(&tb, nv * 64)); HIPCHK(hipMemset(tb, 0, nv * 64));
  g_gemm_force_cfg = 16;
  for (int i = 0; i < 10; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipDeviceSynchronize());
  cl::fl_timing_set(tb);
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipEventRecord(e0, 0));
  for (int i = 0; i < 10; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  HIPCHK(hipDeviceSynchronize());
  cl::fl_timing_set(nullptr);
  g_gemm_force_cfg = -1;
  std::vector<unsigned long long> tt(nv * 8);
  HIPCHK(hipMemcpy(tt.data(), tb, nv * 64, hipMemcpyDeviceToHost));
  double ph[3] = {0, 0, 0};
  for (long i = 0; i < nv; ++i) { ph[0] += (double)(tt[i * 8 + 2] - tt[i * 8]); ph[1] += (double)(tt[i * 8 + 3] - tt[i * 8 + 2]); ph[2] += (double)(tt[i * 8 + 4] - tt[i * 8 + 3]); }
  const double tot = (ph[0] + ph[1] + ph[2]) / nv;
  printf("[TIMING] %-52s wall %6.1f us | per tile: entry -> stage 0 landed %6.0f | main loop (%d stages) %6.0f = %5.0f per stage | epilogue %6.0f | "
         "sum %6.0f cycles -> %.2f GHz\n", "cfg 16 (ping-pong tiles, stamps cost ~1 %)", ms * 1e3, ph[0] / nv, nst, ph[1] / nv, ph[1] / nv / nst, ph[2] / nv, tot, tot / (ms * 1e3) * 1e-3);
  hipFree(tb); hipFree(A.d); hipFree(Wt.d); hipFree(Cb.d);
}
#endif

#ifdef FL_TIMING
namespace cl { void fl_timing_set(unsigned long long* buf); }
#endif

// persistent vs one-tile-per-workgroup form of the same tile code: results must be IDENTICAL (same tiles, same math)
static void persist_case(const char* name, int M, int N, int K1, int K2, int act, int cfg, int pcfg, bool timing) {
  Buf A, Wt, A2, W2, C0, C1;
  A.init((size_t)M * K1, CL_BF16); Wt.init((size_t)N * K1, CL_BF16, 0.05f);
  if (K2) { A2.init((size_t)M * K2, CL_BF16); W2.init((size_t)N * K2, CL_BF16, 0.05f); }
  const int No = act == ACT_GEGLU ? N / 2 : N;
  C0.init((size_t)M * No, CL_BF16, 1.f, true); C1.init((size_t)M * No, CL_BF16, 1.f, true);
  GemmParams p{}; p.A1 = A.d; p.lda1 = K1; p.K1 = K1; p.W1 = Wt.d; p.ldw1 = K1; p.M = M; p.N = N; p.mode = GEMM_LINEAR;
  if (K2) { p.A2 = A2.d; p.lda2 = K2; p.K2 = K2; p.W2 = W2.d; p.ldw2 = K2; }
  p.zero_page = g_zero; p.alpha = 1.f; p.ldc = No; p.splitk = 1; p.act = act;
  float ms[2]; int rc[2];
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  for (int v = 0; v < 2; ++v) {
    g_gemm_force_cfg = v ? pcfg : cfg; p.C = v ? C1.d : C0.d;
    for (int i = 0; i < 3; ++i) rc[v] = launch_gemm(p, CL_BF16, 0);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(e0, 0));
    for (int i = 0; i < 20; ++i) launch_gemm(p, CL_BF16, 0);
    HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipEventElapsedTime(&ms[v], e0, e1)); ms[v] /= 20;
  }
  std::vector<uint16_t> h0((size_t)M * No), h1((size_t)M * No);
  HIPCHK(hipMemcpy(h0.data(), C0.d, h0.size() * 2, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(h1.data(), C1.d, h1.size() * 2, hipMemcpyDeviceToHost));
  size_t diff = 0, nz = 0;
  for (size_t i = 0; i < h0.size(); ++i) { diff += h0[i] != h1[i]; nz += h0[i] != 0; }
  const bool ok = rc[0] == 0 && rc[1] == 0 && diff == 0 && nz > h0.size() / 2;
  double fl = 2.0 * M * N * ((double)K1 + K2);
  printf("[%s] %-40s cfg %2d: %7.1f us %7.1f TF | persistent cfg %2d: %7.1f us %7.1f TF | %zu differing of %zu\n", ok ? "PASS" : "FAIL",
         name, cfg, ms[0] * 1e3, fl / ms[0] * 1e-9, pcfg, ms[1] * 1e3, fl / ms[1] * 1e-9, diff, h0.size());
  if (!ok) g_fail++;
#ifdef FL_TIMING
  if (timing) {
    for (int v = 0; v < 2; ++v) {
      const int bm = ((v ? pcfg : cfg) == 16 || (v ? pcfg : cfg) == 25) ? 256 : 128;
      const long nv = (long)((M + bm - 1) / bm) * ((N + 159) / 160);
      unsigned long long* tb; HIPCHK(hipMalloc(&tb, nv * 64)); HIPCHK(hipMemset(tb, 0, nv * 64));
      cl::fl_timing_set(tb);
      g_gemm_force_cfg = v ? pcfg : cfg; p.C = C1.d;
      launch_gemm(p, CL_BF16, 0); HIPCHK(hipDeviceSynchronize());
      cl::fl_timing_set(nullptr);
      std::vector<unsigned long long> t(nv * 8);
      HIPCHK(hipMemcpy(t.data(), tb, nv * 64, hipMemcpyDeviceToHost)); HIPCHK(hipFree(tb));
      // phases per tile, and per-CU turnaround between consecutive tiles on the same compute unit
      double ph[4] = {0, 0, 0, 0}; unsigned long long tmin = ~0ull, tmax = 0;
      std::vector<std::pair<unsigned long long, long>> order;
      for (long i = 0; i < nv; ++i) {
        for (int k = 0; k < 4; ++k) ph[k] += (double)(t[i * 8 + k + 1] - t[i * 8 + k]);
        tmin = std::min(tmin, t[i * 8]); tmax = std::max(tmax, t[i * 8 + 4]);
        order.push_back({t[i * 8], i});
      }
      std::sort(order.begin(), order.end());
      std::map<unsigned long long, unsigned long long> last_end; double gap = 0; long ngap = 0;
      for (auto& o : order) {
        const long i = o.second;
        const unsigned long long hw = t[i * 8 + 5], xcc = t[i * 8 + 6] & 0xf;
        const unsigned long long cu = (xcc << 16) | (hw & 0xff00) | ((hw >> 4) & 0x3 ? 0 : 0);   // xcc, se/sh/cu bits
        auto it = last_end.find(cu);
        if (it != last_end.end() && t[i * 8] > it->second) { gap += (double)(t[i * 8] - it->second); ++ngap; }
        if (it == last_end.end() || t[i * 8 + 4] > it->second) last_end[cu] = t[i * 8 + 4];
      }
      printf("       timing cfg %2d: %ld tiles on %zu CUs, span %.0f ticks; per tile: setup %.0f | first stage %.0f | main loop %.0f | epilogue %.0f | "
             "idle between tiles on a CU %.0f (n=%ld)  [s_memtime ticks]\n", v ? pcfg : cfg, nv, last_end.size(), (double)(tmax - tmin),
             ph[0] / nv, ph[1] / nv, ph[2] / nv, ph[3] / nv, ngap ? gap / ngap : 0.0, ngap);
    }
  }
#endif
  g_gemm_force_cfg = -1;
  hipFree(A.d); hipFree(Wt.d); hipFree(C0.d); hipFree(C1.d); if (K2) { hipFree(A2.d); hipFree(W2.d); }
}


#ifdef FL_TIMING
// s_memtime breakdown of one short-K linear launch on the full-line tile kernels (FL_STAMP 0 entry | 1 set-up done, ring fill issued |
// 2 stage 0 landed | 3 K loop done | 4 stores issued | 7 stores retired), per workgroup, plus the launch's own span (first entry ->
// last retire) against the wall time per launch in a back-to-back train of launches.
static void fl_timing_linear(const char* name, int M, int N, int K1, int K2, int cfg, int sk, bool resid) {
  Buf A, Wt, A2, W2, Cb, R; A.init((size_t)M * K1, CL_BF16); Wt.init((size_t)N * K1, CL_BF16, 0.05f); Cb.init((size_t)M * N, CL_BF16, 1.f, true);
  if (K2) { A2.init((size_t)M * K2, CL_BF16); W2.init((size_t)N * K2, CL_BF16, 0.05f); }
  if (resid) R.init((size_t)M * N, CL_BF16);
  GemmParams p{}; p.A1 = A.d; p.lda1 = K1; p.K1 = K1; p.W1 = Wt.d; p.ldw1 = K1; p.M = M; p.N = N; p.mode = GEMM_LINEAR;
  if (K2) { p.A2 = A2.d; p.lda2 = K2; p.K2 = K2; p.W2 = W2.d; p.ldw2 = K2; }
  if (resid) { p.residual = R.d; p.ldr = N; p.beta = 1.f; }
  p.alpha = 1.f; p.C = Cb.d; p.ldc = N; p.splitk = sk;
  const long nv = 1 << 16;
  unsigned long long* tb; HIPCHK(hipMalloc(&tb, nv * 64)); HIPCHK(hipMemset(tb, 0, nv * 64));
  g_gemm_force_cfg = cfg;
  for (int i = 0; i < 10; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipDeviceSynchronize());
  hipEvent_t e0, e1; HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
  float ms0; HIPCHK(hipEventElapsedTime(&ms0, e0, e1)); ms0 /= 20;
  cl::fl_timing_set(tb);
  for (int i = 0; i < 5; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipEventRecord(e0, 0));
  for (int i = 0; i < 10; ++i) launch_gemm(p, CL_BF16, 0);
  HIPCHK(hipEventRecord(e1, 0)); HIPCHK(hipEventSynchronize(e1));
  float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  HIPCHK(hipDeviceSynchronize());
  cl::fl_timing_set(nullptr);
  g_gemm_force_cfg = -1;
  std::vector<unsigned long long> tt(nv * 8);
  HIPCHK(hipMemcpy(tt.data(), tb, nv * 64, hipMemcpyDeviceToHost));
  double ph[5] = {0, 0, 0, 0, 0};
  unsigned long long first = ~0ull, last = 0, last_start = 0;
  long n = 0;
  for (long i = 0; i < nv; ++i) {
    const unsigned long long* t = &tt[i * 8];
    if (!t[0] || !t[7]) continue;
    ++n;
    ph[0] += (double)(t[1] - t[0]); ph[1] += (double)(t[2] - t[1]); ph[2] += (double)(t[3] - t[2]); ph[3] += (double)(t[4] - t[3]);
    ph[4] += (double)(t[7] - t[4]);
    first = std::min(first, t[0]); last = std::max(last, t[7]); last_start = std::max(last_start, t[0]);
  }
  if (!n) { printf("[SHORTK] %-44s cfg %d: no stamps (not a full-line tile configuration)\n", name, cfg); return; }
  const double tot = (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / n;
  printf("[SHORTK] %-44s cfg %2d sk %d: wall %5.1f us (%5.1f with stamps) | %4ld workgroups | per workgroup, cycles (100 MHz-independent: shader clock): "
         "set-up %5.0f | ring fill %5.0f | K loop (%d stages) %6.0f | epilogue issue %5.0f | store drain %5.0f | sum %6.0f | "
         "launch span %6.0f (last workgroup starts at %5.0f)\n", name, cfg, sk, ms0 * 1e3, ms * 1e3, n, ph[0] / n, ph[1] / n, (K1 + K2) / 64, ph[2] / n,
         ph[3] / n, ph[4] / n, tot, (double)(last - first), (double)(last_start - first));
  hipFree(tb);
}
#endif

int main(int argc, char** argv) {
  hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s  CUs=%d  arch=%s\n", prop.name, prop.multiProcessorCount, prop.gcnArchName);
  HIPCHK(hipMalloc(&g_zero, 16384)); HIPCHK(hipMemset(g_zero, 0, 16384));
  void* ws; HIPCHK(hipMalloc(&ws, 64 << 20)); gemm_set_workspace(ws, 64 << 20);

  if (argc > 3 && !strcmp(argv[1], "--one")) {   // profiling aid: one timed shape under one forced config
    g_gemm_force_cfg = atoi(argv[2]);
    const int which = atoi(argv[3]);
    if (which == 0) time_case("conv bf16 320->320 @64^2 B8", CL_BF16, GEMM_CONV_S1, 8 * 64 * 64, 320, 320, 8, 64, 64);
    if (which == 1) time_case("gemm bf16 4096^3", CL_BF16, GEMM_LINEAR, 4096, 4096, 4096, 0, 0, 0);
    if (which == 2) time_case("conv bf16 1280->1280 @16^2 B8", CL_BF16, GEMM_CONV_S1, 8 * 16 * 16, 1280, 1280, 8, 16, 16);
    if (which == 3) time_case("gemm bf16 32768x2560x320 (GEGLU proj)", CL_BF16, GEMM_LINEAR, 32768, 2560, 320, 0, 0, 0);
    if (which == 4) {
      time_case("gemm bf16 131072x2560x320 normal", CL_BF16, GEMM_LINEAR, 131072, 2560, 320, 0, 0, 0);
      g_probe_act = 77;
      time_case("gemm bf16 131072x2560x320 no stores", CL_BF16, GEMM_LINEAR, 131072, 2560, 320, 0, 0, 0);
      g_probe_act = 0;
      time_case("gemm bf16 131072x320x320 normal", CL_BF16, GEMM_LINEAR, 131072, 320, 320, 0, 0, 0);
      g_probe_act = 77;
      time_case("gemm bf16 131072x320x320 no stores", CL_BF16, GEMM_LINEAR, 131072, 320, 320, 0, 0, 0);
      g_probe_act = 0;
      time_case("gemm bf16 131072x2560x1280 normal", CL_BF16, GEMM_LINEAR, 131072, 2560, 1280, 0, 0, 0);
      g_probe_act = 77;
      time_case("gemm bf16 131072x2560x1280 no stores", CL_BF16, GEMM_LINEAR, 131072, 2560, 1280, 0, 0, 0);
      g_probe_act = 0;
    }
    if (which == 5) {   // round 5: the training-batch wide-N / short-K products, with and without their stores
      const int sh[][3] = {{32768, 2560, 320}, {32768, 1280, 320}, {32768, 960, 320}, {8192, 5120, 640}, {2048, 10240, 1280}};
      for (auto& q : sh) {
        g_probe_act = 0;  time_case("gemm bf16 wide-N normal", CL_BF16, GEMM_LINEAR, q[0], q[1], q[2], 0, 0, 0);
        g_probe_act = 77; time_case("gemm bf16 wide-N no stores", CL_BF16, GEMM_LINEAR, q[0], q[1], q[2], 0, 0, 0);
      }
      g_probe_act = 0;
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--wgrad5")) {   // 64-row steps vs 32-row steps
    const int forms[][2] = {{32, 4}, {32, 3}, {64, 2}};
    for (auto& f : forms) {
      const int rows = f[0];
      g_wgrad_rows = rows; g_wgrad_ring = f[1];
      printf("---- rows per step %d, ring %d\n", rows, f[1]);
      case_wgrad("wgrad 300x200x136", 300, 200, 136, 1.0f);
      case_wgrad("wgrad 616x320x128 alpha 0.5", 616, 320, 128, 0.5f);
      case_wgrad("wgrad M=8 1280x128", 8, 1280, 128, 1.0f);
      case_wgrad("wgrad 4096x128x320", 4096, 128, 320, 1.0f);
      case_wgrad("wgrad 1000x8x40 (tiny)", 1000, 8, 40, 1.0f);
      case_wgrad("wgrad 32768x320x128 (dB @64^2)", 32768, 320, 128, 1.0f, true);
      case_wgrad("wgrad 32768x128x320 (dA @64^2)", 32768, 128, 320, 1.0f, true);
      case_wgrad("wgrad 32768x2560x128 (dB GEGLU)", 32768, 2560, 128, 1.0f, true);
      case_wgrad("wgrad 32768x320x320 (zero conv)", 32768, 320, 320, 1.0f, true);
      case_wgrad("wgrad 512x1280x1280 (zero conv 8^2)", 512, 1280, 1280, 1.0f, true);
    }
    g_wgrad_rows = 32; g_wgrad_ring = 4;
    wgrad_tap_group(8, 64, 320, 320);
    wgrad_tap_group(8, 32, 640, 640);
    wgrad_tap_group(8, 16, 1280, 1280);
    wgrad_tap_group(8, 8, 1280, 1280);
    printf("%s\n", g_fail ? "WGRAD PROBE: FAILURES" : "WGRAD PROBE: all pass");
    return g_fail ? 1 : 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--xs")) {   // x-stationary kernel: correctness on ragged / grouped cases, then the production shapes
    case_xs("xs 300x320x320 bias", 300, 320, 320, 0, true, 1, 1.f, 0, 0);
    case_xs("xs 1000x960x320+128 groups 3 alpha_n", 1000, 960, 320, 128, true, 3, 0.37f, 320, 0);
    case_xs("xs 129x64x320 no bias alpha", 129, 64, 320, 0, false, 1, 0.5f, 0, 0);
    case_xs("xs 640x2560x320+128 nsplit 4", 640, 2560, 320, 128, true, 1, 1.f, 0, 4);
    case_xs("xs 2048x3200x320 (run > 80 chunks)", 2048, 3200, 320, 0, true, 1, 1.f, 0, 1);
    case_xs("xs 500x1920x640 groups 3", 500, 1920, 640, 0, true, 1, 0.25f, 640, 0);
    case_xs("xs 384x1280x640+128 groups 2", 384, 1280, 640, 128, true, 2, 1.f, 0, 0);
    case_xs("xs 256x5120x640+128", 256, 5120, 640, 128, true, 1, 1.f, 0, 0);
    g_xs_beta = 1.0f;  case_xs("xs res 300x320x320 bias beta 1", 300, 320, 320, 0, true, 1, 1.f, 0, 0);
    g_xs_beta = -0.5f; case_xs("xs res 1000x320x320+128 alpha", 1000, 320, 320, 128, true, 1, 0.7f, 0, 2);
    g_xs_beta = 1.0f;  case_xs("xs res 515x640x640+128", 515, 640, 640, 128, true, 1, 1.f, 0, 0);
    g_xs_beta = 2.0f;  case_xs("xs res 640x1280x640 nsplit 4", 640, 1280, 640, 0, false, 1, 1.f, 0, 4);
    g_xs_geglu = true; case_xs("xs geglu 300x(2*320)x320", 300, 640, 320, 0, true, 1, 1.f, 0, 0);
    g_xs_geglu = true; case_xs("xs geglu 1000x(2*1280)x320+128", 1000, 2560, 320, 128, true, 1, 1.f, 0, 0);
    g_xs_geglu = true; case_xs("xs geglu 384x(2*2560)x640 nsplit 3", 384, 5120, 640, 0, true, 1, 1.f, 0, 3);
    g_xs_geglu = true; case_xs("xs geglu 257x(2*96)x640+128 (3 blocks)", 257, 192, 640, 128, false, 1, 1.f, 0, 1);
    g_xs_ln = true; case_xs("xs LN 300x960x320 alpha_n", 300, 960, 320, 0, true, 1, 0.3f, 320, 0);
    g_xs_ln = true; case_xs("xs LN 1000x320x320 nsplit 2", 1000, 320, 320, 0, false, 1, 1.f, 0, 2);
    g_xs_ln = true; case_xs("xs LN 515x1920x640", 515, 1920, 640, 0, true, 1, 1.f, 0, 0);
    g_xs_ln = true; g_xs_geglu = true; case_xs("xs LN geglu 700x(2*1280)x320", 700, 2560, 320, 0, true, 1, 1.f, 0, 0);
    g_xs_ln = true; g_xs_geglu = true; case_xs("xs LN geglu 384x(2*2560)x640", 384, 5120, 640, 0, true, 1, 1.f, 0, 0);
    const bool quick = argc > 2 && !strcmp(argv[2], "quick");
    const int ns_all[] = {0, 1, 2, 4};
    for (int n : ns_all) {
      if (quick && n) break;
      char nm[96];
      snprintf(nm, sizeof nm, "xs 32768x2560x320 nsplit %d", n);       case_xs(nm, 32768, 2560, 320, 0, true, 1, 1.f, 0, n, true);
      snprintf(nm, sizeof nm, "xs 32768x2560x320+128 nsplit %d", n);   case_xs(nm, 32768, 2560, 320, 128, true, 1, 1.f, 0, n, true);
      snprintf(nm, sizeof nm, "xs 32768x1280x320 nsplit %d", n);       case_xs(nm, 32768, 1280, 320, 0, false, 1, 1.f, 0, n, true);
      snprintf(nm, sizeof nm, "xs 32768x960x320+128 g3 nsplit %d", n); case_xs(nm, 32768, 960, 320, 128, false, 3, 0.2f, 320, n, true);
      snprintf(nm, sizeof nm, "xs 32768x320x320 nsplit %d", n);        case_xs(nm, 32768, 320, 320, 0, true, 1, 1.f, 0, n, true);
      snprintf(nm, sizeof nm, "xs 8192x5120x640 nsplit %d", n);        case_xs(nm, 8192, 5120, 640, 0, true, 1, 1.f, 0, n, true);
      snprintf(nm, sizeof nm, "xs 8192x5120x640+128 nsplit %d", n);    case_xs(nm, 8192, 5120, 640, 128, true, 1, 1.f, 0, n, true);
      snprintf(nm, sizeof nm, "xs 8192x1920x640+128 g3 nsplit %d", n); case_xs(nm, 8192, 1920, 640, 128, false, 3, 0.2f, 640, n, true);
      snprintf(nm, sizeof nm, "xs 8192x2560x640 nsplit %d", n);        case_xs(nm, 8192, 2560, 640, 0, false, 1, 1.f, 0, n, true);
      g_xs_beta = 1.f; snprintf(nm, sizeof nm, "xs res 32768x320x320 nsplit %d", n);     case_xs(nm, 32768, 320, 320, 0, true, 1, 1.f, 0, n, true);
      g_xs_beta = 1.f; snprintf(nm, sizeof nm, "xs res 32768x320x320+128 nsplit %d", n); case_xs(nm, 32768, 320, 320, 128, true, 1, 1.f, 0, n, true);
      g_xs_beta = 1.f; snprintf(nm, sizeof nm, "xs res 8192x640x640 nsplit %d", n);      case_xs(nm, 8192, 640, 640, 0, true, 1, 1.f, 0, n, true);
      g_xs_beta = 1.f; snprintf(nm, sizeof nm, "xs res 131072x320x320 nsplit %d", n);    case_xs(nm, 131072, 320, 320, 0, true, 1, 1.f, 0, n, true);
      g_xs_geglu = true; snprintf(nm, sizeof nm, "xs geglu 32768x2560x320 nsplit %d", n);   case_xs(nm, 32768, 2560, 320, 0, true, 1, 1.f, 0, n, true);
      g_xs_geglu = true; snprintf(nm, sizeof nm, "xs geglu 131072x2560x320 nsplit %d", n);  case_xs(nm, 131072, 2560, 320, 0, true, 1, 1.f, 0, n, true);
      g_xs_geglu = true; snprintf(nm, sizeof nm, "xs geglu 32768x5120x640 nsplit %d", n);   case_xs(nm, 32768, 5120, 640, 0, true, 1, 1.f, 0, n, true);
      g_xs_geglu = true; snprintf(nm, sizeof nm, "xs geglu 8192x5120x640 nsplit %d", n);    case_xs(nm, 8192, 5120, 640, 0, true, 1, 1.f, 0, n, true);
    }
    g_xs_ln = true; case_xs("xs LN 131072x960x320", 131072, 960, 320, 0, true, 1, 0.3f, 320, 0, true);
    g_xs_ln = true; case_xs("xs LN 131072x320x320", 131072, 320, 320, 0, false, 1, 1.f, 0, 0, true);
    g_xs_ln = true; case_xs("xs LN 32768x1920x640", 32768, 1920, 640, 0, true, 1, 1.f, 0, 0, true);
    g_xs_ln = true; g_xs_geglu = true; case_xs("xs LN geglu 131072x2560x320", 131072, 2560, 320, 0, true, 1, 1.f, 0, 0, true);
    g_xs_ln = true; g_xs_geglu = true; case_xs("xs LN geglu 32768x2560x320", 32768, 2560, 320, 0, true, 1, 1.f, 0, 0, true);
    g_xs_ln = true; g_xs_geglu = true; case_xs("xs LN geglu 32768x5120x640", 32768, 5120, 640, 0, true, 1, 1.f, 0, 0, true);
    // the tile kernels' fused GEGLU on the same products (rows permuted per 160-column tile), for the comparison
    g_gemm_force_cfg = -1;
    g_probe_act = ACT_GEGLU;
    time_case("tile GEGLU 32768x2560x320", CL_BF16, GEMM_LINEAR, 32768, 2560, 320, 0, 0, 0);
    time_case("tile GEGLU 131072x2560x320", CL_BF16, GEMM_LINEAR, 131072, 2560, 320, 0, 0, 0);
    time_case("tile GEGLU 32768x5120x640", CL_BF16, GEMM_LINEAR, 32768, 5120, 640, 0, 0, 0);
    time_case("tile GEGLU 8192x5120x640", CL_BF16, GEMM_LINEAR, 8192, 5120, 640, 0, 0, 0);
    g_probe_act = 0;
    printf("%s\n", g_fail ? "XS PROBE: FAILURES" : "XS PROBE: all pass");
    return g_fail ? 1 : 0;
  }
#ifdef FL_TIMING
  if (argc > 1 && !strcmp(argv[1], "--shortk")) {   // verdict r5 item 2: where the fixed cost of the short-K linears sits
    const int cfgs[] = {35, 36, 1, 21, 24, 16, 10};
    for (int c : cfgs) fl_timing_linear("linear 2048x1280x1280", 2048, 1280, 1280, 0, c, 1, false);
    for (int c : cfgs) fl_timing_linear("linear 2048x1280x1280 + residual", 2048, 1280, 1280, 0, c, 1, true);
    for (int c : cfgs) fl_timing_linear("linear 2048x1280x1280+128 (LoRA)", 2048, 1280, 1280, 128, c, 1, false);
    for (int c : cfgs) fl_timing_linear("linear 512x1280x1280", 512, 1280, 1280, 0, c, 1, false);
    for (int c : cfgs) fl_timing_linear("linear 8192x640x640", 8192, 640, 640, 0, c, 1, false);
    for (int c : cfgs) fl_timing_linear("linear 32768x320x320", 32768, 320, 320, 0, c, 1, false);
    // the same products with K cut to ONE stage: what a launch costs when there is nothing to compute
    for (int c : {35, 36}) fl_timing_linear("linear 2048x1280x64 (one stage)", 2048, 1280, 64, 0, c, 1, false);
    return 0;
  }
#endif
  if (argc > 1 && !strcmp(argv[1], "--w4")) {   // loader / consumer kernel: correctness, then interleaved A/B against the ping-pong tiles
    const bool quick = argc > 2 && !strcmp(argv[2], "quick");
    const bool abl_only = argc > 2 && !strcmp(argv[2], "abl");
    if (abl_only) goto w4_ablations;
    {
    const int cf[] = {40, 41, 47, 48};
    for (int c : cf) { char tag[32]; snprintf(tag, 32, "forced cfg %d", c); g_gemm_force_cfg = c; correctness_suite(tag); }
    g_gemm_force_cfg = -1;
    conv_sampled("conv 320->320 @64^2 B2 sampled vs fp64, cfg 40", 40, 2, 64, 64, 320, 320, 4000);
    conv_sampled("conv 128->128 @96x80 B1 sampled vs fp64, cfg 41", 41, 1, 96, 80, 128, 128, 4000);
    conv_sampled("conv 640->320 ragged 3x33x31 sampled, cfg 40", 40, 3, 33, 31, 640, 320, 4000);
    conv_sampled("conv 640->640 @32^2 B2 sampled (halo), cfg 40", 40, 2, 32, 32, 640, 640, 4000);
    conv_sampled("conv 1280->320 @16^2 B4 sampled (halo, split-K), cfg 40", 40, 4, 16, 16, 1280, 320, 4000);
    conv_sampled("conv 64->160 @64^2 B1 sampled (halo, one chunk), cfg 40", 40, 1, 64, 64, 64, 160, 4000);
    conv_sampled("conv 128->128 @32^2 B3 sampled (halo), cfg 41", 41, 3, 32, 32, 128, 128, 4000);
    conv_sampled("conv 320->320 @64^2 B32 sampled, persistent cfg 47 (4 tiles per CU)", 47, 32, 64, 64, 320, 320, 6000);
    conv_sampled("conv 128->128 @256^2 B2 sampled, persistent cfg 48", 48, 2, 256, 256, 128, 128, 6000);
    conv_sampled("conv 640->320 ragged 5x33x31 sampled, persistent cfg 47", 47, 5, 33, 31, 640, 320, 4000);
    conv_sampled("conv 640->640 @32^2 B32 sampled, persistent halo cfg 47 (2 tiles per CU)", 47, 32, 32, 32, 640, 640, 6000);
    conv_sampled("conv 320->320 @16^2 B160 sampled, persistent halo cfg 47", 47, 160, 16, 16, 320, 320, 6000);
    conv_sampled("conv 128->128 @64^2 B24 sampled, persistent halo cfg 48", 48, 24, 64, 64, 128, 128, 6000);
    conv_sampled("conv 64->160 @64^2 B20 sampled, persistent halo cfg 47 (one chunk per tile)", 47, 20, 64, 64, 64, 160, 6000);
    conv_sampled("conv 320->320 @64^2 B17 sampled, persistent halo cfg 47 (ragged tile count)", 47, 17, 64, 64, 320, 320, 6000);
    const int ab[] = {16, 40, 47}, ab128[] = {17, 41, 48};
    ab_case("conv 320->320 @64^2 B8", GEMM_CONV_S1, 8 * 64 * 64, 320, 320, 8, 64, 64, 0, ab, 3, 7);
    ab_case("conv 320->320 @64^2 B8 rowbias+silu", GEMM_CONV_S1, 8 * 64 * 64, 320, 320, 8, 64, 64, 0, ab, 3, 3, false, true);
    ab_case("conv 320->320 @64^2 B32 (DDIM)", GEMM_CONV_S1, 32 * 64 * 64, 320, 320, 32, 64, 64, 0, ab, 3, 3);
    ab_case("conv 640->640 @32^2 B32 (DDIM)", GEMM_CONV_S1, 32 * 32 * 32, 640, 640, 32, 32, 32, 0, ab, 3, 3);
    ab_case("conv 1280->1280 @16^2 B32 (DDIM)", GEMM_CONV_S1, 32 * 16 * 16, 1280, 1280, 32, 16, 16, 0, ab, 3, 3);
    ab_case("conv 960->320 @64^2 B32 (DDIM)", GEMM_CONV_S1, 32 * 64 * 64, 320, 960, 32, 64, 64, 0, ab, 3, 3);
    ab_case("conv 960->320 @64^2 B8", GEMM_CONV_S1, 8 * 64 * 64, 320, 960, 8, 64, 64, 0, ab, 3, 3);
    ab_case("conv 640->320 @64^2 B8 + residual", GEMM_CONV_S1, 8 * 64 * 64, 320, 640, 8, 64, 64, 0, ab, 3, 3, true);
    ab_case("conv 640->640 @32^2 B8", GEMM_CONV_S1, 8 * 32 * 32, 640, 640, 8, 32, 32, 0, ab, 3, 3);
    ab_case("conv 1280->1280 @16^2 B8", GEMM_CONV_S1, 8 * 16 * 16, 1280, 1280, 8, 16, 16, 0, ab, 3, 3);
    ab_case("conv 2560->1280 @16^2 B8", GEMM_CONV_S1, 8 * 16 * 16, 1280, 2560, 8, 16, 16, 0, ab, 3, 3);
    ab_case("conv 1280->1280 @8^2 B8 (split-K)", GEMM_CONV_S1, 8 * 8 * 8, 1280, 1280, 8, 8, 8, 0, ab, 3, 3);
    ab_case("conv 128->128 @256^2 B4 (VAE)", GEMM_CONV_S1, 4 * 256 * 256, 128, 128, 4, 256, 256, 0, ab128, 3, 3);
    ab_case("conv 256->256 @128^2 B4 (VAE)", GEMM_CONV_S1, 4 * 128 * 128, 256, 256, 4, 128, 128, 0, ab128, 3, 3);
    ab_case("gemm 4096^3", GEMM_LINEAR, 4096, 4096, 4096, 0, 0, 0, 0, ab128, 3, 3);
    ab_case("gemm 32768x320x1280 (FF out)", GEMM_LINEAR, 32768, 320, 1280, 0, 0, 0, 0, ab, 3, 3);
    ab_case("gemm 32768x320x1280+128 (FF out, LoRA)", GEMM_LINEAR, 32768, 320, 1280, 0, 0, 0, 128, ab, 3, 3, true);
    ab_case("gemm 8192x640x2560 (FF out 32^2)", GEMM_LINEAR, 8192, 640, 2560, 0, 0, 0, 0, ab, 3, 3);
    ab_case("gemm 131072x320x1280 (FF out, DDIM)", GEMM_LINEAR, 131072, 320, 1280, 0, 0, 0, 0, ab, 3, 3);
    ab_case("conv 128->128 @512^2 B2 (VAE)", GEMM_CONV_S1, 2 * 512 * 512, 128, 128, 2, 512, 512, 0, ab128, 3, 3);
    if (!quick) {
      ab_case("gemm 32768x320x320", GEMM_LINEAR, 32768, 320, 320, 0, 0, 0, 0, ab, 3, 3);
      ab_case("gemm 32768x2560x320", GEMM_LINEAR, 32768, 2560, 320, 0, 0, 0, 0, ab, 3, 3);
      ab_case("gemm 2048x1280x1280", GEMM_LINEAR, 2048, 1280, 1280, 0, 0, 0, 0, ab, 3, 3);
      ab_case("gemm 8192x640x640", GEMM_LINEAR, 8192, 640, 640, 0, 0, 0, 0, ab, 3, 3);
    }
    }
w4_ablations:
#ifdef W4_PROBE
    const int only40[] = {40};
    {
      printf("---- conv S1 under cfg 40: halo-resident image (1) vs per-tap DMA (0)\n");
      for (int rep = 0; rep < 2; ++rep)
        for (int hm = 0; hm < 2; ++hm) {
          cl::w4_halo_set(hm);
          char nm[64]; snprintf(nm, 64, "conv 320->320 @64^2 B8  halo %d", hm);
          ab_case(nm, GEMM_CONV_S1, 8 * 64 * 64, 320, 320, 8, 64, 64, 0, only40, 1, 5);
        }
      cl::w4_halo_set(1);
    }
    const int abls[] = {0, 1, 2, 3, 4, 5, 6, 7, 37, 8};
    const char* an[] = {"full", "no fragment reads", "no DMA", "no reads, no DMA", "no stores", "no reads + no stores", "no DMA + no stores",
                        "skeleton: MFMA + barrier only", "DMA + barriers only (no MFMA, no reads, no stores)",
                        "stores as one contiguous KB per instruction (hypothesis test: is the drain slow because of 64-byte row pieces?)"};
    for (int ai = 0; ai < 10; ++ai) {
      cl::w4_abl_set(abls[ai]);
      printf("---- ablation %d: %s (results wrong by construction; FAIL lines below are expected)\n", abls[ai], an[ai]);
      const int keep = g_fail;
      ab_case("conv 320->320 @64^2 B8", GEMM_CONV_S1, 8 * 64 * 64, 320, 320, 8, 64, 64, 0, only40, 1, 3);
      w4_timing_case(an[ai]);
      g_fail = keep;
    }
    cl::w4_abl_set(0);
    w4_timing_case("full");
#if defined(W4_PROBE) && defined(FL_TIMING)
    fl_timing_case16();
    w4_timing_case("full (again)");
    fl_timing_case16();
    // one tile per CU at the lower resolutions (256 tiles, no split-K)
    w4_timing_case("640->640 @32^2 B16 (90 stages)", 16, 32, 640, 640);
    fl_timing_case16(16, 32, 640, 640);
    w4_timing_case("1280->1280 @16^2 B32 (180 stages)", 32, 16, 1280, 1280);
    fl_timing_case16(32, 16, 1280, 1280);
    cl::w4_halo_set(0);
    w4_timing_case("640->640 @32^2 B16, per-tap DMA", 16, 32, 640, 640);
    cl::w4_halo_set(1);
#endif
#endif
    printf("probe_gemm --w4: %s (%d failures)\n", g_fail ? "FAILED" : "ALL PASS", g_fail);
    return g_fail ? 1 : 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--stagger")) {   // two workgroups per CU, the second one held back once
    for (int st : {0, 4000, 8000, 12000, 16000, 24000, 0}) {
      g_fl_persist_stagger = st;
      printf("---- stagger %d ticks\n", st);
      persist_case("131072x2560x320 GEGLU (DDIM)", 131072, 2560, 320, 0, ACT_GEGLU, 10, 29, false);
      persist_case("131072x320x320 (DDIM)", 131072, 320, 320, 0, 0, 10, 29, false);
      persist_case("131072x960x320 (DDIM qkv)", 131072, 960, 320, 0, 0, 10, 29, false);
      persist_case("131072x320x1280", 131072, 320, 1280, 0, 0, 10, 29, false);
      persist_case("32768x2560x320 (FF proj)", 32768, 2560, 320, 0, 0, 10, 29, false);
      persist_case("32768x960x320", 32768, 960, 320, 0, 0, 10, 29, false);
      persist_case("131072x128x320 (LoRA down)", 131072, 128, 320, 0, 0, 11, 30, false);
    }
    g_fl_persist_stagger = 0;
    printf("probe_gemm --stagger: %s (%d failures)\n", g_fail ? "FAILED" : "ALL PASS", g_fail);
    return g_fail ? 1 : 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--persist")) {
    const int pairs[][2] = {{16, 25}, {20, 27}, {10, 29}};
    for (auto& pr : pairs) {
      persist_case("32768x320x320 (to_q @64^2 B8)", 32768, 320, 320, 0, 0, pr[0], pr[1], true);
      persist_case("32768x320x320+r128", 32768, 320, 320, 128, 0, pr[0], pr[1], false);
      persist_case("32768x2560x320 (FF proj)", 32768, 2560, 320, 0, 0, pr[0], pr[1], true);
      persist_case("32768x2560x320 GEGLU", 32768, 2560, 320, 0, ACT_GEGLU, pr[0], pr[1], false);
      persist_case("131072x2560x320 GEGLU (DDIM)", 131072, 2560, 320, 0, ACT_GEGLU, pr[0], pr[1], true);
      persist_case("131072x320x320 (DDIM)", 131072, 320, 320, 0, 0, pr[0], pr[1], false);
      persist_case("131072x960x320 (DDIM qkv)", 131072, 960, 320, 0, 0, pr[0], pr[1], false);
      persist_case("32768x320x1280 (FF out)", 32768, 320, 1280, 0, 0, pr[0], pr[1], false);
      persist_case("8192x5120x640", 8192, 5120, 640, 0, 0, pr[0], pr[1], false);
      persist_case("8192x640x640", 8192, 640, 640, 0, 0, pr[0], pr[1], false);
      persist_case("2048x10240x1280", 2048, 10240, 1280, 0, 0, pr[0], pr[1], false);
      persist_case("ragged 40000x320x320", 40000, 320, 320, 0, 0, pr[0], pr[1], false);
      persist_case("ragged 33333x480x192+64", 33333, 480, 192, 64, 0, pr[0], pr[1], false);
    }
    printf("probe_gemm --persist: %s (%d failures)\n", g_fail ? "FAILED" : "ALL PASS", g_fail);
    return g_fail ? 1 : 0;
  }
  if (argc > 1 && !strcmp(argv[1], "--knobs")) {   // split-K tuning knobs on the shapes they affect
    for (int want : {256, 128, 256, 128}) {
      g_fl128_split_want = want;
      printf("---- g_fl128_split_want = %d\n", want);
      time_case("gemm bf16 2048x1280x1280", CL_BF16, GEMM_LINEAR, 2048, 1280, 1280, 0, 0, 0);
      time_case("gemm bf16 2048x1280x1280+r128", CL_BF16, GEMM_LINEAR, 2048, 1280, 1280, 0, 0, 0, 128);
      time_case("gemm bf16 512x1280x1280", CL_BF16, GEMM_LINEAR, 512, 1280, 1280, 0, 0, 0);
      time_case("gemm bf16 2048x128x1280", CL_BF16, GEMM_LINEAR, 2048, 128, 1280, 0, 0, 0);
      time_case("gemm bf16 8192x640x640", CL_BF16, GEMM_LINEAR, 8192, 640, 640, 0, 0, 0);
    }
    g_fl128_split_want = 256;
    for (int ms : {16, 8, 4, 16, 8, 4}) {
      g_tiny_m_minsub = ms;
      printf("---- g_tiny_m_minsub = %d\n", ms);
      time_case("gemm bf16 8x128x1280", CL_BF16, GEMM_LINEAR, 8, 128, 1280, 0, 0, 0);
      time_case("gemm bf16 8x1280x1280", CL_BF16, GEMM_LINEAR, 8, 1280, 1280, 0, 0, 0);
      time_case("gemm bf16 8x1280x1280+r128", CL_BF16, GEMM_LINEAR, 8, 1280, 1280, 0, 0, 0, 128);
      time_case("gemm bf16 8x640x1280", CL_BF16, GEMM_LINEAR, 8, 640, 1280, 0, 0, 0);
    }
    return 0;
  }
  wgrad_suite(argc > 1 && !strcmp(argv[1], "--time"));
  if (argc > 2 && !strcmp(argv[2], "--wgrad-only")) return g_fail ? 1 : 0;
  g_gemm_force_cfg = -1; correctness_suite("heuristic config");
  const int cfgs[] = {16, 17, 20, 21};
  for (int c : cfgs) { char tag[32]; snprintf(tag, 32, "forced cfg %d", c); g_gemm_force_cfg = c; correctness_suite(tag); }
  g_gemm_force_cfg = -1;

  if (argc > 1 && !strcmp(argv[1], "--time")) {
    const int tc[] = {-1, -1};
    for (int c : tc) {
      g_gemm_force_cfg = c;
      printf("---- timing, cfg %d (-1 heuristic, 6 = round-0 structure)\n", c);
      time_case("gemm bf16 4096^3", CL_BF16, GEMM_LINEAR, 4096, 4096, 4096, 0, 0, 0);
      time_case("gemm bf16 32768x320x320 (to_q @64^2 B8)", CL_BF16, GEMM_LINEAR, 32768, 320, 320, 0, 0, 0);
      time_case("gemm bf16 32768x320x320+r128 (LoRA)", CL_BF16, GEMM_LINEAR, 32768, 320, 320, 0, 0, 0, 128);
      time_case("gemm bf16 32768x128x320 (LoRA down)", CL_BF16, GEMM_LINEAR, 32768, 128, 320, 0, 0, 0);
      time_case("gemm bf16 32768x2560x320 (GEGLU proj)", CL_BF16, GEMM_LINEAR, 32768, 2560, 320, 0, 0, 0);
      time_case("gemm bf16 32768x320x1280 (FF out)", CL_BF16, GEMM_LINEAR, 32768, 320, 1280, 0, 0, 0);
      time_case("gemm bf16 8192x640x640", CL_BF16, GEMM_LINEAR, 8192, 640, 640, 0, 0, 0);
      time_case("gemm bf16 2048x1280x1280", CL_BF16, GEMM_LINEAR, 2048, 1280, 1280, 0, 0, 0);
      time_case("gemm bf16 2048x128x1280 (LoRA down 16^2)", CL_BF16, GEMM_LINEAR, 2048, 128, 1280, 0, 0, 0);
      time_case("gemm bf16 8192x128x640 (LoRA down 32^2)", CL_BF16, GEMM_LINEAR, 8192, 128, 640, 0, 0, 0);
      time_case("gemm bf16 512x1280x1280", CL_BF16, GEMM_LINEAR, 512, 1280, 1280, 0, 0, 0);
      time_case("gemm bf16 2048x3840x1280 (qkv 16^2)", CL_BF16, GEMM_LINEAR, 2048, 3840, 1280, 0, 0, 0);
      time_case("gemm bf16 8192x5120x640 (GEGLU proj 32^2)", CL_BF16, GEMM_LINEAR, 8192, 5120, 640, 0, 0, 0);
      time_case("gemm bf16 8192x640x2560 (FF out 32^2)", CL_BF16, GEMM_LINEAR, 8192, 640, 2560, 0, 0, 0);
      time_case("conv bf16 320->320 @64^2 B8", CL_BF16, GEMM_CONV_S1, 8 * 64 * 64, 320, 320, 8, 64, 64);
      time_case("conv bf16 960->320 @64^2 B8", CL_BF16, GEMM_CONV_S1, 8 * 64 * 64, 320, 960, 8, 64, 64);
      time_case("conv bf16 640->640 @32^2 B8", CL_BF16, GEMM_CONV_S1, 8 * 32 * 32, 640, 640, 8, 32, 32);
      time_case("conv bf16 1280->1280 @16^2 B8", CL_BF16, GEMM_CONV_S1, 8 * 16 * 16, 1280, 1280, 8, 16, 16);
      time_case("conv bf16 1280->1280 @8^2 B8", CL_BF16, GEMM_CONV_S1, 8 * 8 * 8, 1280, 1280, 8, 8, 8);
      time_case("conv bf16 2560->1280 @16^2 B8", CL_BF16, GEMM_CONV_S1, 8 * 16 * 16, 1280, 2560, 8, 16, 16);
      time_case("conv up2 bf16 640->640 32^2->64^2 B8", CL_BF16, GEMM_CONV_UP2, 8 * 64 * 64, 640, 640, 8, 32, 32);
      time_case("conv s2 bf16 320->320 64^2->32^2 B8", CL_BF16, GEMM_CONV_S2, 8 * 32 * 32, 320, 320, 8, 64, 64);
    }
    g_gemm_force_cfg = -1;
    time_case("gemm f32 4096x1280x1280", CL_F32, GEMM_LINEAR, 4096, 1280, 1280, 0, 0, 0);
  }
  printf("probe_gemm: %s (%d failures)\n", g_fail ? "FAILED" : "ALL PASS", g_fail);
  return g_fail ? 1 : 0;
}
