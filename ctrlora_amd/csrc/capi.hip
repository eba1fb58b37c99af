// extern "C" surface of libctrlora_hip.so (see include/ctrlora_hip.h for the contract).
#define CTRLORA_HIP_INTERNAL
#include "../../include/ctrlora_hip.h"
#include "attention.h"
#include "debug_hooks.h"
#include "elementwise.h"
#include "gemm.h"
#include "norm.h"

using namespace cl;

// Every entry point converts its stream argument through S(): it also clears a stale "last error" left by
// unrelated runtime calls of the host process (e.g. an event query that returned hipErrorNotReady), so that
// the hipGetLastError() check after our own launch reports OUR launch only.
static inline hipStream_t S(void* s) { (void)hipGetLastError(); return reinterpret_cast<hipStream_t>(s); }

static GemmParams base_params() {
  GemmParams g{};
  g.alpha = 1.f; g.beta = 0.f; g.splitk = 1; g.mode = GEMM_LINEAR;
  return g;
}

extern "C" {

int cl_abi_version(void) { return CL_ABI_VERSION; }
int cl_last_hip_error(void) { return g_last_hip_error; }
const char* cl_last_hip_error_string(void) { return hipGetErrorString((hipError_t)g_last_hip_error); }

int cl_set_workspace(void* ptr, long bytes) {
  if (bytes < 0 || (bytes > 0 && !ptr) || (reinterpret_cast<uintptr_t>(ptr) & 15)) return CL_EINVAL;
  gemm_set_workspace(ptr, bytes);
  return CL_OK;
}
int cl_set_stream_workspace(void* stream, void* ptr, long bytes) {
  if (bytes <= 0 || !ptr || (reinterpret_cast<uintptr_t>(ptr) & 15)) return CL_EINVAL;
  return gemm_set_stream_workspace(S(stream), ptr, bytes);
}
int cl_gemm_force_config(int cfg) { g_gemm_force_cfg = cfg; return CL_OK; }
int cl_gemm_force_splitk(int splitk) { g_gemm_force_splitk = splitk < 0 ? 0 : splitk; return CL_OK; }
int cl_gemm_tune_set(int dtype, int mode, int M, int N, int K1, int K2, int geglu, int cfg, int splitk) {
  return gemm_tune_set(dtype, mode, M, N, K1, K2, geglu, cfg, splitk);
}
int cl_gemm_tune_clear(void) { gemm_tune_clear(); return CL_OK; }
int cl_gemm_tune_size(void) { return gemm_tune_size(); }
// ---- probe hooks (csrc/debug_hooks.h; NOT part of include/ctrlora_hip.h: results are identical whatever they select)
int cl_debug_attention_variant(int v) {
  switch (v) {
    case 0: case 1: case 11: case 13: case 14:
      g_attn_variant = v; g_attn_variant_dkv4 = 0; return CL_OK;
    case 21:                               // = 0 with the four-fragment dK/dV probe kernel
      g_attn_variant = 0; g_attn_variant_dkv4 = 1; return CL_OK;
    default: return CL_EINVAL;
  }
}
int cl_debug_attention_fuse_delta(int on) { g_attn_fuse_delta = on ? 1 : 0; return CL_OK; }
int cl_debug_groupnorm_form(int three_pass, int one_pass) {
  g_gn_three_pass = three_pass ? 1 : 0; g_gn_one_pass = one_pass ? 1 : 0; return CL_OK;
}

int cl_debug_groupnorm_coop(int on) { g_gn_coop = on ? 1 : 0; return CL_OK; }
int cl_debug_groupnorm_coop_timeouts(void) { return (int)gnc_timeouts(); }

int cl_debug_wgrad_ring(int slots) { if (slots != 3 && slots != 4 && slots != 6) return CL_EINVAL; g_wgrad_ring = slots; return CL_OK; }
int cl_debug_gemm_xs_rules(int on) { g_gemm_xs_rules = on ? 1 : 0; return CL_OK; }
int cl_debug_gemm_tag(int on) { g_gemm_tag_on = on ? 1 : 0; return CL_OK; }
int cl_debug_gemm_tag_count(void) { return gemm_tag_count(); }
int cl_debug_gemm_tag_get(int i, long* out12) { return out12 ? gemm_tag_get(i, out12) : CL_EINVAL; }

int cl_gemm(const cl_gemm_params* p, int dtype, void* stream) {
  if (!p) return CL_EINVAL;
  GemmParams g{};
  g.A1 = p->A1; g.lda1 = p->lda1; g.K1 = p->K1; g.W1 = p->W1; g.ldw1 = p->ldw1;
  g.A2 = p->A2; g.lda2 = p->lda2; g.K2 = p->K2; g.W2 = p->W2; g.ldw2 = p->ldw2;
  g.M = p->M; g.N = p->N; g.mode = p->mode;
  g.B = p->B; g.Hin = p->Hin; g.Win = p->Win; g.Hout = p->Hout; g.Wout = p->Wout;
  g.zero_page = p->zero_page; g.bias = p->bias;
  g.rowbias = p->rowbias; g.ldrb = p->ldrb; g.rows_per_batch = p->rows_per_batch;
  g.residual = p->residual; g.ldr = p->ldr; g.alpha = p->alpha; g.beta = p->beta; g.act = p->act;
  g.C = p->C; g.ldc = p->ldc; g.out_f32 = p->out_f32; g.atomic = p->atomic; g.splitk = p->splitk < 1 ? 1 : p->splitk;
  g.a1_group_n = p->a1_group_n; g.a2_group_n = p->a2_group_n; g.alpha_n = p->alpha_n;
  g.ln_gamma = p->ln_gamma; g.ln_beta = p->ln_beta; g.ln_eps = p->ln_eps; g.ln_stats = p->ln_stats;
  return launch_gemm(g, dtype, S(stream));
}

int cl_lora_down(int dtype, const void* x, long ldx, const void* A, int r, void* t, long ldt, int M, int K,
                 void* stream) {
  GemmParams g = base_params();
  g.A1 = x; g.lda1 = ldx; g.K1 = K; g.W1 = A; g.ldw1 = K; g.M = M; g.N = r; g.C = t; g.ldc = ldt;
  return launch_gemm(g, dtype, S(stream));
}

int cl_lora_linear_fwd(int dtype, const void* x, long ldx, const void* W, const float* bias, const void* t,
                       long ldt, const void* Bup, int r, const void* residual, long ldr, int act, void* y,
                       long ldy, int M, int N, int K, void* stream) {
  GemmParams g = base_params();
  g.A1 = x; g.lda1 = ldx; g.K1 = K; g.W1 = W; g.ldw1 = K;
  if (r > 0) { g.A2 = t; g.lda2 = ldt; g.K2 = r; g.W2 = Bup; g.ldw2 = r; }
  g.M = M; g.N = N; g.bias = bias; g.act = act;
  if (residual) { g.residual = residual; g.ldr = ldr; g.beta = 1.f; }
  g.C = y; g.ldc = ldy;
  return launch_gemm(g, dtype, S(stream));
}

int cl_lora_linear_bwd_data(int dtype, const void* dy, long lddy, const void* Wt, const void* At, const void* Bt,
                            int r, void* u, long ldu, const void* accum, long ldacc, void* dx, long lddx, int M,
                            int N, int K, void* stream) {
  if (r > 0) {  // u = dy . B   ([M,N] x [N,r]) as an NT product against B^T [r,N]
    GemmParams g = base_params();
    g.A1 = dy; g.lda1 = lddy; g.K1 = N; g.W1 = Bt; g.ldw1 = N; g.M = M; g.N = r; g.C = u; g.ldc = ldu;
    const int rc = launch_gemm(g, dtype, S(stream));
    if (rc) return rc;
  }
  GemmParams g = base_params();
  g.A1 = dy; g.lda1 = lddy; g.K1 = N; g.W1 = Wt; g.ldw1 = N;
  if (r > 0) { g.A2 = u; g.lda2 = ldu; g.K2 = r; g.W2 = At; g.ldw2 = r; }
  g.M = M; g.N = K;
  if (accum) { g.residual = accum; g.ldr = ldacc; g.beta = 1.f; }
  g.C = dx; g.ldc = lddx;
  return launch_gemm(g, dtype, S(stream));
}

int cl_weight_grad(int dtype, const void* dyT, long lddyt, const void* xT, long ldxt, float* dW, long lddw, int N,
                   int K, int Mp, float scale, void* stream) {
  GemmParams g = base_params();
  g.A1 = dyT; g.lda1 = lddyt; g.K1 = Mp; g.W1 = xT; g.ldw1 = ldxt; g.M = N; g.N = K;
  g.alpha = scale; g.C = dW; g.ldc = lddw; g.out_f32 = 1; g.atomic = 1;
  // deep-K, small-MN product: split K so that a few hundred workgroups are in flight
  const int kpb = dtype == CL_BF16 ? 32 : 16;
  const long tiles = (long)((N + 63) / 64) * ((K + 63) / 64);
  int sk = (int)(512 / (tiles > 0 ? tiles : 1));
  const int ksteps = Mp / kpb;
  if (sk > ksteps / 4) sk = ksteps / 4;
  if (sk < 1) sk = 1;
  if (sk > 64) sk = 64;
  g.splitk = sk;
  return launch_gemm(g, dtype, S(stream));
}

int cl_weight_grad_tn(int dtype, const void* dy, long lddy, const void* x, long ldx, float* dW, long lddw, int M,
                      int N, int K, float scale, const void* zero_page, void* stream) {
  if (dtype != CL_BF16) return CL_EINVAL;   // fp32 parity mode uses cl_transpose + cl_weight_grad
  return launch_wgrad_tn(dy, lddy, x, ldx, dW, lddw, M, N, K, scale, zero_page, S(stream));
}

int cl_weight_grad_tn_group(int dtype, int n, const cl_wgrad_desc* descs, const void* zero_page, void* stream) {
  if (dtype != CL_BF16) return CL_EINVAL;
  if (n <= 0) return CL_OK;
  if (!descs || n > 4096) return CL_EINVAL;
  static_assert(sizeof(cl_wgrad_desc) == sizeof(WgradDesc), "descriptor layout");
  return launch_wgrad_tn_group(reinterpret_cast<const WgradDesc*>(descs), n, zero_page, S(stream));
}

int cl_conv3x3_fwd(int dtype, int mode, const void* x, long ldx, const void* Wp, const float* bias, const void* emb,
                   long ldemb, const void* residual, long ldr, void* y, long ldy, int B, int Hin, int Win, int Cin,
                   int Cout, const void* zero_page, void* stream) {
  GemmParams g = base_params();
  int Hout = Hin, Wout = Win;
  if (mode == GEMM_CONV_S2 || mode == GEMM_CONV_S2A) { Hout = Hin / 2; Wout = Win / 2; }
  else if (mode == GEMM_CONV_UP2 || mode == GEMM_CONV_T2) { Hout = 2 * Hin; Wout = 2 * Win; }
  else if (mode != GEMM_CONV_S1) return CL_EINVAL;
  g.mode = mode; g.A1 = x; g.lda1 = ldx; g.K1 = Cin; g.W1 = Wp; g.ldw1 = 9L * Cin;
  g.M = B * Hout * Wout; g.N = Cout; g.B = B; g.Hin = Hin; g.Win = Win; g.Hout = Hout; g.Wout = Wout;
  g.zero_page = zero_page; g.bias = bias;
  if (emb) { g.rowbias = emb; g.ldrb = ldemb; g.rows_per_batch = Hout * Wout; }
  if (residual) { g.residual = residual; g.ldr = ldr; g.beta = 1.f; }
  g.C = y; g.ldc = ldy;
  return launch_gemm(g, dtype, S(stream));
}

int cl_conv3x3_bwd_data(int dtype, int mode, const void* dy, long lddy, const void* Wd, const void* accum,
                        long ldacc, void* dx, long lddx, int B, int Hdy, int Wdy, int Cout, int Cin,
                        const void* zero_page, void* stream) {
  return cl_conv3x3_fwd(dtype, mode, dy, lddy, Wd, nullptr, nullptr, 0, accum, ldacc, dx, lddx, B, Hdy, Wdy, Cout,
                        Cin, zero_page, stream);
}

int cl_conv1x1_fwd(int dtype, const void* x, long ldx, const void* W, const float* bias, float scale,
                   const void* residual, long ldr, float beta, void* y, long ldy, int M, int Cin, int Cout,
                   void* stream) {
  GemmParams g = base_params();
  g.A1 = x; g.lda1 = ldx; g.K1 = Cin; g.W1 = W; g.ldw1 = Cin; g.M = M; g.N = Cout; g.bias = bias; g.alpha = scale;
  if (residual) { g.residual = residual; g.ldr = ldr; g.beta = beta; }
  g.C = y; g.ldc = ldy;
  return launch_gemm(g, dtype, S(stream));
}

long cl_groupnorm_ws_floats(int B, int HW, int C) { return gn_ws_floats(B, HW, C); }

int cl_groupnorm_silu_fwd(int dtype, const void* x, long ldx, void* y, long ldy, const float* gamma,
                          const float* beta, int B, int HW, int C, int groups, float eps, int silu, float* stats,
                          float* ws, void* stream) {
  GnArgs a{}; a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.gamma = gamma; a.beta = beta; a.B = B; a.HW = HW; a.C = C;
  a.G = groups; a.eps = eps; a.silu = silu; a.stats = stats; a.ws = ws;
  return gn_fwd(a, dtype, S(stream));
}

int cl_groupnorm_silu_bwd(int dtype, const void* x, long ldx, const void* dy, long lddy, const void* accum,
                          long ldacc, void* dx, long lddx, const float* gamma, const float* beta, const float* stats,
                          int B, int HW, int C, int groups, int silu, float* dgamma, float* dbeta, float* ws,
                          void* stream) {
  GnBwdArgs a{}; a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.accum = accum; a.ldacc = ldacc; a.dx = dx; a.lddx = lddx;
  a.gamma = gamma; a.beta = beta; a.stats = stats; a.B = B; a.HW = HW; a.C = C; a.G = groups; a.silu = silu;
  a.dgamma = dgamma; a.dbeta = dbeta; a.ws = ws;
  return gn_bwd(a, dtype, S(stream));
}

int cl_layernorm_fwd(int dtype, const void* x, long ldx, void* y, long ldy, const float* gamma, const float* beta,
                     int M, int D, float eps, float* stats, void* stream) {
  LnArgs a{}; a.x = x; a.ldx = ldx; a.y = y; a.ldy = ldy; a.gamma = gamma; a.beta = beta; a.M = M; a.D = D; a.eps = eps;
  a.stats = stats;
  return ln_fwd(a, dtype, S(stream));
}

int cl_layernorm_bwd(int dtype, const void* x, long ldx, const void* dy, long lddy, const void* accum, long ldacc,
                     void* dx, long lddx, const float* gamma, const float* stats, int M, int D, float* dgamma,
                     float* dbeta, void* stream) {
  LnBwdArgs a{}; a.x = x; a.ldx = ldx; a.dy = dy; a.lddy = lddy; a.accum = accum; a.ldacc = ldacc; a.dx = dx; a.lddx = lddx;
  a.gamma = gamma; a.stats = stats; a.M = M; a.D = D; a.dgamma = dgamma; a.dbeta = dbeta;
  return ln_bwd(a, dtype, S(stream));
}

int cl_attention_fwd(int dtype, const void* Q, long ldq, const void* K, long ldk, const void* Vt, int nkv_pad,
                     void* O, long ldo, float* LSE, int lse_stride, int B, int H, int N, int Nkv, int dh, float scale,
                     void* stream) {
  AttnFwdArgs a{}; a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.Vt = Vt; a.nkv_pad = nkv_pad; a.O = O; a.ldo = ldo;
  a.LSE = LSE; a.lse_stride = lse_stride; a.B = B; a.H = H; a.N = N; a.Nkv = Nkv; a.DH = dh; a.scale = scale;
  return attn_fwd(a, dtype, S(stream));
}

int cl_attention_bwd(int dtype, const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                     const void* O, long ldo, const void* dO, long lddo, const void* Qt, const void* dOt, int n_pad,
                     const void* Kt, int nkv_pad, const float* LSE, float* Delta, int lse_stride, void* dQ, long lddq,
                     void* dK, long lddk, void* dV, long lddv, int B, int H, int N, int Nkv, int dh, float scale,
                     void* stream) {
  AttnBwdArgs a{}; a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo;
  a.dO = dO; a.lddo = lddo; a.Qt = Qt; a.dOt = dOt; a.n_pad = n_pad; a.Kt = Kt; a.nkv_pad = nkv_pad; a.LSE = LSE;
  a.Delta = Delta; a.lse_stride = lse_stride; a.dQ = dQ; a.lddq = lddq; a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv;
  a.B = B; a.H = H; a.N = N; a.Nkv = Nkv; a.DH = dh; a.scale = scale;
  return attn_bwd(a, dtype, S(stream));
}

int cl_attention_fwd_v2(int dtype, const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv, void* O,
                        long ldo, float* LSE, int lse_stride, int B, int H, int N, int Nkv, int dh, float scale,
                        int flags, void* stream) {
  if (dtype != CL_BF16 || (flags & ~CL_ATTN_Q_PRESCALED)) return CL_EINVAL;
  AttnFwdArgs a{}; a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.O = O; a.ldo = ldo;
  a.LSE = LSE; a.lse_stride = lse_stride; a.B = B; a.H = H; a.N = N; a.Nkv = Nkv; a.DH = dh; a.scale = scale;
  a.q_prescaled = (flags & CL_ATTN_Q_PRESCALED) ? 1 : 0;
  return attn_fwd_tr(a, V, ldv, S(stream));
}

int cl_attention_bwd_v2(int dtype, const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                        const void* O, long ldo, const void* dO, long lddo, const float* LSE, float* Delta,
                        int lse_stride, void* dQ, long lddq, void* dK, long lddk, void* dV, long lddv, int B, int H,
                        int N, int Nkv, int dh, float scale, int flags, void* row_ws, void* stream) {
  if (dtype != CL_BF16 || (flags & ~CL_ATTN_Q_PRESCALED)) return CL_EINVAL;
  AttnBwdArgs a{}; a.Q = Q; a.ldq = ldq; a.K = K; a.ldk = ldk; a.V = V; a.ldv = ldv; a.O = O; a.ldo = ldo;
  a.dO = dO; a.lddo = lddo; a.LSE = LSE; a.Delta = Delta; a.lse_stride = lse_stride; a.dQ = dQ; a.lddq = lddq;
  a.dK = dK; a.lddk = lddk; a.dV = dV; a.lddv = lddv; a.B = B; a.H = H; a.N = N; a.Nkv = Nkv; a.DH = dh; a.scale = scale;
  a.q_prescaled = (flags & CL_ATTN_Q_PRESCALED) ? 1 : 0;
  a.row_ws = row_ws;
  return attn_bwd_tr(a, S(stream));
}

int cl_geglu_fwd(int dtype, const void* h, long ldh, void* out, long ldo, long M, int F, void* stream) { return geglu_fwd(dtype, h, ldh, out, ldo, M, F, S(stream)); }
int cl_geglu_bwd(int dtype, const void* h, long ldh, const void* dout, long lddo, void* dh, long lddh, long M, int F, void* stream) { return geglu_bwd(dtype, h, ldh, dout, lddo, dh, lddh, M, F, S(stream)); }
int cl_silu_fwd(int dtype, const void* x, void* y, long n, void* stream) { return silu_fwd(dtype, x, y, n, S(stream)); }
int cl_silu_bwd(int dtype, const void* x, const void* dy, void* dx, long n, void* stream) { return silu_bwd(dtype, x, dy, dx, n, S(stream)); }
int cl_axpby(int dtype, const void* x, long ldx, void* y, long ldy, long M, int C, float a, float b, void* stream) { return axpby(dtype, x, ldx, y, ldy, M, C, a, b, S(stream)); }
int cl_transpose(int in_dtype, int out_dtype, const void* in, long ldi, long bsi, void* out, long ldo, long bso, int Bt, int R, int C, int Rpad, void* stream) { return transpose(in_dtype, out_dtype, in, ldi, bsi, out, ldo, bso, Bt, R, C, Rpad, S(stream)); }
int cl_nchw_to_tok(int dtype, const float* in, void* out, long ldo, int B, int Cin, int Cpad, int HW, void* stream) { return nchw_to_tok(dtype, in, out, ldo, B, Cin, Cpad, HW, S(stream)); }
int cl_tok_to_nchw(int dtype, const void* in, long ldi, float* out, int B, int C, int HW, float alpha, float beta, void* stream) { return tok_to_nchw(dtype, in, ldi, out, B, C, HW, alpha, beta, S(stream)); }
int cl_colsum(int dtype, const void* in, long ldi, float* out, long ldo, int B, int HW, int C, float scale, void* stream) { return colsum(dtype, in, ldi, out, ldo, B, HW, C, scale, S(stream)); }
int cl_pool2x2(int dtype, const void* in, long ldi, void* out, long ldo, int B, int H, int W, int C, int accumulate, void* stream) { return pool2x2(dtype, in, ldi, out, ldo, B, H, W, C, accumulate, S(stream)); }
int cl_pack2d(int dtype, const float* in, long ldi, void* out, long ldo, long R, int C, int Cpad, void* stream) { return pack2d(dtype, in, ldi, out, ldo, R, C, Cpad, S(stream)); }
int cl_repack(int dtype, const float* flat, const long* desc, const int* tile_prefix, int ndesc, int total_tiles, void* stream) { return repack(dtype, flat, desc, tile_prefix, ndesc, total_tiles, S(stream)); }
int cl_timestep_embedding(int dtype, const long* t, const float* freqs, void* out, long ldo, int B, int half, void* stream) { return timestep_embed(dtype, t, freqs, out, ldo, B, half, S(stream)); }
int cl_qsample(const float* z, const float* noise, const long* t, const float* sqrt_ac, const float* sqrt_1mac, float* out, int B, long per_sample, void* stream) { return qsample(z, noise, t, sqrt_ac, sqrt_1mac, out, B, per_sample, S(stream)); }
int cl_mse_loss(const float* eps, const float* target, float* d_eps, float* loss, long n, float gscale, void* stream) { return mse_loss(eps, target, d_eps, loss, n, gscale, S(stream)); }
int cl_ddim_step(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coef, int index, float scale, float* x_prev, float* pred_x0, long n, void* stream) { return ddim_step(x, e_c, e_u, noise, coef, index, scale, x_prev, pred_x0, n, S(stream)); }
int cl_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, void* stream) { return adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, S(stream)); }

int cl_p_losses_mse(const float* eps, const float* target, float* d_eps, const long* t, const float* lvlb, float* out, float* per_sample, float* scratch, int B, long per_sample_elems, float gscale, float w_simple, float w_elbo, void* stream) { return plosses_mse(eps, target, d_eps, t, lvlb, out, per_sample, scratch, B, per_sample_elems, gscale, w_simple, w_elbo, S(stream)); }
int cl_conv_tap_gather(int dtype, const void* x, long ldx, void* out, long ldo, int B, int Hin, int Win, int Hout, int Wout, int C, int tap, int stride, int pad, void* stream) { return conv_tap_gather(dtype, x, ldx, out, ldo, B, Hin, Win, Hout, Wout, C, tap, stride, pad, S(stream)); }
int cl_softmax_rows(int dtype, const float* Sm, long lds_, void* P, long ldp, long M, int N, float scale, void* stream) { return softmax_rows(dtype, Sm, lds_, P, ldp, M, N, scale, S(stream)); }
int cl_zero(void* p, long nbytes, void* stream) { return zero_bytes(p, nbytes, S(stream)); }
int cl_tick(int* counter, void* stream) { return tick(counter, S(stream)); }
int cl_adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, int* step, void* stream) { return adamw_dev(p, g, m, v, n, hyper, step, S(stream)); }
int cl_ddim_set_t(const long* table, const int* cursor, int S_, long* ts, int n, void* stream) { return ddim_set_t(table, cursor, S_, ts, n, S(stream)); }
int cl_ddim_step_dev(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coef, const int* cursor, int S_, float scale, float* x_prev, float* pred_x0, long n, void* stream) { return ddim_step_dev(x, e_c, e_u, noise, coef, cursor, S_, scale, x_prev, pred_x0, n, S(stream)); }

}  // extern "C"
