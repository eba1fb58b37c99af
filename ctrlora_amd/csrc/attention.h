// Argument blocks of the fused attention kernels (attention_fwd.hip / attention_bwd.hip).
#pragma once
#include "common.h"

namespace cl {

// Layouts (T = bf16 or float), heads h in [0,H), head dim DH, inner = H*DH:
//   Q  : [B*N,   ldq]  row b*N+n,   head h at columns [h*DH, (h+1)*DH)
//   K  : [B*Nkv, ldk]
//   V  : [B*Nkv, ldv]                       (backward only)
//   Vt : [B][inner][nkv_pad]  transposed V, zero padded to a multiple of 64 keys (forward)
//   O  : [B*N,   ldo]
//   LSE: [B][H][lse_stride] fp32, log2-domain log-sum-exp of scale*log2(e)*q.k (saved for
//        backward); lse_stride = N rounded up to a multiple of 64
struct AttnFwdArgs {
  const void* Q; long ldq;
  const void* K; long ldk;
  const void* Vt; int nkv_pad;
  void* O; long ldo;
  float* LSE; int lse_stride;
  int B, H, N, Nkv, DH;
  float scale;   // d_head^-0.5 (ldm/modules/attention.py:151)
  // 1: Q holds q * scale * log2(e) (the producing projection applied the factor in ITS fp32 epilogue: one rounding, like any
  // stored q) -- the kernels then take scores straight from the matrix product (log2 domain).  bf16 kernels only.
  int q_prescaled;
};

struct AttnBwdArgs {
  const void* Q; long ldq; const void* K; long ldk; const void* V; long ldv;
  const void* O; long ldo; const void* dO; long lddo;
  const void* Qt; const void* dOt; int n_pad;     // [B][inner][n_pad]   transposes over queries
  const void* Kt; int nkv_pad;                    // [B][inner][nkv_pad] transpose over keys
  const float* LSE; float* Delta; int lse_stride; // [B][H][lse_stride]
  void* dQ; long lddq; void* dK; long lddk; void* dV; long lddv;  // dK/dV may be null (frozen context)
  int B, H, N, Nkv, DH;
  float scale;
  int q_prescaled;   // as in AttnFwdArgs; dQ / dK / dV are the gradients of the TRUE q, k, v either way
  // Optional scratch of B * H * lse_stride * 32 bytes (16-byte aligned): with a pre-scaled Q and d_head 40 the dQ kernel
  // leaves (-lse, -delta) of every query row there, split into three bf16 pieces each, and the dK/dV kernel stages them as
  // the pad columns of its Q / dO tiles: the matrix products then deliver s - lse and dP - delta (see attention_tr.hip).
  void* row_ws;
};

int attn_fwd(const AttnFwdArgs& a, int dtype, hipStream_t st);
int attn_bwd(const AttnBwdArgs& a, int dtype, hipStream_t st);
// bf16, no materialised transposes (attention_tr.hip): Vt / Qt / dOt / Kt of the argument blocks are ignored
int attn_fwd_tr(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st);
int attn_bwd_tr(const AttnBwdArgs& a, hipStream_t st);
extern int g_attn_fuse_delta;   // 1 = the dQ kernel forms delta (default), 0 = separate attn_delta launch
extern int g_attn_variant_dkv4;
extern int g_attn_variant;   // probe hook: 0 = heuristic, 1 = tile-synchronous kernels only (no ping-pong schedule)
// pre-scaled-Q forward for d_head 40 (attention_fwd40.hip)
bool attn_fwd40_applies(const AttnFwdArgs& a);
int attn_fwd40(const AttnFwdArgs& a, const void* V, long ldv, hipStream_t st);
int attn_delta(const AttnBwdArgs& a, hipStream_t st);   // delta[q] = sum_d dO[q,d] O[q,d]  (bf16)

}  // namespace cl
