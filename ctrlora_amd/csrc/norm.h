// Argument blocks of the normalisation kernels (norm.hip).
#pragma once
#include "common.h"

namespace cl {

extern int g_gn_three_pass;   // A/B hook: 1 = partial -> finalize -> apply for every GroupNorm
extern int g_gn_one_pass;     // A/B hook: 0 = never the one-launch register-resident form

struct GnArgs {
  const void* x; long ldx;      // [B*HW, C] token-major (NHWC), row stride ldx
  void* y; long ldy;
  const float* gamma; const float* beta;
  int B, HW, C, G; float eps; int silu;
  float* stats;                 // out [B][G][2] = mean, rstd (kept for backward)
  float* ws;                    // gn_ws_floats(B,HW,C) floats of scratch
};

struct GnBwdArgs {
  const void* x; long ldx; const void* dy; long lddy;
  const void* accum; long ldacc;  // optional: dx = accum + grad (gradient fan-in fused)
  void* dx; long lddx;
  const float* gamma; const float* beta; const float* stats;
  int B, HW, C, G; int silu;
  float* dgamma; float* dbeta;    // optional fp32 accumulators (trainable norms only)
  float* ws;
};

struct LnArgs {
  const void* x; long ldx; void* y; long ldy;
  const float* gamma; const float* beta; int M, D; float eps;
  float* stats;                 // out [M][2] = mean, rstd (may be null at inference)
};

struct LnBwdArgs {
  const void* x; long ldx; const void* dy; long lddy;
  const void* accum; long ldacc; void* dx; long lddx;
  const float* gamma; const float* stats; int M, D;
  float* dgamma; float* dbeta;
};

long gn_ws_floats(int B, int HW, int C);
// one-launch, one-pass cooperative form (norm_coop.hip); CL_EINVAL = not its case
extern int g_gn_coop;
int gnc_fwd(const GnArgs& a, int dtype, hipStream_t st);
int gnc_bwd(const GnBwdArgs& a, int dtype, hipStream_t st);
unsigned gnc_timeouts();
int gn_fwd(const GnArgs& a, int dtype, hipStream_t st);
int gn_bwd(const GnBwdArgs& a, int dtype, hipStream_t st);
int ln_fwd(const LnArgs& a, int dtype, hipStream_t st);
int ln_bwd(const LnBwdArgs& a, int dtype, hipStream_t st);

}  // namespace cl
