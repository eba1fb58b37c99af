// Host launchers of the elementwise / layout kernels (elementwise.hip).
#pragma once
#include "common.h"

namespace cl {

int geglu_fwd(int dtype, const void* h, long ldh, void* out, long ldo, long M, int F, hipStream_t st);
int geglu_bwd(int dtype, const void* h, long ldh, const void* dout, long lddo, void* dh, long lddh, long M, int F, hipStream_t st);
int silu_fwd(int dtype, const void* x, void* y, long n, hipStream_t st);
int silu_bwd(int dtype, const void* x, const void* dy, void* dx, long n, hipStream_t st);
int axpby(int dtype, const void* x, long ldx, void* y, long ldy, long M, int C, float a, float b, hipStream_t st);
int transpose(int in_dtype, int out_dtype, const void* in, long ldi, long bsi, void* out, long ldo, long bso,
              int Bt, int R, int C, int Rpad, hipStream_t st);
int nchw_to_tok(int dtype, const float* in, void* out, long ldo, int B, int Cin, int Cpad, int HW, hipStream_t st);
int tok_to_nchw(int dtype, const void* in, long ldi, float* out, int B, int C, int HW, float alpha, float beta, hipStream_t st);
int timestep_embed(int dtype, const long* t, const float* freqs, void* out, long ldo, int B, int half, hipStream_t st);
int qsample(const float* z, const float* noise, const long* t, const float* sqrt_ac, const float* sqrt_1mac,
            float* out, int B, long per, hipStream_t st);
int mse_loss(const float* eps, const float* target, float* d_eps, float* loss, long n, float gscale, hipStream_t st);
int plosses_mse(const float* eps, const float* target, float* d_eps, const long* t, const float* lvlb, float* out,
                float* per_sample, float* scratch, int B, long per, float gscale, float w_simple, float w_elbo,
                hipStream_t st);
int zero_bytes(void* p, long nbytes, hipStream_t st);
int conv_tap_gather(int dtype, const void* x, long ldx, void* out, long ldo, int B, int Hin, int Win, int Hout, int Wout,
                    int C, int tap, int stride, int pad, hipStream_t st);
int softmax_rows(int dtype, const float* S, long lds_, void* P, long ldp, long M, int N, float scale, hipStream_t st);
int ddim_step(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coef, int index,
              float scale, float* x_prev, float* pred_x0, long n, hipStream_t st);
int tick(int* counter, hipStream_t st);
int adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, int* step, hipStream_t st);
int ddim_set_t(const long* table, const int* cursor, int S, long* ts, int n, hipStream_t st);
int ddim_step_dev(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coef,
                  const int* cursor, int S, float scale, float* x_prev, float* pred_x0, long n, hipStream_t st);
int adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
          float wd, int step, float gscale, hipStream_t st);
int pool2x2(int dtype, const void* in, long ldi, void* out, long ldo, int B, int H, int W, int C, int accumulate, hipStream_t st);
int colsum(int dtype, const void* in, long ldi, float* out, long ldo, int B, int HW, int C, float scale, hipStream_t st);
int repack(int dtype, const float* flat, const long* desc, const int* tile_prefix, int ndesc, int total_tiles,
           hipStream_t st);
int pack2d(int dtype, const float* in, long ldi, void* out, long ldo, long R, int C, int Cpad, hipStream_t st);

}  // namespace cl
