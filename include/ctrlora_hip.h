/* ctrlora_hip.h -- C ABI of libctrlora_hip.so: the MI355X (gfx950) kernels behind the
 * CtrLoRA hot path (SD1.5 UNet + ControlNet + condition-LoRA forward/backward, DDIM step).
 *
 * The reference (xyfJASON/ctrlora) has no FFI: the hot path sits behind Python class
 * contracts (cldm.*), and every operator below replaces the ATen kernel(s) that the cited
 * reference lines issue.  The Python host side (ctrlora_amd/, cldm/, ldm/) binds these
 * entry points with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, CL_EINVAL (1) for an unsupported shape /
 *     alignment, CL_ELAUNCH (2) for a HIP launch failure.  Nothing is allocated, nothing
 *     synchronises; all work is enqueued on `stream` (a hipStream_t) and is hipGraph-capture safe.
 *   - dtype: 0 = bf16 storage (fp32 accumulate, fp32 statistics/softmax),
 *            1 = fp32 storage ("parity mode": f32-input MFMA, exact fmaf chains).
 *   - activations are token-major ("NHWC"): a [rows, C] matrix with an explicit row stride
 *     (ld, in elements) so that operators can read/write column slices of wider buffers
 *     (decoder concat buffers, fused QKV).  rows = b*H*W + y*W + x.
 *   - channel counts / K dimensions must be multiples of 32 (bf16) or 16 (fp32); N and all
 *     ld's multiples of 8.  Base pointers 16-byte aligned.
 */
#ifndef CTRLORA_HIP_H
#define CTRLORA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef CTRLORA_HIP_INTERNAL  /* the library's own sources carry these as C++ enums */
#define CL_OK 0
#define CL_EINVAL 1
#define CL_ELAUNCH 2
#define CL_BF16 0
#define CL_F32 1
#endif

/* the version this header describes; cl_abi_version() returns the one the loaded library was built from */
#define CL_ABI_VERSION 7
int cl_abi_version(void);
/* the hipError_t behind the most recent CL_ELAUNCH return (diagnostics) */
int cl_last_hip_error(void);

/* Scratch for the deterministic split-K path of the contraction kernels (fp32 partial slabs for
 * the deep-K / small-MN products of the 8x8 and 16x16 UNet levels).  The library never allocates:
 * the host (torch) owns the buffer and registers it once per process; without one, split-K is off.
 * 64 MiB covers every CtrLoRA shape at batch 16.  Used stream-ordered on the launching stream.
 * The same scratch carries the per-workgroup column-sum partials of cl_colsum and of cl_layernorm_bwd's
 * dgamma / dbeta (partials + a finishing launch instead of same-cache-line atomics); without a registered
 * buffer those two fall back to fp32 atomics.  (NULL, 0) unregisters. */
int cl_set_workspace(void* device_ptr, long bytes);
/* A second (third, fourth) scratch region bound to one stream: contractions launched on that stream use it
 * instead of the default, so concurrent streams never share split-K slabs. */
int cl_set_stream_workspace(void* stream, void* device_ptr, long bytes);
/* tuning hook: force a tile configuration of csrc/gemm.hip (-1 = built-in heuristic) */
int cl_gemm_force_config(int cfg);
/* tuning hook: impose the split-K factor of the workspace path on every following contraction (0 = built-in rule) */
int cl_gemm_force_splitk(int splitk);
/* Measured launch table: the contraction with this signature (dtype CL_BF16/CL_F32, cl_gemm_mode, M, N, K1, K2,
 * geglu = GEGLU epilogue (act 2)) is launched with tile configuration `cfg` and split-K factor `splitk` (0 = built-in rule
 * for the factor) instead of the built-in choice.  The host loads ctrlora_amd/gemm_tuned_gfx950.json (written by
 * tools/gemm_autotune.py from timings on an MI355X) through this entry at start-up; signatures not in the table
 * keep the built-in rules, and every configuration re-checks its own preconditions at launch. */
int cl_gemm_tune_set(int dtype, int mode, int M, int N, int K1, int K2, int geglu, int cfg, int splitk);
int cl_gemm_tune_clear(void);
int cl_gemm_tune_size(void);
/* (Schedule / launch-form A-B switches that do not change results live in ctrlora_amd/csrc/debug_hooks.h, outside this
 * boundary.) */

/* ---- dense contractions -------------------------------------------------------------
 * One MFMA kernel family (csrc/gemm.hip) behind all of them:
 *   out[M,N] = act( A1.W1^T + A2.W2^T + bias[n] + rowbias[m / rows_per_batch, n] ) * alpha
 *              + beta * residual[m, n]
 */
enum cl_gemm_mode {
  CL_GEMM_LINEAR = 0,   /* A1 row-major [M,K1]                                              */
  CL_GEMM_CONV_S1 = 1,  /* A1 = NHWC [B,Hin,Win,K1]; 3x3 stride 1 pad 1                      */
  CL_GEMM_CONV_S2 = 2,  /* 3x3 stride 2 pad 1                                                */
  CL_GEMM_CONV_UP2 = 3, /* 3x3 over nearest-x2 upsampled input                               */
  CL_GEMM_CONV_T2 = 4,  /* 3x3 over zero-stuffed x2 grid: data-gradient of CL_GEMM_CONV_S2   */
  CL_GEMM_CONV_S2A = 5, /* 3x3 stride 2, pad (0,1,0,1): AutoencoderKL's Downsample
                           (ldm/modules/diffusionmodules/model.py:80-84)                        */
  /* Phase-decomposed forms of UP2 / T2 with the same results (Upsample.forward, openaimodel.py:108-118; the data gradient of
   * Downsample's stride-2 conv, :150): output pixel (2y + a, 2x + b) depends on a 2 x 2 (UP2) / (1 + a) x (1 + b) (T2) window
   * of source pixels only, so the four phases (a, b) are stride-1 window products 4 K1 / on average 2.25 K1 deep instead
   * of 9 K1.  A1 = source NHWC [B,Hin,Win,K1]; M = 4 B Hin Win (B Hin Win a multiple of 128), Hout = 2 Hin, Wout = 2 Win;
   * C / residual / rowbias rows are OUTPUT pixels (the kernel interleaves the phases); W1 = phase-packed weights
   * (ctrlora_amd/engine/packing.py: Conv3W.phase_weights), ldw1 ignored.  bf16 / fp32, K1 whole 128-byte lines, no K2.   */
  CL_GEMM_CONV_UP2P = 6,
  CL_GEMM_CONV_T2P = 7,
  /* 4x4 window, stride 2, pad 1: the data gradient of CL_GEMM_CONV_UP2 formed on the source grid (each source pixel receives
   * from upsampled rows / columns 2y - 1 .. 2y + 2; coincident 3x3 taps summed) -- 16 K1 deep at M / 4 rows instead of a
   * stride-1 data gradient on the upsampled grid plus a 2x2 sum pool.  A1 = dy NHWC [B,Hin,Win,K1] (upsampled grid),
   * Hout = Hin / 2, Wout = Win / 2, M = B Hout Wout, W1 = [N][4][4][K1], ldw1 = 16 K1 (Conv3W.phase_weights("up2d")).       */
  CL_GEMM_CONV_S2K4 = 8
};

typedef struct cl_gemm_params {
  const void* A1; long lda1; int K1;
  const void* W1; long ldw1;          /* [N, taps*K1], K contiguous (taps = 9 in conv modes: ky,kx,c) */
  const void* A2; long lda2; int K2;  /* optional second K segment                           */
  const void* W2; long ldw2;
  int M, N;
  int mode;
  int B, Hin, Win, Hout, Wout;        /* conv geometry                                       */
  const void* zero_page;              /* zeros on the device for conv halos: >= 4*K1 + 128 bytes (16 KiB is enough for every
                                         CtrLoRA shape); the stride-1 conv path walks it like an image row     */
  const float* bias;                  /* fp32 [N] or NULL                                    */
  const void* rowbias; long ldrb; int rows_per_batch;
  const void* residual; long ldr;
  float alpha, beta;
  int act;                            /* 0 none, 1 SiLU, 2 GEGLU (value / gate rows of W interleaved per 160-row tile, C is
                                         [M, N/2]: ldm/modules/attention.py:49-56 fused into the projection; no-grad forwards),
                                         3 (ABI 7) the same fusion with W's rows in their natural [value | gate] order:
                                         bf16, K1 in {320, 640}, K2 in {0, 128}, N % 64 == 0 only -- CL_EINVAL otherwise   */
  void* C; long ldc;
  int out_f32;                        /* store fp32 regardless of dtype                      */
  int atomic;                         /* fp32 atomicAdd into C (gradient accumulation)       */
  int splitk;                         /* K splits in atomic mode; otherwise chosen internally */
  /* Grouped K segments (linear mode; ABI 5): G LoRA linears that share their input (to_q | to_k | to_v of one
   * CrossAttention, cldm/lora.py:285-291 x3) as ONE product with the outputs side by side.  a2_group_n = width of one
   * linear's output: columns [g*a2_group_n, (g+1)*a2_group_n) read their second segment from columns [g*K2, (g+1)*K2)
   * of A2 (A2 = [x Aq^T | x Ak^T | x Av^T], W2 = [Bq; Bk; Bv]).  a1_group_n: the same for the FIRST segment
   * (u = [dq Bq | dk Bk | dv Bv] from dy = [dq | dk | dv]: A1 columns [g*K1, (g+1)*K1)).  0 = ungrouped. */
  int a1_group_n, a2_group_n;
  /* ABI 6: alpha multiplies output columns [0, alpha_n) only (0 = every column; a multiple of 8).  The q | k | v
   * projection of a CrossAttention writes q * (d_head^-0.5 * log2 e) and plain k, v in ONE launch: the pre-scaled-Q
   * contract of cl_attention_*_v2 (CL_ATTN_Q_PRESCALED). */
  int alpha_n;
  /* ABI 7: LayerNorm as a prologue of the product (ldm/modules/attention.py:271-275: x + attn(norm(x)) -- the norm never
   * exists in memory).  ln_gamma != NULL: A1 holds the UN-normalised rows and the kernel forms (row - mean) * rstd * gamma
   * + beta over the K1 columns before the contraction (fp32 statistics; rounded to `dtype` like cl_layernorm_fwd's output).
   * ln_stats (may be NULL): [M][2] fp32 {mean, rstd} for cl_layernorm_bwd.  bf16, linear mode, K1 in {320, 640}, K2 = 0,
   * no rowbias / residual only -- CL_EINVAL otherwise (the caller then runs cl_layernorm_fwd itself). */
  const float* ln_gamma; const float* ln_beta; float ln_eps; float* ln_stats;
} cl_gemm_params;

/* Generic entry; the named operators below are thin fillers of cl_gemm_params. */
int cl_gemm(const cl_gemm_params* p, int dtype, void* stream);

/* LoRACompatibleLinear.forward (cldm/lora.py:285-291): y = x W^T + b + (x A^T) B^T (+ residual).
 * t = x A^T (LoRALinearLayer.down, cldm/lora.py:74) is produced by cl_lora_down and kept for
 * the backward pass; the up-projection is folded into the main MFMA chain as a second K segment.
 * W [N,K], A [r,K], Bup [N,r] row-major in `dtype`; pass t = NULL / r = 0 for a plain nn.Linear. */
int cl_lora_down(int dtype, const void* x, long ldx, const void* A, int r, void* t, long ldt,
                 int M, int K, void* stream);
int cl_lora_linear_fwd(int dtype, const void* x, long ldx, const void* W, const float* bias,
                       const void* t, long ldt, const void* Bup, int r,
                       const void* residual, long ldr, int act, void* y, long ldy,
                       int M, int N, int K, void* stream);
/* data gradient: dx = dy W + (dy B) A (+ accum).  Wt = W^T [K,N], At = A^T [K,r], Bt = B^T [r,N];
 * u = dy B [M,r] is written to `u` (needed again for dA).  No dW is ever formed for frozen W. */
int cl_lora_linear_bwd_data(int dtype, const void* dy, long lddy, const void* Wt, const void* At,
                            const void* Bt, int r, void* u, long ldu, const void* accum, long ldacc,
                            void* dx, long lddx, int M, int N, int K, void* stream);
/* weight gradient of a (LoRA / zero-conv) matrix: dW[N,K] += dyT[N,Mp] . xT[K,Mp]^T, fp32 atomics.
 * dyT / xT are the zero-padded transposes produced by cl_transpose (Mp multiple of 32). */
int cl_weight_grad(int dtype, const void* dyT, long lddyt, const void* xT, long ldxt, float* dW,
                   long lddw, int N, int K, int Mp, float scale, void* stream);

/* the same weight gradient without materialised transposes (bf16 only): dW[N,K] += scale * dy[M,N]^T . x[M,K],
 * both operands row-major as the forward/backward pass left them; fragments are built with the gfx950 LDS
 * transpose read (csrc/wgrad.hip).  zero_page: >= 64 zero bytes on the device (rows past M; >= 64 bytes). */
int cl_weight_grad_tn(int dtype, const void* dy, long lddy, const void* x, long ldx, float* dW, long lddw,
                      int M, int N, int K, float scale, const void* zero_page, void* stream);

/* grouped form: n independent weight gradients in ONE launch (+ one slab-reduce launch).  The LoRA
 * matrices are small, so a single problem cannot fill the chip; the engine queues the problems of a
 * transformer block and flushes them together.  `descs` is a HOST array (copied into the kernel arguments,
 * hipGraph-replayable). */
typedef struct cl_wgrad_desc {
  const void* dy; long lddy;      /* [M, N] row-major bf16 */
  const void* x; long ldx;        /* [M, K] row-major bf16 */
  float* dW; long lddw;           /* [N, K] fp32, accumulated */
  int M, N, K; float scale;
  /* tap >= 0: ONE TAP of a 3x3 convolution's weight gradient (Base-ControlNet pre-training trains the conv weights,
   * cldm/cldm_ctrlora_pretrain.py:174-182).  x is then the NHWC input [B*Hin*Win, K]; row m = (b, oy, ox) of dy pairs
   * with input pixel (oy*stride + tap/3 - pad, ox*stride + tap%3 - pad), zero outside the image; dW points at the tap's
   * [N, K] slice of a [N][3][3][K] gradient (lddw = 9 K).  tap = -1: plain dy^T x (the other fields are ignored).
   * tap = 16 + ky (stride 1, pad 1, Wout a multiple of 32 or a divisor of 32, M a multiple of 32): the THREE taps (ky, 0..2) of a
   * kernel row from one problem -- dW points at tap (ky, 0), taps kx = 1, 2 lie K and 2 K floats further on in each row. */
  int tap, Hin, Win, Hout, Wout, stride, pad, reserved;
} cl_wgrad_desc;
int cl_weight_grad_tn_group(int dtype, int n, const cl_wgrad_desc* descs, const void* zero_page, void* stream);

/* 3x3 convolutions of ResBlock / Downsample / Upsample / input conv / out conv
 * (ldm/modules/diffusionmodules/openaimodel.py:108-118,150,203,229,729; cldm/cldm.py:141):
 * out = conv(x) + bias + emb[b, :] (openaimodel.py:272) + residual (openaimodel.py:274).
 * Wp = [Cout][ky][kx][Cin] in `dtype`.  mode is one of CL_GEMM_CONV_*. */
int cl_conv3x3_fwd(int dtype, int mode, const void* x, long ldx, const void* Wp, const float* bias,
                   const void* emb, long ldemb, const void* residual, long ldr, void* y, long ldy,
                   int B, int Hin, int Win, int Cin, int Cout, const void* zero_page, void* stream);
/* data gradient: same kernel on the tap-flipped, in/out-swapped weights Wd = [Cin][2-ky][2-kx][Cout]
 * (mode CL_GEMM_CONV_S1 for stride-1 convs, CL_GEMM_CONV_T2 for the stride-2 Downsample). */
int cl_conv3x3_bwd_data(int dtype, int mode, const void* dy, long lddy, const void* Wd,
                        const void* accum, long ldacc, void* dx, long lddx,
                        int B, int Hdy, int Wdy, int Cout, int Cin, const void* zero_page, void* stream);

/* 1x1 convs (SpatialTransformer.proj_in/out, ResBlock.skip_connection, ControlNet zero convs,
 * cldm/cldm.py:281-282) with the ControlNet residual injection fused in
 * (cldm_ctrlora_finetune.py:79, cldm/cldm.py:35,41):  y = (x W^T + b) * scale + beta * residual. */
int cl_conv1x1_fwd(int dtype, const void* x, long ldx, const void* W, const float* bias, float scale,
                   const void* residual, long ldr, float beta, void* y, long ldy,
                   int M, int Cin, int Cout, void* stream);

/* ---- normalisation ------------------------------------------------------------------- */
/* GroupNorm32(32, C) [+ SiLU] in fp32 statistics (util.py:217-219; openaimodel.py:201-202). */
long cl_groupnorm_ws_floats(int B, int HW, int C);
int cl_groupnorm_silu_fwd(int dtype, const void* x, long ldx, void* y, long ldy, const float* gamma,
                          const float* beta, int B, int HW, int C, int groups, float eps, int silu,
                          float* stats, float* ws, void* stream);
int cl_groupnorm_silu_bwd(int dtype, const void* x, long ldx, const void* dy, long lddy,
                          const void* accum, long ldacc, void* dx, long lddx, const float* gamma,
                          const float* beta, const float* stats, int B, int HW, int C, int groups,
                          int silu, float* dgamma, float* dbeta, float* ws, void* stream);
/* nn.LayerNorm over the last dim (attention.py:263-265) */
int cl_layernorm_fwd(int dtype, const void* x, long ldx, void* y, long ldy, const float* gamma,
                     const float* beta, int M, int D, float eps, float* stats, void* stream);
int cl_layernorm_bwd(int dtype, const void* x, long ldx, const void* dy, long lddy, const void* accum,
                     long ldacc, void* dx, long lddx, const float* gamma, const float* stats, int M,
                     int D, float* dgamma, float* dbeta, void* stream);

/* ---- attention ----------------------------------------------------------------------- */
/* CrossAttention.forward core (attention.py:171-192): softmax(q k^T d^-1/2) v per head, fp32
 * scores/softmax, never materialising the score matrix.  Vt / Qt / dOt / Kt are the
 * [B][H*dh][pad64] transposes made with cl_transpose. */
int cl_attention_fwd(int dtype, const void* Q, long ldq, const void* K, long ldk, const void* Vt,
                     int nkv_pad, void* O, long ldo, float* LSE, int lse_stride, int B, int H, int N,
                     int Nkv, int dh, float scale, void* stream);
int cl_attention_bwd(int dtype, const void* Q, long ldq, const void* K, long ldk, const void* V,
                     long ldv, const void* O, long ldo, const void* dO, long lddo, const void* Qt,
                     const void* dOt, int n_pad, const void* Kt, int nkv_pad, const float* LSE,
                     float* Delta, int lse_stride, void* dQ, long lddq, void* dK, long lddk, void* dV,
                     long lddv, int B, int H, int N, int Nkv, int dh, float scale, void* stream);

/* bf16 variants without any materialised transposes (csrc/attention_tr.hip): every tile is staged
 * row-major as the projections wrote it and the P.V-type operands are built with the gfx950 LDS
 * transpose read.  V is [B*Nkv, ldv] like K.  fp32 parity mode keeps the entry points above.
 * flags (ABI 6): CL_ATTN_Q_PRESCALED -- Q holds q * scale * log2(e): the to_q projection applied the factor in its own
 * fp32 epilogue (cl_gemm_params.alpha / alpha_n), so the kernels read log2-domain scores straight off the matrix
 * product (for d_head 40 the forward also carries -max through a spare contraction slot: no per-score multiply-add
 * at all).  `scale` is still d_head^-0.5; O, LSE, dQ, dK, dV mean the same with or without the flag (gradients are
 * those of the TRUE q, k, v).  An explicit argument of every call -- no process-global state selects it. */
enum { CL_ATTN_Q_PRESCALED = 1 };
int cl_attention_fwd_v2(int dtype, const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                        void* O, long ldo, float* LSE, int lse_stride, int B, int H, int N, int Nkv, int dh,
                        float scale, int flags, void* stream);
int cl_attention_bwd_v2(int dtype, const void* Q, long ldq, const void* K, long ldk, const void* V, long ldv,
                        const void* O, long ldo, const void* dO, long lddo, const float* LSE, float* Delta,
                        int lse_stride, void* dQ, long lddq, void* dK, long lddk, void* dV, long lddv,
                        int B, int H, int N, int Nkv, int dh, float scale, int flags, void* row_ws, void* stream);
/* row_ws (ABI 6; may be NULL): scratch of B * H * lse_stride * 32 bytes, 16-byte aligned.  With CL_ATTN_Q_PRESCALED and
 * d_head 40 the backward kernels keep (-lse, -delta) of every query row there as bf16 triples and feed them through spare
 * contraction slots of the matrix products, which then deliver s - lse and dP - delta (attention.py:171-192's backward with
 * 3 instead of 5 vector instructions per score pair).  Results are the same with or without it. */

/* ---- elementwise / layout ------------------------------------------------------------ */
int cl_geglu_fwd(int dtype, const void* h, long ldh, void* out, long ldo, long M, int F, void* stream); /* attention.py:55-56 */
int cl_geglu_bwd(int dtype, const void* h, long ldh, const void* dout, long lddo, void* dh, long lddh, long M, int F, void* stream);
int cl_silu_fwd(int dtype, const void* x, void* y, long n, void* stream);
int cl_silu_bwd(int dtype, const void* x, const void* dy, void* dx, long n, void* stream);
int cl_axpby(int dtype, const void* x, long ldx, void* y, long ldy, long M, int C, float a, float b, void* stream);
int cl_transpose(int in_dtype, int out_dtype, const void* in, long ldi, long bsi, void* out, long ldo,
                 long bso, int Bt, int R, int C, int Rpad, void* stream);
int cl_nchw_to_tok(int dtype, const float* in, void* out, long ldo, int B, int Cin, int Cpad, int HW, void* stream);
int cl_tok_to_nchw(int dtype, const void* in, long ldi, float* out, int B, int C, int HW, float alpha, float beta, void* stream);
/* out[b][c] += scale * sum_p in[b*HW + p][c]  (zero-conv bias gradients, time-embedding gradients) */
int cl_colsum(int dtype, const void* in, long ldi, float* out, long ldo, int B, int HW, int C, float scale, void* stream);
int cl_pool2x2(int dtype, const void* in, long ldi, void* out, long ldo, int B, int H, int W, int C, int accumulate, void* stream);
int cl_pack2d(int dtype, const float* in, long ldi, void* out, long ldo, long R, int C, int Cpad, void* stream);
/* One-launch refresh of the engine's storage-dtype copies of all trainable matrices from the flat fp32
 * master buffer after an optimizer step.  desc = device table of 8 longs per matrix {src offset in floats,
 * rows << 32 | cols, dst [rows][cols] or 0, dst^T [cols][rows] or 0, source row stride (0 = cols), dst row stride
 * (0 = cols), dst^T row stride (0 = rows), reserved}; the strides let one 3x3-conv weight [O][9][I] be re-packed tap
 * by tap into [O][9][I_pad] and the tap-flipped data-gradient form [I][9][O_pad].  tile_prefix[i] = number of 32x32
 * tiles before matrix i (ndesc + 1 entries). */
int cl_repack(int dtype, const float* flat, const long* desc, const int* tile_prefix, int ndesc,
              int total_tiles, void* stream);
/* timestep_embedding (util.py:154-174); freqs = the fp32 table exp(-ln(1e4) * arange(half)/half) */
int cl_timestep_embedding(int dtype, const long* t, const float* freqs, void* out, long ldo, int B, int half, void* stream);

/* out[m, :] = x[pixel(m, tap), :] (zero outside the image), m = (b, oy, ox), pixel = (oy*stride + tap/3 - pad,
 * ox*stride + tap%3 - pad): the shifted operand of one tap of a 3x3 conv weight gradient, materialised (fp32 parity
 * mode; the bf16 weight-gradient kernel gathers in its own addressing, see cl_wgrad_desc.tap). */
int cl_conv_tap_gather(int dtype, const void* x, long ldx, void* out, long ldo, int B, int Hin, int Win, int Hout,
                       int Wout, int C, int tap, int stride, int pad, void* stream);
/* Row softmax for the VAE's single-head attention (ldm/modules/diffusionmodules/model.py:183-186): fp32 scores
 * S [M, N] (row stride lds) -> `dtype` probabilities P [M, N] (row stride ldp), P = softmax(S * scale) per row. */
int cl_softmax_rows(int dtype, const float* S, long lds, void* P, long ldp, long M, int N, float scale, void* stream);

/* ---- diffusion bookkeeping ------------------------------------------------------------ */
/* DDPM.q_sample (ddpm.py:356-359): out = sqrt_ac[t_b] * z + sqrt_1mac[t_b] * noise */
int cl_qsample(const float* z, const float* noise, const long* t, const float* sqrt_ac, const float* sqrt_1mac,
               float* out, int B, long per_sample, void* stream);
/* p_losses MSE (ddpm.py:902-918): *loss = mean((eps - target)^2); d_eps = 2 (eps - target) / n * gscale */
int cl_mse_loss(const float* eps, const float* target, float* d_eps, float* loss, long n, float gscale, void* stream);
/* LatentDiffusion.p_losses' reduction, deterministic (ddpm.py:902-918, eps-parameterisation, logvar == 0):
 *   out[0] = loss_simple = mean_b mean_chw (eps - target)^2        out[1] = loss_vlb = mean_b lvlb[t_b] * (per-sample mean)
 *   out[2] = loss = w_simple * loss_simple + w_elbo * loss_vlb     per_sample[B] (optional) = per-sample means
 *   d_eps (optional) = d (w_simple * loss_simple) / d eps * gscale.  scratch: 16 * B floats.  lvlb may be NULL. */
int cl_p_losses_mse(const float* eps, const float* target, float* d_eps, const long* t, const float* lvlb, float* out,
                    float* per_sample, float* scratch, int B, long per_sample_elems, float gscale, float w_simple,
                    float w_elbo, void* stream);
/* optimizer.zero_grad() / accumulator clears without an ATen launch: a fill KERNEL on `stream` (not hipMemsetAsync: memset
   nodes of small buffers were mis-ordered when a step is replayed as several consecutive hipGraphs) */
int cl_zero(void* p, long nbytes, void* stream);
/* DDIMSampler.p_sample_ddim update (cldm/ddim_hacked.py:192,203-231); coef = device [S][4] fp32 table
 * {a_t, a_prev, sigma_t, sqrt(1-a_t)}; e_u = NULL disables classifier-free guidance. */
int cl_ddim_step(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coef,
                 int index, float scale, float* x_prev, float* pred_x0, long n, void* stream);
/* torch.optim.AdamW step over one flat fp32 buffer (cldm_ctrlora_finetune.py:105) */
int cl_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
             float eps, float weight_decay, int step, float grad_scale, void* stream);

/* ---- device-resident step state: what a captured hipGraph needs --------------------------
 * A replayed graph re-issues the same kernel arguments, so per-step scalars live in device memory.
 * cl_tick: *counter += 1.  cl_adamw_dev: AdamW with hyper = device {lr, beta1, beta2, eps, weight_decay,
 * grad_scale} and a device step counter that the CALLER advances with cl_tick once per optimizer step, before the
 * first cl_adamw_dev of that step (one tick however many parameter banks follow; torch increments first too).  cl_ddim_set_t / cl_ddim_step_dev:
 * the DDIM loop body with a device cursor i (iteration number): index = S-1-i, ts[:] = ddim_timesteps[index]
 * (cldm/ddim_hacked.py:157-160,203-231); x_prev may alias x. */
int cl_tick(int* counter, void* stream);
int cl_adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, int* step, void* stream);
int cl_ddim_set_t(const long* table, const int* cursor, int S, long* ts, int n, void* stream);
int cl_ddim_step_dev(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coef,
                     const int* cursor, int S, float scale, float* x_prev, float* pred_x0, long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CTRLORA_HIP_H */
