// Parameter block of the MFMA GEMM / implicit-conv core (gemm.hip).
#pragma once
#include "common.h"

namespace cl {

enum GemmMode {
  GEMM_LINEAR = 0,   // A is row-major [M, K1]
  GEMM_CONV_S1 = 1,  // A is NHWC [B,Hin,Win,C]; 3x3, stride 1, pad 1
  GEMM_CONV_S2 = 2,  // 3x3, stride 2, pad 1 (Downsample, openaimodel.py:150)
  GEMM_CONV_UP2 = 3, // 3x3 over the nearest-x2 upsampled input (Upsample, openaimodel.py:115-117)
  GEMM_CONV_T2 = 4,  // 3x3 over the zero-stuffed x2 grid (= data-gradient of GEMM_CONV_S2)
  GEMM_CONV_S2A = 5, // 3x3, stride 2, pad (0,1,0,1): the VAE encoder's Downsample (ldm/modules/diffusionmodules/model.py:80-84)
  // Phase-decomposed forms of UP2 / T2 (full-line kernel only).  An output pixel (2y + a, 2x + b) of either product touches
  // only a 2 x 2 (UP2) or (1 + a) x (1 + b) (T2) window of SOURCE pixels, so each of the four output phases (a, b) is a small
  // stride-1 window product on the source grid -- 4 K1 (UP2: the 3x3 taps that fall on the same source pixel are summed
  // into one weight) or on average 2.25 K1 (T2: the taps whose zero-stuffed input is non-zero) deep instead of 9 K1.
  // A1 = source NHWC [B, Hin, Win, K1]; M = 4 B Hin Win, internal row order [phase = 2a + b][b][y][x] (B Hin Win a multiple of
  // the tile height), mapped to output row ((b Hin + y) 2 + a) 2 Win + 2 x + b in the epilogue (C, residual, rowbias);
  // W1 = phase-packed weights, phase ph at element offset N K1 * {0, 4, 8, 12} (UP2P) / {0, 1, 3, 5} (T2P), rows [N][taps][K1],
  // tap t = ty * ntx + tx reading source pixel (y + dy0 + ty, x + dx0 + tx): UP2P dy0 = a - 1, dx0 = b - 1, 2 x 2 taps;
  // T2P dy0 = dx0 = 0, (1 + a) x (1 + b) taps.  ldw1 is ignored.
  GEMM_CONV_UP2P = 6,
  GEMM_CONV_T2P = 7,
  // 4x4 window, stride 2, pad 1 (full-line kernel only): the data gradient of UP2 (nearest x2 + 3x3 conv) taken directly on
  // the SOURCE grid -- source pixel y receives from upsampled rows 2y - 1 .. 2y + 2, with the 3x3 taps that coincide summed
  // (16 K1 deep at M / 4 rows = 4 K1 M instead of the 9 K1 M of a stride-1 data gradient on the upsampled grid + a 2x2 pool).
  // A1 = dy NHWC [B, Hin, Win, K1] on the upsampled grid, Hout = Hin / 2, Wout = Win / 2, W1 = [N][4][4][K1] (ldw1 = 16 K1).
  GEMM_CONV_S2K4 = 8,
};
__host__ __device__ inline bool gemm_phase_mode(int mode) { return mode == GEMM_CONV_UP2P || mode == GEMM_CONV_T2P; }

enum GemmAct {
  ACT_NONE = 0, ACT_SILU = 1,
  // GEGLU fused into the projection (attention.py:49-56), inference / no-grad forwards only: W's rows are
  // permuted so that every 160-column tile holds 80 value columns followed by their 80 gate columns; the
  // epilogue writes value * gelu(gate) into C[M, N/2] (ldc).  Needs N % 160 == 0, no rowbias / residual.
  ACT_GEGLU = 2,
  // The same fusion for the x-stationary kernel (gemm_xs.hip) ONLY: W's rows stay in their natural order [value (N / 2) | gate
  // (N / 2)] (attention.py:55: chunk(2, dim=-1)).  Needs K1 in {320, 640}, K2 in {0, 128}, N % 64 == 0; other kernels return CL_EINVAL.
  ACT_GEGLU_SPLIT = 3
};

struct GemmParams {
  // out[M,N] = act( A1 . W1^T + A2 . W2^T + bias[n] + rowbias[m / rows_per_batch, n] ) * alpha
  //            + beta * residual[m, n]
  const void* A1; long lda1; int K1;   // K1 = per-tap channels C in conv modes
  const void* W1; long ldw1;           // [N, taps*K1], K contiguous
  const void* A2; long lda2; int K2;   // optional second K segment (LoRA up-projection folded in)
  const void* W2; long ldw2;
  int M, N;
  int mode;
  int B, Hin, Win, Hout, Wout;         // conv geometry (A1 pixel stride = lda1)
  const void* zero_page;               // >= 64 zero bytes, for halo taps
  const float* bias;                   // [N] fp32 or null
  const void* rowbias; long ldrb; int rows_per_batch;  // T [M/rows_per_batch, N] or null
  const void* residual; long ldr;      // T [M, N] or null
  float alpha, beta;
  int act;
  void* C; long ldc;
  int out_f32;                         // store fp32 regardless of T
  int atomic;                          // fp32 atomicAdd into C (grad accumulation; caller may set splitk)
  int splitk;                          // atomic mode: K splits.  Otherwise chosen by the launcher (workspace slabs)
  // Grouped K segments (linear mode): several LoRA linears that share their input run as ONE product whose output
  // columns are the linears side by side.  Output columns [g * a2_group_n, (g+1) * a2_group_n) take their SECOND segment
  // from columns [g * K2, (g+1) * K2) of A2 (q | k | v with their own x A^T); with a1_group_n the FIRST segment comes from
  // columns [g * K1, (g+1) * K1) of A1 (u_g = dy_g B_g for all g in one launch).  0 = ungrouped.  A tile never straddles
  // groups: the launcher only takes configurations whose BN divides the group width.
  int a1_group_n, a2_group_n;
  // alpha applies to output columns [0, alpha_n) only (0 = all columns): the producer of a fused q | k | v writes
  // q * (d_head^-0.5 * log2 e) -- the attention kernels' pre-scaled-Q contract -- and leaves k, v alone.  Multiple of 8.
  int alpha_n;
  // LayerNorm as a PROLOGUE of the product (x-stationary kernel only, gemm_xs.hip; bf16, K1 in {320, 640}, K2 = 0):
  // A1 holds the UN-normalised rows; the kernel forms LN(row) = (row - mean) * rstd * gamma + beta over the K1 columns in
  // registers (fp32 statistics, two passes, rounded to bf16 exactly as the stand-alone cl_layernorm_fwd stores it) before
  // the MFMAs -- BasicTransformerBlock's norm1/2/3 (ldm/modules/attention.py:271-275) never exist in HBM.  ln_stats
  // (optional, [M][2] fp32: mean, rstd) is what cl_layernorm_bwd needs when a backward pass follows.
  const float* ln_gamma; const float* ln_beta; float ln_eps; float* ln_stats;
};

int launch_gemm(const GemmParams& p, int dtype, hipStream_t stream);
// x-stationary streaming product (gemm_xs.hip, bf16, launch configuration 34): CL_EINVAL when the product is not one it covers.
// nsplit: column runs per group (0 = the launcher's rule).
int launch_gemm_xs(const GemmParams& p, hipStream_t stream, int nsplit);
// loader / consumer tile kernel (gemm_w4.hip, bf16, launch configurations 40 / 41 = 256 x 160 / 256 x 128 tiles; 47 / 48 = their
// persistent forms): four consumer
// waves (one per SIMD) run nothing but the MFMA stream and its fragment reads, four loader waves issue every LDS-DMA.
// CL_EINVAL when the product is not one it covers (K segments not whole 128-byte lines, fp32 atomics).
int launch_gemm_w4(const GemmParams& p, hipStream_t stream, int bn, int persist);   // persist: one workgroup per CU walks the tiles
int gemm_pick_splitk(GemmParams& p, long tiles, int steps, int want, int min_steps, float** slab, hipStream_t stream);
void gemm_launch_splitk_reduce_bf16(const GemmParams& p, const float* slab, hipStream_t stream);
// Device scratch for the deterministic split-K path (fp32 partial slabs).  Owned by the host;
// one workspace per process, used stream-ordered by whichever stream launches the GEMM.
void gemm_set_workspace(void* p, long bytes);
void gemm_get_workspace(void** p, long* bytes);
void gemm_get_workspace_for(hipStream_t st, void** p, long* bytes);
int gemm_set_stream_workspace(hipStream_t st, void* p, long bytes);
// transpose-free weight gradient (wgrad.hip, bf16 only): dW[N,K] += alpha * dy[M,N]^T . x[M,K]
int launch_wgrad_tn(const void* dy, long lddy, const void* x, long ldx, float* dW, long lddw, int M, int N, int K,
                    float alpha, const void* zero_page, hipStream_t stream);
// grouped form: many (dy, x, dW) problems in one launch (+ one reduce launch)
// tap >= 0: x is an NHWC activation [B*Hin*Win, K] and row m = (b, oy, ox) of dy pairs with the input pixel
// (oy*stride + tap/3 - pad, ox*stride + tap%3 - pad) (zero outside the image): one tap of a 3x3 conv's weight gradient,
// dW pointing at that tap's [N, K] slice of a [N][9][K] gradient (lddw = 9 K).  tap < 0: plain dy^T x.
struct WgradDesc {
  const void* dy; long lddy; const void* x; long ldx; float* dW; long lddw; int M, N, K; float alpha;
  int tap, Hin, Win, Hout, Wout, stride, pad, reserved;
};
int launch_wgrad_tn_group(const WgradDesc* probs, int n, const void* zero_page, hipStream_t stream);
extern int g_wgrad_blocks, g_wgrad_min_steps, g_wgrad_ring, g_wgrad_rows;
extern int g_fl128_split_want, g_tiny_m_minsub;
extern int g_fl_persist_stagger;
extern int g_gemm_force_splitk;   // tuning hook: > 0 imposes the split-K factor of the workspace path
int gemm_tune_set(int dtype, int mode, int M, int N, int K1, int K2, int geglu, int cfg, int splitk);
void gemm_tune_clear();
int gemm_tune_size();
extern int g_gemm_force_cfg;   // tuning/probe hook (tile configuration override), -1 = heuristic
extern int g_gemm_xs_rules;    // 1 = untabled signatures may take the x-stationary kernel by rule (gemm.hip: launch_t_cfg)
// launch tags (csrc/debug_hooks.h: cl_debug_gemm_tag): extra, immediately exiting workgroups that name the product signature
extern int g_gemm_tag_on;
int gemm_cur_tag();                                   // tag of the product being launched on this thread (0 = tagging off)
void gemm_tag_note(long real_wgs, int wg_size);       // the main kernel's real grid, for the table
int gemm_tag_count();
int gemm_tag_get(int i, long* out12);

}  // namespace cl
