// HBM-bound elementwise / layout kernels of the CtrLoRA hot path (gfx950).
// All activations are token-major [rows, C] with an explicit row stride so they
// can address column slices of wider buffers (decoder concat, fused QKV).
#include "elementwise.h"
#include "gemm.h"

namespace cl {

static inline int ew_grid(long nvec, int threads = 256) {
  long g = (nvec + threads - 1) / threads;
  if (g < 1) g = 1;
  return (int)(g < 4096 ? g : 4096);
}

// ---------------------------------------------------------------- GEGLU (attention.py:49-56)
template <typename T>
__global__ void geglu_fwd_kernel(const T* __restrict__ h, long ldh, T* __restrict__ out, long ldo, long M, int F) {
  const int F8 = F / 8;
  const long n = M * F8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / F8; const int v = (int)(i - r * F8);
    float a[8], g[8];
    load8(h + r * ldh + v * 8, a);
    load8(h + r * ldh + F + v * 8, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] *= gelu_f(g[e]);
    store8(out + r * ldo + v * 8, a);
  }
}
template <typename T>
__global__ void geglu_bwd_kernel(const T* __restrict__ h, long ldh, const T* __restrict__ dout, long lddo,
                                 T* __restrict__ dh, long lddh, long M, int F) {
  const int F8 = F / 8;
  const long n = M * F8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / F8; const int v = (int)(i - r * F8);
    float a[8], g[8], d[8], da[8], dg[8];
    load8(h + r * ldh + v * 8, a);
    load8(h + r * ldh + F + v * 8, g);
    load8(dout + r * lddo + v * 8, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) { da[e] = d[e] * gelu_f(g[e]); dg[e] = d[e] * a[e] * dgelu_f(g[e]); }
    store8(dh + r * lddh + v * 8, da);
    store8(dh + r * lddh + F + v * 8, dg);
  }
}

// ---------------------------------------------------------------- SiLU on small [rows, C] (emb path)
template <typename T>
__global__ void silu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8]; load8(x + i * 8, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
    store8(y + i * 8, f);
  }
}
template <typename T>
__global__ void silu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float f[8], d[8]; load8(x + i * 8, f); load8(dy + i * 8, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] *= dsilu_f(f[e]);
    store8(dx + i * 8, d);
  }
}

// ---------------------------------------------------------------- y = a*x + b*y over [rows, C] with strides
template <typename T>
__global__ void axpby_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, long M, int C,
                             float a, float b) {
  const int C8 = C / 8;
  const long n = M * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C8; const int v = (int)(i - r * C8);
    float f[8], g[8];
    load8(x + r * ldx + v * 8, f);
    if (b != 0.f) {
      load8(y + r * ldy + v * 8, g);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = a * f[e] + b * g[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = a * f[e];
    }
    store8(y + r * ldy + v * 8, f);
  }
}

// ---------------------------------------------------------------- batched transpose with zero padding
// in [Bt][R][C] (row stride ldi, batch stride bsi) -> out [Bt][C][Rpad] (row stride ldo >= Rpad, batch stride bso)
template <typename U, typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const U* __restrict__ in, long ldi, long bsi,
                                                        T* __restrict__ out, long ldo, long bso, int R, int C,
                                                        int Rpad) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const U* ib = in + (long)blockIdx.z * bsi;
  T* ob = out + (long)blockIdx.z * bso;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int r = r0 + ty + 4 * k, c = c0 + tx;
    tile[ty + 4 * k][tx] = (r < R && c < C) ? to_f<U>(ib[(long)r * ldi + c]) : 0.f;
  }
  __syncthreads();
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int c = c0 + ty + 4 * k, r = r0 + tx;
    if (c < C && r < Rpad) ob[(long)c * ldo + r] = from_f<T>(tile[tx][ty + 4 * k]);
  }
}

// ---------------------------------------------------------------- NCHW fp32 <-> token-major T
// in [B, Cin, HW] fp32 -> out [B*HW, ldo] T, channels [Cin, Cpad) zero filled
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_tok_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                          long ldo, int Cin, int Cpad, int HW) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int c = c0 + ty + 4 * k, p = p0 + tx;
    tile[ty + 4 * k][tx] = (c < Cin && p < HW) ? in[((long)b * Cin + c) * HW + p] : 0.f;
  }
  __syncthreads();
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int p = p0 + ty + 4 * k, c = c0 + tx;
    if (p < HW && c < Cpad) out[((long)b * HW + p) * ldo + c] = from_f<T>(tile[tx][ty + 4 * k]);
  }
}
// in [B*HW, ldi] T -> out [B, C, HW] fp32  (out = alpha*in + beta*out)
template <typename T>
__global__ __launch_bounds__(256) void tok_to_nchw_kernel(const T* __restrict__ in, long ldi,
                                                          float* __restrict__ out, int C, int HW, float alpha,
                                                          float beta) {
  __shared__ float tile[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int p = p0 + ty + 4 * k, c = c0 + tx;
    tile[ty + 4 * k][tx] = (p < HW && c < C) ? to_f<T>(in[((long)b * HW + p) * ldi + c]) : 0.f;
  }
  __syncthreads();
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int c = c0 + ty + 4 * k, p = p0 + tx;
    if (c < C && p < HW) {
      float* o = out + ((long)b * C + c) * HW + p;
      const float v = alpha * tile[tx][ty + 4 * k];
      *o = beta != 0.f ? v + beta * *o : v;
    }
  }
}

// ---------------------------------------------------------------- timestep embedding (util.py:154-174)
// out[b, i] = cos(t_b * f_i), out[b, half + i] = sin(t_b * f_i); freqs is the fp32 table the host
// builds exactly as the reference does (exp(-ln(1e4) * arange(half) / half)).
template <typename T>
__global__ void timestep_embed_kernel(const long* __restrict__ t, const float* __restrict__ freqs,
                                      T* __restrict__ out, long ldo, int B, int half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * half) return;
  const int b = i / half, k = i - b * half;
  const float arg = (float)t[b] * freqs[k];
  out[(long)b * ldo + k] = from_f<T>(cosf(arg));
  out[(long)b * ldo + half + k] = from_f<T>(sinf(arg));
}

// ---------------------------------------------------------------- q_sample + MSE (ddpm.py:356-359,902)
// x_noisy = sqrt_ac[t_b] * z + sqrt_1mac[t_b] * noise       (fp32, any layout; per = elems per sample)
__global__ void qsample_kernel(const float* __restrict__ z, const float* __restrict__ noise,
                               const long* __restrict__ t, const float* __restrict__ sqrt_ac,
                               const float* __restrict__ sqrt_1mac, float* __restrict__ out, long per, long n) {
  // two rounded products and a rounded sum, as torch evaluates a * z + b * noise: with the default contraction the sum
  // becomes an FMA and differs from the reference's x_noisy / stochastic_encode in the last bit of some elements
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long tb = t[i / per];
    const float a = sqrt_ac[tb] * z[i], b = sqrt_1mac[tb] * noise[i];
    out[i] = a + b;
  }
}
// loss += sum((eps - target)^2) / n ; d_eps = 2 (eps - target) / n * gscale
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ eps, const float* __restrict__ target,
                                                  float* __restrict__ d_eps, float* __restrict__ loss, long n,
                                                  float gscale) {
  __shared__ float red[4];
  float acc = 0.f;
  const float inv = 1.0f / (float)n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = eps[i] - target[i];
    acc += d * d;
    if (d_eps) d_eps[i] = 2.0f * d * inv * gscale;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (red[0] + red[1] + red[2] + red[3]) * inv);
}

// ---------------------------------------------------------------- 3x3-conv tap gather (fp32 parity mode of the conv dW)
// out[m, :] = x[pixel(m, tap), :] or 0 outside the image, m = (b, oy, ox): the shifted operand of one tap of a conv
// weight gradient.  The bf16 path does this inside wgrad_tn_kernel's DMA addressing; the fp32 parity mode, whose
// weight-gradient kernel wants explicit transposes anyway, materialises it.
template <typename T>
__global__ void conv_tap_gather_kernel(const T* __restrict__ x, long ldx, T* __restrict__ out, long ldo, int Hin, int Win,
                                       int Hout, int Wout, int C8, int tap, int stride, int pad, long total) {
  const int ky = tap / 3, kx = tap - 3 * ky;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / C8; const int c = (int)(i - m * C8) * 8;
    const int ox = (int)(m % Wout); const long t2 = m / Wout; const int oy = (int)(t2 % Hout); const long ob = t2 / Hout;
    const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win) load8(x + ((ob * Hin + iy) * Win + ix) * ldx + c, f);
    store8(out + m * ldo + c, f);
  }
}

// ---------------------------------------------------------------- row softmax (VAE mid-block attention)
// one workgroup per row; N <= 256 * 32: the row lives in registers (16-byte loads), fp32 math, exp2 with the
// scale folded in
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, long lds_, T* __restrict__ P,
                                                           long ldp, int N, float scale) {
  __shared__ float red[8];
  const float* s = S + (long)blockIdx.x * lds_;
  T* o = P + (long)blockIdx.x * ldp;
  constexpr int MAXV = 8;                       // 8 x float4 per thread = 8192 columns
  float4 v[MAXV];
  const float sl2 = scale * 1.4426950408889634f;
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 256 + threadIdx.x) * 4;
    if (c < N) {
      v[i] = *reinterpret_cast<const float4*>(s + c);
      mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * sl2;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 256 + threadIdx.x) * 4;
    if (c < N) {
      v[i].x = __builtin_amdgcn_exp2f(v[i].x * sl2 - mx); v[i].y = __builtin_amdgcn_exp2f(v[i].y * sl2 - mx);
      v[i].z = __builtin_amdgcn_exp2f(v[i].z * sl2 - mx); v[i].w = __builtin_amdgcn_exp2f(v[i].w * sl2 - mx);
      sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
  __syncthreads();
  const float inv = 1.0f / ((red[4] + red[5]) + (red[6] + red[7]));
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = (i * 256 + threadIdx.x) * 4;
    if (c < N) {
      const float w[4] = {v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv};
      store4(o + c, w);
    }
  }
}

// ---------------------------------------------------------------- p_losses reduction (ddpm.py:902-918)
// Deterministic (no float atomics): block (chunk, b) reduces one slice of sample b into scratch[b][chunk] and
// writes d_eps; the finishing block sums the slices in a fixed order:
//   out[0] = loss_simple = mean_b mean((eps-target)^2)          (logvar == 0)
//   out[1] = loss_vlb    = mean_b lvlb[t_b] * mean((eps-target)^2)
//   out[2] = loss        = l_simple_weight * loss_simple + elbo_weight * loss_vlb
constexpr int PL_CHUNKS = 16;
__global__ __launch_bounds__(256) void plosses_partial_kernel(const float* __restrict__ eps,
                                                              const float* __restrict__ target,
                                                              float* __restrict__ d_eps, float* __restrict__ scratch,
                                                              long per, float gmul) {
  __shared__ float red[4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const long base = (long)b * per;
  const long c0 = per * chunk / PL_CHUNKS, c1 = per * (chunk + 1) / PL_CHUNKS;
  float acc = 0.f;
  for (long i = c0 + threadIdx.x; i < c1; i += 256) {
    const float d = eps[base + i] - target[base + i];
    acc += d * d;
    if (d_eps) d_eps[base + i] = d * gmul;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) scratch[b * PL_CHUNKS + chunk] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void plosses_finish_kernel(const float* __restrict__ scratch, const long* __restrict__ t,
                                                            const float* __restrict__ lvlb, float* __restrict__ out,
                                                            float* __restrict__ per_sample, int B, long per,
                                                            float w_simple, float w_elbo) {
  if (threadIdx.x != 0) return;
  float ls = 0.f, lv = 0.f;
  for (int b = 0; b < B; ++b) {
    float s = 0.f;
    for (int c = 0; c < PL_CHUNKS; ++c) s += scratch[b * PL_CHUNKS + c];
    s /= (float)per;
    if (per_sample) per_sample[b] = s;
    ls += s;
    if (lvlb) lv += lvlb[t[b]] * s;
  }
  ls /= (float)B; lv /= (float)B;
  out[0] = ls; out[1] = lv; out[2] = w_simple * ls + w_elbo * lv;
}

// ---------------------------------------------------------------- DDIM update (ddim_hacked.py:192,203-231)
// coef = device table [S][4] = {a_t, a_prev, sigma_t, sqrt(1 - a_t)} (fp32), row `index` is used.
__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ e_c,
                                 const float* __restrict__ e_u, const float* __restrict__ noise,
                                 const float* __restrict__ coef, int index, float scale,
                                 float* __restrict__ x_prev, float* __restrict__ pred_x0, long n) {
  const float a_t = coef[index * 4 + 0], a_prev = coef[index * 4 + 1];
  const float sigma = coef[index * 4 + 2], s1m = coef[index * 4 + 3];
  const float sqrt_at = sqrtf(a_t), sqrt_ap = sqrtf(a_prev), dir = sqrtf(1.0f - a_prev - sigma * sigma);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float e = e_c[i];
    if (e_u) { const float u = e_u[i]; e = u + scale * (e - u); }
    const float p0 = (x[i] - s1m * e) / sqrt_at;
    float xp = sqrt_ap * p0 + dir * e;
    if (noise) xp += sigma * noise[i];
    x_prev[i] = xp;
    if (pred_x0) pred_x0[i] = p0;
  }
}

// ---------------------------------------------------------------- fused AdamW over one flat fp32 buffer
// torch.optim.AdamW defaults (cldm_ctrlora_finetune.py:105): decoupled decay, bias correction.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float beta1, float beta2, float eps,
                             float wd, float bc1, float bc2_sqrt, float gscale) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * mi / denom;
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// ---------------------------------------------------------------- device-resident step state (hipGraph replay)
// A captured graph replays the SAME kernel arguments, so anything that changes from step to step must
// live in device memory: the optimizer step count / hyper-parameters and the DDIM loop cursor.
__global__ void tick_kernel(int* c) { *c += 1; }

// hyper = {lr, beta1, beta2, eps, weight_decay, grad_scale}; *step was already incremented for this step
__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long n, const float* __restrict__ hyper,
                                 const int* __restrict__ step) {
  const float lr = hyper[0], beta1 = hyper[1], beta2 = hyper[2], eps = hyper[3], wd = hyper[4], gscale = hyper[5];
  const float st = (float)*step;
  const float bc1 = 1.0f - powf(beta1, st), bc2_sqrt = sqrtf(1.0f - powf(beta2, st));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * mi / denom;
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

// DDIM loop with a device cursor i = 0..S-1 (ddim_hacked.py:157-160): index = S-1-i, ts = ddim_timesteps[index]
__global__ void ddim_set_t_kernel(const long* __restrict__ table, const int* __restrict__ cursor, int S,
                                  long* __restrict__ ts, int n) {
  const int index = max(0, S - 1 - *cursor);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) ts[i] = table[index];
}
__global__ void ddim_step_dev_kernel(const float* __restrict__ x, const float* __restrict__ e_c,
                                     const float* __restrict__ e_u, const float* __restrict__ noise,
                                     const float* __restrict__ coef, const int* __restrict__ cursor, int S,
                                     float scale, float* __restrict__ x_prev, float* __restrict__ pred_x0, long n) {
  const int index = max(0, S - 1 - *cursor);
  const float a_t = coef[index * 4 + 0], a_prev = coef[index * 4 + 1];
  const float sigma = coef[index * 4 + 2], s1m = coef[index * 4 + 3];
  const float sqrt_at = sqrtf(a_t), sqrt_ap = sqrtf(a_prev), dir = sqrtf(1.0f - a_prev - sigma * sigma);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float e = e_c[i];
    if (e_u) { const float u = e_u[i]; e = u + scale * (e - u); }
    const float p0 = (x[i] - s1m * e) / sqrt_at;
    float xp = sqrt_ap * p0 + dir * e;
    if (noise) xp += sigma * noise[i];
    x_prev[i] = xp;            // x_prev may alias x (element-wise)
    if (pred_x0) pred_x0[i] = p0;
  }
}

// ---------------------------------------------------------------- 2x2 sum pool (data-gradient of nearest x2)
// in [B, 2H, 2W, C] (ldi) -> out [B, H, W, C] (ldo), out (+)= sum of the 2x2 block
template <typename T>
__global__ void pool2x2_kernel(const T* __restrict__ in, long ldi, T* __restrict__ out, long ldo, int B, int H,
                               int W, int C, int accumulate) {
  const int C8 = C / 8;
  const long n = (long)B * H * W * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % C8); long r = i / C8;
    const int x = (int)(r % W); r /= W; const int y = (int)(r % H); const int b = (int)(r / H);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const long orow = ((long)b * H + y) * W + x;
    if (accumulate) load8(out + orow * ldo + v * 8, s);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float f[8];
        load8(in + (((long)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * ldi + v * 8, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
    store8(out + orow * ldo + v * 8, s);
  }
}

// ---------------------------------------------------------------- per-sample column sums
// out[b, c] += sum_p in[b*HW + p, c]   (fp32 atomics; data-gradient of the `h + emb_out[:, :, None, None]`
// broadcast, openaimodel.py:272, and bias gradients of the zero convs)
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ in, long ldi, float* __restrict__ out, long ldo, int HW, int C,
                              int ppc, float scale, float* __restrict__ partial) {
  // partial != nullptr: the block's column sums go to partial[(b * nchunk + chunk) * C + c] and a second kernel
  // adds them up (one writer per column: deterministic, and no same-cache-line atomics -- 512 blocks x 320
  // atomics onto ten cache lines cost ~25 us, five times the 21 MB read itself)
  __shared__ float red[256 * 8];
  const int C8 = C / 8;
  const int b = blockIdx.y, p0 = blockIdx.x * ppc, p1 = min(HW, p0 + ppc);
  const int VX = blockDim.x < C8 ? blockDim.x : C8;
  const int PY = blockDim.x / VX;
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const bool live = py < PY;
  // uniform trip count (C8 <= VX, or C8 a multiple of VX is NOT required: guard inside)
  for (int v0 = 0; v0 < C8; v0 += VX) {
    const int v = v0 + vx;
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    if (live && v < C8) {
      const T* src = in + ((long)b * HW) * ldi + v * 8;
      int p = p0 + py;
      for (; p + 3 * PY < p1; p += 4 * PY) {   // four independent loads in flight
        float f0[8], f1[8], f2[8], f3[8];
        load8(src + (long)p * ldi, f0);
        load8(src + (long)(p + PY) * ldi, f1);
        load8(src + (long)(p + 2 * PY) * ldi, f2);
        load8(src + (long)(p + 3 * PY) * ldi, f3);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += (f0[e] + f1[e]) + (f2[e] + f3[e]);
      }
      for (; p < p1; p += PY) {
        float f[8];
        load8(src + (long)p * ldi, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += f[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
    __syncthreads();
    if (py == 0 && v < C8) {
      for (int k = 1; k < PY; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += red[(k * VX + vx) * 8 + e];
      if (partial) {
        float* dst = partial + ((long)b * gridDim.x + blockIdx.x) * C + v * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[e] = s[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(out + (long)b * ldo + v * 8 + e, s[e] * scale);
      }
    }
    __syncthreads();
  }
}

// out[b][c] += scale * sum_k partial[(b * nchunk + k) * C + c]; 32 columns x 32 chunk lanes per workgroup
__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float* __restrict__ partial, int nchunk, int C,
                                                             float* __restrict__ out, long ldo, float scale) {
  __shared__ float red[1024];
  const int b = blockIdx.y, cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < C)
#pragma unroll 8
    for (int k = kl; k < nchunk; k += 32) s += partial[((long)b * nchunk + k) * C + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (kl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k * 32 + cl];
    out[(long)b * ldo + c] += scale * t;
  }
}

// ---------------------------------------------------------------- strided 2-D convert fp32 -> T
template <typename T>
__global__ void pack_kernel(const float* __restrict__ in, long ldi, T* __restrict__ out, long ldo, long R, int C,
                            int Cpad) {
  const long n = R * Cpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / Cpad; const int c = (int)(i - r * Cpad);
    out[r * ldo + c] = from_f<T>(c < C ? in[r * ldi + c] : 0.f);
  }
}

// ================================================================= host launchers
int geglu_fwd(int dtype, const void* h, long ldh, void* out, long ldo, long M, int F, hipStream_t st) {
  if (F % 8 || ldh % 8 || ldo % 8) return CL_EINVAL;
  if (dtype == CL_BF16) hipLaunchKernelGGL((geglu_fwd_kernel<bf16_t>), dim3(ew_grid(M * (F / 8))), dim3(256), 0, st, (const bf16_t*)h, ldh, (bf16_t*)out, ldo, M, F);
  else hipLaunchKernelGGL((geglu_fwd_kernel<float>), dim3(ew_grid(M * (F / 8))), dim3(256), 0, st, (const float*)h, ldh, (float*)out, ldo, M, F);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int geglu_bwd(int dtype, const void* h, long ldh, const void* dout, long lddo, void* dh, long lddh, long M, int F, hipStream_t st) {
  if (F % 8 || ldh % 8 || lddo % 8 || lddh % 8) return CL_EINVAL;
  if (dtype == CL_BF16) hipLaunchKernelGGL((geglu_bwd_kernel<bf16_t>), dim3(ew_grid(M * (F / 8))), dim3(256), 0, st, (const bf16_t*)h, ldh, (const bf16_t*)dout, lddo, (bf16_t*)dh, lddh, M, F);
  else hipLaunchKernelGGL((geglu_bwd_kernel<float>), dim3(ew_grid(M * (F / 8))), dim3(256), 0, st, (const float*)h, ldh, (const float*)dout, lddo, (float*)dh, lddh, M, F);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int silu_fwd(int dtype, const void* x, void* y, long n, hipStream_t st) {
  if (n % 8) return CL_EINVAL;
  if (dtype == CL_BF16) hipLaunchKernelGGL((silu_fwd_kernel<bf16_t>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n / 8);
  else hipLaunchKernelGGL((silu_fwd_kernel<float>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const float*)x, (float*)y, n / 8);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int silu_bwd(int dtype, const void* x, const void* dy, void* dx, long n, hipStream_t st) {
  if (n % 8) return CL_EINVAL;
  if (dtype == CL_BF16) hipLaunchKernelGGL((silu_bwd_kernel<bf16_t>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n / 8);
  else hipLaunchKernelGGL((silu_bwd_kernel<float>), dim3(ew_grid(n / 8)), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, n / 8);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int axpby(int dtype, const void* x, long ldx, void* y, long ldy, long M, int C, float a, float b, hipStream_t st) {
  if (C % 8 || ldx % 8 || ldy % 8) return CL_EINVAL;
  if (dtype == CL_BF16) hipLaunchKernelGGL((axpby_kernel<bf16_t>), dim3(ew_grid(M * (C / 8))), dim3(256), 0, st, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, M, C, a, b);
  else hipLaunchKernelGGL((axpby_kernel<float>), dim3(ew_grid(M * (C / 8))), dim3(256), 0, st, (const float*)x, ldx, (float*)y, ldy, M, C, a, b);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int transpose(int in_dtype, int out_dtype, const void* in, long ldi, long bsi, void* out, long ldo, long bso,
              int Bt, int R, int C, int Rpad, hipStream_t st) {
  if (Rpad < R || ldo < Rpad) return CL_EINVAL;
  dim3 grid((Rpad + 63) / 64, (C + 63) / 64, Bt);
  if (in_dtype == CL_F32 && out_dtype == CL_BF16) hipLaunchKernelGGL((transpose_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float*)in, ldi, bsi, (bf16_t*)out, ldo, bso, R, C, Rpad);
  else if (in_dtype == CL_F32 && out_dtype == CL_F32) hipLaunchKernelGGL((transpose_kernel<float, float>), grid, dim3(256), 0, st, (const float*)in, ldi, bsi, (float*)out, ldo, bso, R, C, Rpad);
  else if (in_dtype == CL_BF16 && out_dtype == CL_BF16) hipLaunchKernelGGL((transpose_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, ldi, bsi, (bf16_t*)out, ldo, bso, R, C, Rpad);
  else return CL_EINVAL;
  CL_CHECK_LAUNCH(); return CL_OK;
}
int nchw_to_tok(int dtype, const float* in, void* out, long ldo, int B, int Cin, int Cpad, int HW, hipStream_t st) {
  if (Cpad < Cin || ldo < Cpad) return CL_EINVAL;
  dim3 grid((HW + 63) / 64, (Cpad + 63) / 64, B);
  if (dtype == CL_BF16) hipLaunchKernelGGL((nchw_to_tok_kernel<bf16_t>), grid, dim3(256), 0, st, in, (bf16_t*)out, ldo, Cin, Cpad, HW);
  else hipLaunchKernelGGL((nchw_to_tok_kernel<float>), grid, dim3(256), 0, st, in, (float*)out, ldo, Cin, Cpad, HW);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int tok_to_nchw(int dtype, const void* in, long ldi, float* out, int B, int C, int HW, float alpha, float beta, hipStream_t st) {
  dim3 grid((HW + 63) / 64, (C + 63) / 64, B);
  if (dtype == CL_BF16) hipLaunchKernelGGL((tok_to_nchw_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, ldi, out, C, HW, alpha, beta);
  else hipLaunchKernelGGL((tok_to_nchw_kernel<float>), grid, dim3(256), 0, st, (const float*)in, ldi, out, C, HW, alpha, beta);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int timestep_embed(int dtype, const long* t, const float* freqs, void* out, long ldo, int B, int half, hipStream_t st) {
  const int n = B * half;
  if (dtype == CL_BF16) hipLaunchKernelGGL((timestep_embed_kernel<bf16_t>), dim3((n + 255) / 256), dim3(256), 0, st, t, freqs, (bf16_t*)out, ldo, B, half);
  else hipLaunchKernelGGL((timestep_embed_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, st, t, freqs, (float*)out, ldo, B, half);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int qsample(const float* z, const float* noise, const long* t, const float* sqrt_ac, const float* sqrt_1mac,
            float* out, int B, long per, hipStream_t st) {
  hipLaunchKernelGGL(qsample_kernel, dim3(ew_grid(B * per)), dim3(256), 0, st, z, noise, t, sqrt_ac, sqrt_1mac, out, per, (long)B * per);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int mse_loss(const float* eps, const float* target, float* d_eps, float* loss, long n, float gscale, hipStream_t st) {
  if (int rc = zero_bytes(loss, sizeof(float), st)) return rc;
  hipLaunchKernelGGL(mse_kernel, dim3(ew_grid(n) < 256 ? ew_grid(n) : 256), dim3(256), 0, st, eps, target, d_eps, loss, n, gscale);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int ddim_step(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coef, int index,
              float scale, float* x_prev, float* pred_x0, long n, hipStream_t st) {
  hipLaunchKernelGGL(ddim_step_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, e_c, e_u, noise, coef, index, scale, x_prev, pred_x0, n);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
          float wd, int step, float gscale, hipStream_t st) {
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, wd, bc1, sqrtf(bc2), gscale);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int plosses_mse(const float* eps, const float* target, float* d_eps, const long* t, const float* lvlb, float* out,
                float* per_sample, float* scratch, int B, long per, float gscale, float w_simple, float w_elbo,
                hipStream_t st) {
  if (B < 1 || per < 1) return CL_EINVAL;
  // d loss / d eps with loss = w_simple * mean(...) (the elbo term is not differentiated: weight 0 in every config)
  const float gmul = 2.0f * gscale * w_simple / ((float)per * (float)B);
  hipLaunchKernelGGL(plosses_partial_kernel, dim3(PL_CHUNKS, B), dim3(256), 0, st, eps, target, d_eps, scratch, per, gmul);
  hipLaunchKernelGGL(plosses_finish_kernel, dim3(1), dim3(64), 0, st, scratch, t, lvlb, out, per_sample, B, per, w_simple, w_elbo);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int conv_tap_gather(int dtype, const void* x, long ldx, void* out, long ldo, int B, int Hin, int Win, int Hout, int Wout,
                    int C, int tap, int stride, int pad, hipStream_t st) {
  if (C % 8 || ldx % 8 || ldo % 8 || tap < 0 || tap > 8) return CL_EINVAL;
  const long total = (long)B * Hout * Wout * (C / 8);
  if (dtype == CL_BF16) hipLaunchKernelGGL((conv_tap_gather_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, st, (const bf16_t*)x, ldx, (bf16_t*)out, ldo, Hin, Win, Hout, Wout, C / 8, tap, stride, pad, total);
  else hipLaunchKernelGGL((conv_tap_gather_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, st, (const float*)x, ldx, (float*)out, ldo, Hin, Win, Hout, Wout, C / 8, tap, stride, pad, total);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int softmax_rows(int dtype, const float* S, long lds_, void* P, long ldp, long M, int N, float scale, hipStream_t st) {
  if (N % 4 || N > 8192 || lds_ % 4 || ldp % 4 || M <= 0) return CL_EINVAL;
  if (dtype == CL_BF16) hipLaunchKernelGGL((softmax_rows_kernel<bf16_t>), dim3((unsigned)M), dim3(256), 0, st, S, lds_, (bf16_t*)P, ldp, N, scale);
  else hipLaunchKernelGGL((softmax_rows_kernel<float>), dim3((unsigned)M), dim3(256), 0, st, S, lds_, (float*)P, ldp, N, scale);
  CL_CHECK_LAUNCH(); return CL_OK;
}
// Clears are KERNEL nodes, not hipMemsetAsync: a step captured as SEVERAL consecutive hipGraphs in one memory pool
// (the data-parallel segments, train.py) replayed memset nodes of small buffers out of order with their neighbours on
// ROCm 7.2 -- garbage in exactly the gradients that are accumulated into freshly cleared scratch
// (tests/tools/debug_segmented.py).  A fill kernel is ordered like every other launch and costs the same bytes.
__global__ __launch_bounds__(256) void zero_kernel(unsigned char* __restrict__ p, long head, long nvec, long tail) {
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  uint4* body = reinterpret_cast<uint4*>(p + head);
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (long i = i0; i < nvec; i += (long)gridDim.x * blockDim.x) body[i] = z;
  if (i0 < head) p[i0] = 0;
  if (i0 < tail) p[head + nvec * 16 + i0] = 0;
}
int zero_bytes(void* p, long nbytes, hipStream_t st) {
  if (nbytes <= 0) return CL_OK;
  long head = (long)((16 - ((uintptr_t)p & 15)) & 15);
  if (head > nbytes) head = nbytes;
  const long nvec = (nbytes - head) / 16, tail = nbytes - head - nvec * 16;
  hipLaunchKernelGGL(zero_kernel, dim3(ew_grid(nvec > 16 ? nvec : 16)), dim3(256), 0, st, (unsigned char*)p, head, nvec, tail);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int tick(int* counter, hipStream_t st) {
  hipLaunchKernelGGL(tick_kernel, dim3(1), dim3(1), 0, st, counter);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, int* step, hipStream_t st) {
  hipLaunchKernelGGL(adamw_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, st, p, g, m, v, n, hyper, step);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int ddim_set_t(const long* table, const int* cursor, int S, long* ts, int n, hipStream_t st) {
  hipLaunchKernelGGL(ddim_set_t_kernel, dim3(1), dim3(256), 0, st, table, cursor, S, ts, n);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int ddim_step_dev(const float* x, const float* e_c, const float* e_u, const float* noise, const float* coef,
                  const int* cursor, int S, float scale, float* x_prev, float* pred_x0, long n, hipStream_t st) {
  hipLaunchKernelGGL(ddim_step_dev_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, e_c, e_u, noise, coef, cursor, S, scale, x_prev, pred_x0, n);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int pool2x2(int dtype, const void* in, long ldi, void* out, long ldo, int B, int H, int W, int C, int accumulate, hipStream_t st) {
  if (C % 8 || ldi % 8 || ldo % 8) return CL_EINVAL;
  const long n = (long)B * H * W * (C / 8);
  if (dtype == CL_BF16) hipLaunchKernelGGL((pool2x2_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, B, H, W, C, accumulate);
  else hipLaunchKernelGGL((pool2x2_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)in, ldi, (float*)out, ldo, B, H, W, C, accumulate);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int colsum(int dtype, const void* in, long ldi, float* out, long ldo, int B, int HW, int C, float scale, hipStream_t st) {
  if (C % 8 || ldi % 8) return CL_EINVAL;
  int nchunk = (HW + 63) / 64;
  int want = (512 + B - 1) / B;
  if (nchunk > want) nchunk = want;
  const int ppc = (HW + nchunk - 1) / nchunk;
  nchunk = (HW + ppc - 1) / ppc;
  dim3 grid(nchunk, B);
  // block partials through the stream's registered scratch (cl_set_workspace) when there is one; otherwise
  // fp32 atomics straight into out
  void* wsp = nullptr; long wsb = 0;
  gemm_get_workspace_for(st, &wsp, &wsb);
  float* partial = (nchunk > 1 && wsp && (long)B * nchunk * C * 4 <= wsb) ? (float*)wsp : nullptr;
  if (dtype == CL_BF16) hipLaunchKernelGGL((colsum_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, ldi, out, ldo, HW, C, ppc, scale, partial);
  else hipLaunchKernelGGL((colsum_kernel<float>), grid, dim3(256), 0, st, (const float*)in, ldi, out, ldo, HW, C, ppc, scale, partial);
  if (partial)
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 31) / 32, B), dim3(1024), 0, st, partial, nchunk, C, out, ldo, scale);
  CL_CHECK_LAUNCH(); return CL_OK;
}
int pack2d(int dtype, const float* in, long ldi, void* out, long ldo, long R, int C, int Cpad, hipStream_t st) {
  if (Cpad < C || ldo < Cpad) return CL_EINVAL;
  if (dtype == CL_BF16) hipLaunchKernelGGL((pack_kernel<bf16_t>), dim3(ew_grid(R * Cpad)), dim3(256), 0, st, in, ldi, (bf16_t*)out, ldo, R, C, Cpad);
  else hipLaunchKernelGGL((pack_kernel<float>), dim3(ew_grid(R * Cpad)), dim3(256), 0, st, in, ldi, (float*)out, ldo, R, C, Cpad);
  CL_CHECK_LAUNCH(); return CL_OK;
}

// ------------------------------------------------------------------ fused re-pack of the trainables
// After every optimizer step the engine needs its storage-dtype copies of the trainable matrices
// (LoRA down/up, zero convs) in both orientations (W for the forward NT product, W^T for the data
// gradient).  Round 0 issued one pack + one transpose launch per matrix (~500 launches / 2.6 ms per
// step); this is ONE launch over a device-resident descriptor table: workgroup -> (matrix, 32x32 tile)
// by binary search in the tile prefix, fp32 tile in, straight and transposed tiles out.
//   desc[i] = {src offset (floats) in the flat master, rows << 32 | cols, dst pointer, dst^T pointer (or 0),
//              src row stride, dst row stride, dst^T row stride (0 = dense), reserved}
template <typename T>
__global__ __launch_bounds__(256) void repack_kernel(const float* __restrict__ flat, const long* __restrict__ desc,
                                                     const int* __restrict__ tile_prefix, int ndesc) {
  __shared__ float tile[32][33];
  int lo = 0, hi = ndesc - 1;
  const int blk = blockIdx.x;
  while (lo < hi) {                       // first matrix whose prefix end exceeds blk
    const int mid = (lo + hi) >> 1;
    if (tile_prefix[mid + 1] > blk) hi = mid; else lo = mid + 1;
  }
  const long* d = desc + (long)lo * 8;
  const long src = d[0];
  const int R = (int)(d[1] >> 32), C = (int)(d[1] & 0xffffffff);
  T* dst = reinterpret_cast<T*>(d[2]);
  T* dstT = reinterpret_cast<T*>(d[3]);
  const long lds_ = d[4] ? d[4] : C, ldd = d[5] ? d[5] : C, ldt = d[6] ? d[6] : R;
  const int t = blk - tile_prefix[lo];
  const int tc = (C + 31) / 32;
  const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    const float v = (r < R && c < C) ? flat[src + (long)r * lds_ + c] : 0.f;
    tile[ty + 8 * i][tx] = v;
    if (dst && r < R && c < C) dst[(long)r * ldd + c] = from_f<T>(v);
  }
  if (!dstT) return;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (r < R && c < C) dstT[(long)c * ldt + r] = from_f<T>(tile[tx][ty + 8 * i]);
  }
}

int repack(int dtype, const float* flat, const long* desc, const int* tile_prefix, int ndesc, int total_tiles,
           hipStream_t st) {
  if (ndesc <= 0 || total_tiles <= 0) return CL_OK;
  if (dtype == CL_BF16) hipLaunchKernelGGL((repack_kernel<bf16_t>), dim3(total_tiles), dim3(256), 0, st, flat, desc, tile_prefix, ndesc);
  else hipLaunchKernelGGL((repack_kernel<float>), dim3(total_tiles), dim3(256), 0, st, flat, desc, tile_prefix, ndesc);
  CL_CHECK_LAUNCH(); return CL_OK;
}

}  // namespace cl
