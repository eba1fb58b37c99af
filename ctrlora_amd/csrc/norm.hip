// GroupNorm(32)+SiLU and LayerNorm, forward and backward, for NHWC / token-major
// activations on gfx950.  HBM-bound kernels: 16-byte vector loads, fp32
// statistics (GroupNorm32 computes in fp32, ldm/modules/diffusionmodules/util.py:217-219),
// wave64 shuffle reductions, SiLU fused into the normalisation pass.
//
//   GroupNorm fwd : gn_partial (per-channel sum / sumsq per pixel chunk)
//                   -> gn_finalize (per-(b,group) mean/rstd, per-(b,channel) scale/shift)
//                   -> gn_apply   (y = silu?(x*scale + shift))
//   GroupNorm bwd : gn_bwd_partial -> gn_bwd_finalize (+ dgamma/dbeta) -> gn_bwd_apply
//   LayerNorm     : one wave per row, row kept in registers (D <= 2048)
//
// Reference ops replaced: ResBlock in_layers[0:2]/out_layers[0:2]
// (openaimodel.py:201-202,225-226; eps 1e-5), SpatialTransformer.norm
// (attention.py:88-89,295; eps 1e-6, no SiLU), BasicTransformerBlock.norm1-3
// (attention.py:263-265; eps 1e-5), UNetModel.out[0:2] (openaimodel.py:726-728).
#include <algorithm>
#include "norm.h"
#include "gemm.h"

namespace cl {

static constexpr int GN_MAX_THREADS = 512;

struct GnGeom { int VX, PY, threads, nchunk, ppc; };

static GnGeom gn_geom(int B, int HW, int C) {
  GnGeom g;
  const int C8 = C / 8;
  g.VX = C8 < 320 ? C8 : 320;
  if (g.VX > GN_MAX_THREADS) g.VX = GN_MAX_THREADS;
  g.PY = 256 / g.VX; if (g.PY < 1) g.PY = 1;
  g.threads = g.VX * g.PY;
  // enough blocks to fill 256 CUs a few times over, at least 4 pixels per py lane
  int want = (1024 + B - 1) / B;
  int maxc = HW / (g.PY * 4); if (maxc < 1) maxc = 1;
  g.nchunk = want < maxc ? want : maxc;
  if (g.nchunk > 256) g.nchunk = 256;
  g.ppc = (HW + g.nchunk - 1) / g.nchunk;
  g.nchunk = (HW + g.ppc - 1) / g.ppc;
  return g;
}

long gn_ws_floats(int B, int HW, int C) {
  GnGeom g = gn_geom(B, HW, C);
  return (long)B * g.nchunk * C * 2 + (long)B * C * 4;
}

// ------------------------------------------------------------------ forward

template <typename T>
__global__ void gn_partial_kernel(const T* __restrict__ x, long ldx, int HW, int C, int VX, int PY,
                                  int ppc, int nchunk, float* __restrict__ partial) {
  extern __shared__ float red[];  // [PY][VX][16]
  const int C8 = C / 8;
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  for (int v = vx; v < C8; v += VX) {
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
  #pragma unroll 4
  for (int p = p0 + py; p < p1; p += PY) {
      float f[8];
      load8(x + ((long)b * HW + p) * ldx + v * 8, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
    float* r = red + (py * VX + vx) * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { r[e] = s[e]; r[8 + e] = q[e]; }
    __syncthreads();
    if (py == 0) {
      for (int k = 1; k < PY; ++k) {
        const float* o = red + (k * VX + vx) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += o[e]; q[e] += o[8 + e]; }
      }
      float* dst = partial + (((long)b * nchunk + chunk) * C + v * 8) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) { dst[2 * e] = s[e]; dst[2 * e + 1] = q[e]; }
    }
    __syncthreads();
  }
}

// block-wide sum of two doubles (256 threads = 4 waves); result valid in every thread
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red /*[8]*/) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[wave] = a; red[4 + wave] = b; }
  __syncthreads();
  a = red[0] + red[1] + red[2] + red[3];
  b = red[4] + red[5] + red[6] + red[7];
}

// per-(b,group) mean/rstd and per-(b,channel) scale/shift
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nchunk, int HW, int C,
                                                          int G, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ stats /*[B][G][2]*/,
                                                          float* __restrict__ coef /*[B][C][2]*/) {
  // one workgroup per (sample, group): 256 threads sweep the group's (chunk, channel) partials, coalesced
  __shared__ double red[8];
  const int g = blockIdx.x, b = blockIdx.y, cg = C / G, tid = threadIdx.x;
  double s = 0, q = 0;
  const int total = nchunk * cg;
  for (int i = tid; i < total; i += 256) {
    const int k = i / cg, c = g * cg + (i - k * cg);
    const float2 p = *reinterpret_cast<const float2*>(partial + (((long)b * nchunk + k) * C + c) * 2);
    s += p.x; q += p.y;
  }
  block_sum2(s, q, red);
  const double n = (double)HW * cg;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  if (tid == 0) {
    stats[((long)b * G + g) * 2] = (float)mean;
    stats[((long)b * G + g) * 2 + 1] = (float)rstd;
  }
  for (int c = g * cg + tid; c < (g + 1) * cg; c += 256) {
    const double sc = rstd * (double)gamma[c];
    coef[((long)b * C + c) * 2] = (float)sc;
    coef[((long)b * C + c) * 2 + 1] = (float)((double)beta[c] - mean * sc);
  }
}

template <typename T, bool SILU>
__global__ void gn_apply_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int HW, int C,
                                int VX, int PY, int ppc, const float* __restrict__ coef) {
  const int C8 = C / 8;
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  for (int v = vx; v < C8; v += VX) {
    float sc[8], sh[8];
    const float* cf = coef + ((long)b * C + v * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = cf[2 * e]; sh[e] = cf[2 * e + 1]; }
  #pragma unroll 4
  for (int p = p0 + py; p < p1; p += PY) {
      float f[8];
      const long row = (long)b * HW + p;
      load8(x + row * ldx + v * 8, f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float z = f[e] * sc[e] + sh[e];
        f[e] = SILU ? silu_f(z) : z;
      }
      store8(y + row * ldy + v * 8, f);
    }
  }
}

// ---- two-launch forms (no finalize launch).  The statistics pass reduces its chunk to per-GROUP sums
// ([B][nchunk][G][2] floats); every workgroup of the apply pass then sums the nchunk partials of its sample itself
// (fp64, fixed order: deterministic) -- G * nchunk * 8 bytes from L2, a microsecond -- instead of a third launch whose
// ~6 us are almost all launch latency (149 such launches per training step).  Needs one channel pass per lane
// (C <= 2560) and G <= 64.

// LDS layout helper: channel totals [C][2] floats reuse the py-reduction scratch
template <typename T>
__global__ void gn_gpartial_kernel(const T* __restrict__ x, long ldx, int HW, int C, int G, int VX, int PY, int ppc,
                                   int nchunk, float* __restrict__ gpartial) {
  extern __shared__ float red[];  // max([PY][VX][16], [C][2]) floats
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
#pragma unroll 4
  for (int p = p0 + py; p < p1; p += PY) {
    float f[8];
    load8(x + ((long)b * HW + p) * ldx + vx * 8, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
  }
  float* r = red + (py * VX + vx) * 16;
#pragma unroll
  for (int e = 0; e < 8; ++e) { r[e] = s[e]; r[8 + e] = q[e]; }
  __syncthreads();
  if (py == 0) {
    for (int k = 1; k < PY; ++k) {
      const float* o = red + (k * VX + vx) * 16;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += o[e]; q[e] += o[8 + e]; }
    }
  }
  __syncthreads();
  if (py == 0) {
    float* ch = red + vx * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ch[2 * e] = s[e]; ch[2 * e + 1] = q[e]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int cg = C / G, c0 = threadIdx.x * cg;
    float S = 0.f, Q = 0.f;
    for (int c = c0; c < c0 + cg; ++c) { S += red[2 * c]; Q += red[2 * c + 1]; }
    float* dst = gpartial + (((long)b * nchunk + chunk) * G + threadIdx.x) * 2;
    dst[0] = S; dst[1] = Q;
  }
}

// sum the nchunk group partials of sample b: every thread group (g = tid % G, lane kl = tid / G) takes chunks
// kl, kl + KL, ...; result per group in tot[g][0..1] (double), valid after the trailing barrier
__device__ __forceinline__ void gn_group_totals(const float* __restrict__ gpartial, int b, int nchunk, int G,
                                                double (*part)[64][2], double (*tot)[2]) {
  const int tid = threadIdx.x;
  int KL = blockDim.x / G; if (KL > 8) KL = 8;
  const int g = tid % G, kl = tid / G;
  if (kl < KL) {
    // four independent L2 loads in flight per lane: this walk (nchunk / KL ~ 18 partials) is the prologue of every apply
    // workgroup and, one load at a time, cost about as long as the workgroup's whole pixel loop
    double s = 0, q = 0;
    const float2* src = reinterpret_cast<const float2*>(gpartial) + (long)b * nchunk * G + g;
    int k = kl;
    for (; k + 3 * KL < nchunk; k += 4 * KL) {
      const float2 p0 = src[(long)k * G], p1 = src[(long)(k + KL) * G];
      const float2 p2 = src[(long)(k + 2 * KL) * G], p3 = src[(long)(k + 3 * KL) * G];
      s += ((double)p0.x + (double)p1.x) + ((double)p2.x + (double)p3.x);
      q += ((double)p0.y + (double)p1.y) + ((double)p2.y + (double)p3.y);
    }
    for (; k < nchunk; k += KL) {
      const float2 p = src[(long)k * G];
      s += p.x; q += p.y;
    }
    part[kl][g][0] = s; part[kl][g][1] = q;
  }
  __syncthreads();
  if (tid < G) {
    double s = 0, q = 0;
    for (int k = 0; k < KL; ++k) { s += part[k][tid][0]; q += part[k][tid][1]; }
    tot[tid][0] = s; tot[tid][1] = q;
  }
  __syncthreads();
}

template <typename T, bool SILU>
__global__ void gn_gapply_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int HW, int C, int G,
                                 int VX, int PY, int ppc, int nchunk, float eps, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, const float* __restrict__ gpartial,
                                 float* __restrict__ stats /*[B][G][2]*/) {
  __shared__ double part[8][64][2];
  __shared__ double tot[64][2];
  const int b = blockIdx.y, chunk = blockIdx.x, cg = C / G;
  gn_group_totals(gpartial, b, nchunk, G, part, tot);
  if ((int)threadIdx.x < G) {
    const double n = (double)HW * cg;
    const double mean = tot[threadIdx.x][0] / n;
    double var = tot[threadIdx.x][1] / n - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    tot[threadIdx.x][0] = mean; tot[threadIdx.x][1] = rstd;
    if (chunk == 0) {
      stats[((long)b * G + threadIdx.x) * 2] = (float)mean;
      stats[((long)b * G + threadIdx.x) * 2 + 1] = (float)rstd;
    }
  }
  __syncthreads();
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vx * 8 + e, gi = c / cg;
    const double scd = tot[gi][1] * (double)gamma[c];
    sc[e] = (float)scd;
    sh[e] = (float)((double)beta[c] - tot[gi][0] * scd);
  }
#pragma unroll 4
  for (int p = p0 + py; p < p1; p += PY) {
    float f[8];
    const long row = (long)b * HW + p;
    load8(x + row * ldx + vx * 8, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float z = f[e] * sc[e] + sh[e];
      f[e] = SILU ? silu_f(z) : z;
    }
    store8(y + row * ldy + vx * 8, f);
  }
}

int g_gn_three_pass = 0;   // A/B hook: 1 = always the partial -> finalize -> apply form

// ------------------------------------------------------------------ one-launch GroupNorm (register-resident slab)
// At the 32x32 / 16x16 / 8x8 levels the two-launch form above is bound by launch and dependency latency, not bytes:
// both of its kernels last ~9 us whatever the tensor size (88 forward + 61 backward GroupNorms per training step).
// Here ONE workgroup owns (sample b, channel block of CB = lcm(C / G, 8) channels = whole groups, 16-byte aligned) and
// keeps the block's [HW][CB] slab in registers between the statistics and the apply: the tensor is read once and written
// once, in one launch (the forward stores mean / rstd for the backward as before).  Lanes are laid out LPR (8 or 16) per
// pixel row -- CB / 8 of them active -- so that the per-channel sums reduce over the pixels of a wave with shuffles and
// over the waves through a small LDS array.  Taken when the grid (B * C / CB workgroups of up to 1024 threads) is at
// least GN1_MIN_WG and the slab fits NV vectors per lane; everything else (the 64x64 level, whose groups span 4096
// pixels) stays on the two-launch form.  The backward handles trainable norms too (dgamma / dbeta: one float atomic per
// channel and sample, as the three-launch form did).
static constexpr int GN1_MIN_WG = 96;
int g_gn_one_pass = 1;        // A/B hook (cl_debug_groupnorm_form): 0 = never take the one-launch form

template <typename T> struct Pack8;
template <> struct Pack8<bf16_t> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float f[8]) const {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};
template <> struct Pack8<float> {
  float4 a, b;
  __device__ __forceinline__ void load(const float* p) { a = reinterpret_cast<const float4*>(p)[0]; b = reinterpret_cast<const float4*>(p)[1]; }
  __device__ __forceinline__ void get(float f[8]) const {
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
};

struct Gn1Geom { int CB, VX, LPR, NW, NV, nblk; bool ok; };

static Gn1Geom gn1_geom(int B, int HW, int C, int G, int esize, bool bwd) {
  Gn1Geom g{}; g.ok = false;
  if (!g_gn_one_pass || g_gn_three_pass || C % G) return g;
  const int cg = C / G;
  int cb = cg;
  while (cb % 8) cb += cg;                 // lcm(cg, 8)
  if (cb > 128 || C % cb) return g;
  g.CB = cb; g.VX = cb / 8; g.LPR = g.VX <= 8 ? 8 : 16;
  const int ppw = 64 / g.LPR;
  // forward: up to 16 waves (128 VGPRs per lane); backward: up to 8 waves (256 VGPRs: x AND dy stay in registers next to
  // ~80 registers of per-channel coefficients)
  const int maxw = bwd ? 8 : 16;
  int nw = (HW + ppw - 1) / ppw; if (nw > maxw) nw = maxw;
  // the per-channel reduction (gn1_block_channel_sums) and the dgamma / dbeta atomics need one THREAD per channel of the
  // block: at least ceil(CB / 64) waves even when HW is tiny (C = 2560 at 2x2: CB = 80, HW = 4).  Padding waves own pixels
  // p >= HW and contribute zeros.
  const int minw = (cb + 63) / 64;
  if (nw < minw) nw = minw;
  g.NW = nw;
  const int per_iter = nw * ppw;
  const int nv = (HW + per_iter - 1) / per_iter;
  g.NV = nv <= 1 ? 1 : nv <= 2 ? 2 : nv <= 4 ? 4 : nv <= 8 ? 8 : 16;
  g.nblk = C / cb;
  g.ok = nv <= (esize == 2 ? 16 : 8) && (long)g.nblk * B >= GN1_MIN_WG;
  return g;
}

// per-channel sums of two quantities over the workgroup's pixels; result valid for lanes / threads that read chs afterwards
//   s[e], q[e]: this lane's sums for its 8 channels (lane % LPR = vector index; inactive lanes hold zeros)
__device__ __forceinline__ void gn1_block_channel_sums(float s[8], float q[8], int LPR, int NW, float* red /*[NW][16][16]*/,
                                                       float* chs /*[128][2]*/, int VX) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o >= 8; o >>= 1) {
    if (o >= LPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += __shfl_xor(s[e], o, 64); q[e] += __shfl_xor(q[e], o, 64); }
    }
  }
  if (lane < LPR) {
    float* r = red + (wave * 16 + lane) * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { r[e] = s[e]; r[8 + e] = q[e]; }
  }
  __syncthreads();
  const int t = threadIdx.x;
  if (t < VX * 8) {                        // one thread per channel of the block
    const int vx = t >> 3, e = t & 7;
    float a = 0.f, b = 0.f;
    for (int w = 0; w < NW; ++w) { a += red[(w * 16 + vx) * 16 + e]; b += red[(w * 16 + vx) * 16 + 8 + e]; }
    chs[2 * t] = a; chs[2 * t + 1] = b;
  }
  __syncthreads();
}

template <typename T, bool SILU, int NV>
__global__ __launch_bounds__(1024) void gn1_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, int HW,
                                                       int C, int G, int CB, int VX, int LPR, int NW, float eps,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ stats) {
  __shared__ float red[16 * 16 * 16];
  __shared__ float chs[128 * 2];
  __shared__ float gst[16 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vx = lane % LPR, pl = lane / LPR, ppw = 64 / LPR;
  const int b = blockIdx.y, c0 = blockIdx.x * CB, cg = C / G;
  const bool act = vx < VX;
  const int step = NW * ppw;
  Pack8<T> d[NV];
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = wave * ppw + pl + i * step;
    if (act && p < HW) d[i].load(x + ((long)b * HW + p) * ldx + c0 + vx * 8);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = wave * ppw + pl + i * step;
    if (act && p < HW) {
      float f[8]; d[i].get(f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
  }
  gn1_block_channel_sums(s, q, LPR, NW, red, chs, VX);
  const int ng = CB / cg;
  if ((int)threadIdx.x < ng) {
    double S = 0, Q = 0;
    for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) { S += chs[2 * c]; Q += chs[2 * c + 1]; }
    const double n = (double)HW * cg, mean = S / n;
    double var = Q / n - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    gst[2 * threadIdx.x] = (float)mean; gst[2 * threadIdx.x + 1] = (float)rstd;
    const int gg = c0 / cg + threadIdx.x;
    stats[((long)b * G + gg) * 2] = (float)mean;
    stats[((long)b * G + gg) * 2 + 1] = (float)rstd;
  }
  __syncthreads();
  if (!act) return;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int cl = vx * 8 + e, gi = cl / cg;
    const double scd = (double)gst[2 * gi + 1] * (double)gamma[c0 + cl];
    sc[e] = (float)scd;
    sh[e] = (float)((double)beta[c0 + cl] - (double)gst[2 * gi] * scd);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = wave * ppw + pl + i * step;
    if (p < HW) {
      float f[8]; d[i].get(f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = f[e] * sc[e] + sh[e];
        f[e] = SILU ? silu_f(z) : z;
      }
      store8(y + ((long)b * HW + p) * ldy + c0 + vx * 8, f);
    }
  }
}

template <typename T, bool SILU, int NV>
__global__ __launch_bounds__(512) void gn1_bwd_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                                       const T* __restrict__ accum, long ldacc, T* __restrict__ dx, long lddx,
                                                       int HW, int C, int G, int CB, int VX, int LPR, int NW,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ stats, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta) {
  __shared__ float red[16 * 16 * 16];
  __shared__ float chs[128 * 2];
  __shared__ float gst[16 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int vx = lane % LPR, pl = lane / LPR, ppw = 64 / LPR;
  const int b = blockIdx.y, c0 = blockIdx.x * CB, cg = C / G;
  const bool act = vx < VX;
  const int step = NW * ppw;
  Pack8<T> dxv[NV], ddv[NV];
  float ga[8], be[8], mu[8], rs[8], s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int cl = (act ? vx : 0) * 8 + e, gi = (c0 + cl) / cg;
    ga[e] = gamma[c0 + cl]; be[e] = beta[c0 + cl];
    mu[e] = stats[((long)b * G + gi) * 2]; rs[e] = stats[((long)b * G + gi) * 2 + 1];
    s[e] = 0.f; q[e] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = wave * ppw + pl + i * step;
    if (act && p < HW) {
      dxv[i].load(x + ((long)b * HW + p) * ldx + c0 + vx * 8);
      ddv[i].load(dy + ((long)b * HW + p) * lddy + c0 + vx * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = wave * ppw + pl + i * step;
    if (act && p < HW) {
      float f[8], d[8]; dxv[i].get(f); ddv[i].get(d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (f[e] - mu[e]) * rs[e];
        float dz = d[e];
        if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
        s[e] += dz; q[e] += dz * xh;
      }
    }
  }
  gn1_block_channel_sums(s, q, LPR, NW, red, chs, VX);
  const int ng = CB / cg;
  if (dgamma && (int)threadIdx.x < CB) {
    atomicAdd(dgamma + c0 + threadIdx.x, chs[2 * threadIdx.x + 1]);
    atomicAdd(dbeta + c0 + threadIdx.x, chs[2 * threadIdx.x]);
  }
  if ((int)threadIdx.x < ng) {
    double s1 = 0, s2 = 0;
    for (int c = threadIdx.x * cg; c < (threadIdx.x + 1) * cg; ++c) {
      s1 += (double)gamma[c0 + c] * chs[2 * c]; s2 += (double)gamma[c0 + c] * chs[2 * c + 1];
    }
    gst[2 * threadIdx.x] = (float)s1; gst[2 * threadIdx.x + 1] = (float)s2;   // gamma-weighted group sums
  }
  __syncthreads();
  if (!act) return;
  const double n = (double)HW * cg;
  float k1[8], k2[8], k3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int cl = vx * 8 + e, gi = cl / cg;
    const double rstd = rs[e];
    k1[e] = (float)(rstd * ga[e]);
    k2[e] = (float)(rstd * (double)gst[2 * gi] / n);
    k3[e] = (float)(rstd * (double)gst[2 * gi + 1] / n);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = wave * ppw + pl + i * step;
    if (p < HW) {
      float f[8], d[8], ac[8]; dxv[i].get(f); ddv[i].get(d);
      const long row = (long)b * HW + p;
      if (accum) load8(accum + row * ldacc + c0 + vx * 8, ac);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (f[e] - mu[e]) * rs[e];
        float dz = d[e];
        if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
        float r = dz * k1[e] - k2[e] - xh * k3[e];
        if (accum) r += ac[e];
        f[e] = r;
      }
      store8(dx + row * lddx + c0 + vx * 8, f);
    }
  }
}

template <typename T, bool SILU>
static void gn1_fwd_launch(const GnArgs& a, const Gn1Geom& g, hipStream_t st) {
  dim3 grid(g.nblk, a.B), blk(g.NW * 64);
#define GN1F(N_) hipLaunchKernelGGL((gn1_fwd_kernel<T, SILU, N_>), grid, blk, 0, st, (const T*)a.x, a.ldx, (T*)a.y, a.ldy, a.HW, a.C, \
                                    a.G, g.CB, g.VX, g.LPR, g.NW, a.eps, a.gamma, a.beta, a.stats)
  switch (g.NV) { case 1: GN1F(1); break; case 2: GN1F(2); break; case 4: GN1F(4); break; case 8: GN1F(8); break; default: GN1F(16); }
#undef GN1F
}

template <typename T, bool SILU>
static void gn1_bwd_launch(const GnBwdArgs& a, const Gn1Geom& g, hipStream_t st) {
  dim3 grid(g.nblk, a.B), blk(g.NW * 64);
#define GN1B(N_) hipLaunchKernelGGL((gn1_bwd_kernel<T, SILU, N_>), grid, blk, 0, st, (const T*)a.x, a.ldx, (const T*)a.dy, a.lddy,  \
                                    (const T*)a.accum, a.ldacc, (T*)a.dx, a.lddx, a.HW, a.C, a.G, g.CB, g.VX, g.LPR, g.NW, a.gamma, \
                                    a.beta, a.stats, a.dgamma, a.dbeta)
  switch (g.NV) { case 1: GN1B(1); break; case 2: GN1B(2); break; case 4: GN1B(4); break; case 8: GN1B(8); break; default: GN1B(16); }
#undef GN1B
}

static bool gn_two_pass_ok(const GnGeom& g, int C, int G) {
  return !g_gn_three_pass && C / 8 == g.VX && G <= 64 && g.threads >= G && g.threads >= 64;
}

template <typename T>
static int gn_fwd_t(const GnArgs& a, hipStream_t st) {
  const Gn1Geom g1 = gn1_geom(a.B, a.HW, a.C, a.G, (int)sizeof(T), false);
  if (g1.ok) {
    if (a.silu) gn1_fwd_launch<T, true>(a, g1, st); else gn1_fwd_launch<T, false>(a, g1, st);
    CL_CHECK_LAUNCH();
    return CL_OK;
  }
  const GnGeom g = gn_geom(a.B, a.HW, a.C);
  if (gn_two_pass_ok(g, a.C, a.G)) {
    dim3 grid(g.nchunk, a.B);
    const int lds = std::max(g.threads * 64, a.C * 8);
    hipLaunchKernelGGL((gn_gpartial_kernel<T>), grid, dim3(g.threads), lds, st, (const T*)a.x, a.ldx, a.HW, a.C, a.G,
                       g.VX, g.PY, g.ppc, g.nchunk, a.ws);
    if (a.silu)
      hipLaunchKernelGGL((gn_gapply_kernel<T, true>), grid, dim3(g.threads), 0, st, (const T*)a.x, a.ldx, (T*)a.y, a.ldy,
                         a.HW, a.C, a.G, g.VX, g.PY, g.ppc, g.nchunk, a.eps, a.gamma, a.beta, a.ws, a.stats);
    else
      hipLaunchKernelGGL((gn_gapply_kernel<T, false>), grid, dim3(g.threads), 0, st, (const T*)a.x, a.ldx, (T*)a.y, a.ldy,
                         a.HW, a.C, a.G, g.VX, g.PY, g.ppc, g.nchunk, a.eps, a.gamma, a.beta, a.ws, a.stats);
    CL_CHECK_LAUNCH();
    return CL_OK;
  }
  float* partial = a.ws;
  float* coef = a.ws + (long)a.B * g.nchunk * a.C * 2;
  dim3 grid(g.nchunk, a.B);
  hipLaunchKernelGGL((gn_partial_kernel<T>), grid, dim3(g.threads), g.threads * 64, st,
                     (const T*)a.x, a.ldx, a.HW, a.C, g.VX, g.PY, g.ppc, g.nchunk, partial);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(a.G, a.B), dim3(256), 0, st,
                     partial, g.nchunk, a.HW, a.C, a.G, a.eps, a.gamma, a.beta, a.stats, coef);
  if (a.silu)
    hipLaunchKernelGGL((gn_apply_kernel<T, true>), grid, dim3(g.threads), 0, st,
                       (const T*)a.x, a.ldx, (T*)a.y, a.ldy, a.HW, a.C, g.VX, g.PY, g.ppc, coef);
  else
    hipLaunchKernelGGL((gn_apply_kernel<T, false>), grid, dim3(g.threads), 0, st,
                       (const T*)a.x, a.ldx, (T*)a.y, a.ldy, a.HW, a.C, g.VX, g.PY, g.ppc, coef);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

int gn_fwd(const GnArgs& a, int dtype, hipStream_t st) {
  if (a.C % 8 || a.C % a.G || a.ldx % 8 || a.ldy % 8 || a.C > 8192) return CL_EINVAL;
  if (a.C / 8 > 320 && (a.C / 8) % 320) return CL_EINVAL;  // uniform trip count across the block
  if (gnc_fwd(a, dtype, st) == CL_OK) return CL_OK;        // groups spanning >= 1024 pixels: one launch, one pass (norm_coop.hip)
  return dtype == CL_BF16 ? gn_fwd_t<bf16_t>(a, st) : gn_fwd_t<float>(a, st);
}

// ------------------------------------------------------------------ backward

template <typename T, bool SILU>
__global__ void gn_bwd_partial_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                      int HW, int C, int G, int VX, int PY, int ppc, int nchunk,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ stats, float* __restrict__ partial) {
  extern __shared__ float red[];
  const int C8 = C / 8, cg = C / G;
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  for (int v = vx; v < C8; v += VX) {
    float ga[8], be[8], mu[8], rs[8], s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = v * 8 + e, g = c / cg;
      ga[e] = gamma[c]; be[e] = beta[c];
      mu[e] = stats[((long)b * G + g) * 2]; rs[e] = stats[((long)b * G + g) * 2 + 1];
      s[e] = 0.f; q[e] = 0.f;
    }
  #pragma unroll 4
  for (int p = p0 + py; p < p1; p += PY) {
      float f[8], d[8];
      const long row = (long)b * HW + p;
      load8(x + row * ldx + v * 8, f);
      load8(dy + row * lddy + v * 8, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (f[e] - mu[e]) * rs[e];
        float dz = d[e];
        if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
        s[e] += dz; q[e] += dz * xh;
      }
    }
    float* r = red + (py * VX + vx) * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { r[e] = s[e]; r[8 + e] = q[e]; }
    __syncthreads();
    if (py == 0) {
      for (int k = 1; k < PY; ++k) {
        const float* o = red + (k * VX + vx) * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] += o[e]; q[e] += o[8 + e]; }
      }
      float* dst = partial + (((long)b * nchunk + chunk) * C + v * 8) * 2;
#pragma unroll
      for (int e = 0; e < 8; ++e) { dst[2 * e] = s[e]; dst[2 * e + 1] = q[e]; }
    }
    __syncthreads();
  }
}

// per (b, group): channel sums -> group sums -> apply coefficients; optional dgamma/dbeta accumulation
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, int nchunk, int HW,
                                                              int C, int G, const float* __restrict__ gamma,
                                                              const float* __restrict__ stats,
                                                              float* __restrict__ bcoef /*[B][C][4]: k1,k2,k3,_*/,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
  // one workgroup per (sample, group).  Thread t owns channel t % CL of the group and the chunks
  // k = t / CL, t / CL + 256 / CL, ...  (CL = channels rounded up to a power of two <= 256); per-channel
  // totals are combined through LDS so that dgamma / dbeta need one atomic per channel.
  __shared__ double red[8];
  __shared__ double chs[256], chq[256];
  const int g = blockIdx.x, b = blockIdx.y, cg = C / G, tid = threadIdx.x;
  int CL = 1;
  while (CL < cg && CL < 256) CL <<= 1;
  const int KL = 256 / CL;                         // chunk lanes per channel
  const int cl = tid % CL, kl = tid / CL;
  double s1 = 0, s2 = 0;
  for (int c0 = 0; c0 < cg; c0 += CL) {            // one round unless cg > 256
    const int c = g * cg + c0 + cl;
    double s = 0, q = 0;
    if (c0 + cl < cg) {
      for (int k = kl; k < nchunk; k += KL) {
        const float2 p = *reinterpret_cast<const float2*>(partial + (((long)b * nchunk + k) * C + c) * 2);
        s += p.x; q += p.y;
      }
    }
    __syncthreads();
    chs[tid] = s; chq[tid] = q;
    __syncthreads();
    if (kl == 0 && c0 + cl < cg) {
      for (int k = 1; k < KL; ++k) { s += chs[k * CL + cl]; q += chq[k * CL + cl]; }
      if (dgamma) { atomicAdd(dgamma + c, (float)q); atomicAdd(dbeta + c, (float)s); }
      s1 += (double)gamma[c] * s; s2 += (double)gamma[c] * q;
    }
  }
  block_sum2(s1, s2, red);
  const double n = (double)HW * cg;
  const double rstd = stats[((long)b * G + g) * 2 + 1];
  const float k2 = (float)(rstd * s1 / n), k3 = (float)(rstd * s2 / n);
  for (int c = g * cg + tid; c < (g + 1) * cg; c += 256) {
    float* o = bcoef + ((long)b * C + c) * 4;
    o[0] = (float)(rstd * gamma[c]); o[1] = k2; o[2] = k3; o[3] = 0.f;
  }
}

template <typename T, bool SILU>
__global__ void gn_bwd_apply_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                    const T* __restrict__ accum, long ldacc, T* __restrict__ dx, long lddx,
                                    int HW, int C, int G, int VX, int PY, int ppc,
                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ stats, const float* __restrict__ bcoef) {
  const int C8 = C / 8, cg = C / G;
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  for (int v = vx; v < C8; v += VX) {
    float ga[8], be[8], mu[8], rs[8], k1[8], k2[8], k3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = v * 8 + e, g = c / cg;
      ga[e] = gamma[c]; be[e] = beta[c];
      mu[e] = stats[((long)b * G + g) * 2]; rs[e] = stats[((long)b * G + g) * 2 + 1];
      const float* o = bcoef + ((long)b * C + c) * 4;
      k1[e] = o[0]; k2[e] = o[1]; k3[e] = o[2];
    }
  #pragma unroll 4
  for (int p = p0 + py; p < p1; p += PY) {
      float f[8], d[8], ac[8];
      const long row = (long)b * HW + p;
      load8(x + row * ldx + v * 8, f);
      load8(dy + row * lddy + v * 8, d);
      if (accum) load8(accum + row * ldacc + v * 8, ac);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (f[e] - mu[e]) * rs[e];
        float dz = d[e];
        if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
        float r = dz * k1[e] - k2[e] - xh * k3[e];
        if (accum) r += ac[e];
        f[e] = r;
      }
      store8(dx + row * lddx + v * 8, f);
    }
  }
}

// two-launch backward for FROZEN norms (no dgamma / dbeta): the statistics pass forms the gamma-weighted group sums
// s1 = sum_c gamma_c sum dz, s2 = sum_c gamma_c sum dz xhat of its chunk; the apply pass totals them itself.
template <typename T, bool SILU>
__global__ void gn_bwd_gpartial_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy, int HW,
                                       int C, int G, int VX, int PY, int ppc, int nchunk,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ stats, float* __restrict__ gpartial) {
  extern __shared__ float red[];
  const int cg = C / G;
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  float ga[8], be[8], mu[8], rs[8], s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vx * 8 + e, gi = c / cg;
    ga[e] = gamma[c]; be[e] = beta[c];
    mu[e] = stats[((long)b * G + gi) * 2]; rs[e] = stats[((long)b * G + gi) * 2 + 1];
    s[e] = 0.f; q[e] = 0.f;
  }
#pragma unroll 4
  for (int p = p0 + py; p < p1; p += PY) {
    float f[8], d[8];
    const long row = (long)b * HW + p;
    load8(x + row * ldx + vx * 8, f);
    load8(dy + row * lddy + vx * 8, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (f[e] - mu[e]) * rs[e];
      float dz = d[e];
      if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
      s[e] += dz; q[e] += dz * xh;
    }
  }
  float* r = red + (py * VX + vx) * 16;
#pragma unroll
  for (int e = 0; e < 8; ++e) { r[e] = s[e]; r[8 + e] = q[e]; }
  __syncthreads();
  if (py == 0) {
    for (int k = 1; k < PY; ++k) {
      const float* o = red + (k * VX + vx) * 16;
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += o[e]; q[e] += o[8 + e]; }
    }
  }
  __syncthreads();
  if (py == 0) {
    float* ch = red + vx * 16;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ch[2 * e] = ga[e] * s[e]; ch[2 * e + 1] = ga[e] * q[e]; }
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int c0 = threadIdx.x * cg;
    float S = 0.f, Q = 0.f;
    for (int c = c0; c < c0 + cg; ++c) { S += red[2 * c]; Q += red[2 * c + 1]; }
    float* dst = gpartial + (((long)b * nchunk + chunk) * G + threadIdx.x) * 2;
    dst[0] = S; dst[1] = Q;
  }
}

template <typename T, bool SILU>
__global__ void gn_bwd_gapply_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                     const T* __restrict__ accum, long ldacc, T* __restrict__ dx, long lddx, int HW,
                                     int C, int G, int VX, int PY, int ppc, int nchunk, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, const float* __restrict__ stats,
                                     const float* __restrict__ gpartial) {
  __shared__ double part[8][64][2];
  __shared__ double tot[64][2];
  const int b = blockIdx.y, chunk = blockIdx.x, cg = C / G;
  gn_group_totals(gpartial, b, nchunk, G, part, tot);
  const double n = (double)HW * cg;
  const int vx = threadIdx.x % VX, py = threadIdx.x / VX;
  const int p0 = chunk * ppc, p1 = min(HW, p0 + ppc);
  float ga[8], be[8], mu[8], rs[8], k1[8], k2[8], k3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vx * 8 + e, gi = c / cg;
    ga[e] = gamma[c]; be[e] = beta[c];
    mu[e] = stats[((long)b * G + gi) * 2]; rs[e] = stats[((long)b * G + gi) * 2 + 1];
    const double rstd = rs[e];
    k1[e] = (float)(rstd * ga[e]);
    k2[e] = (float)(rstd * tot[gi][0] / n);
    k3[e] = (float)(rstd * tot[gi][1] / n);
  }
#pragma unroll 4
  for (int p = p0 + py; p < p1; p += PY) {
    float f[8], d[8], ac[8];
    const long row = (long)b * HW + p;
    load8(x + row * ldx + vx * 8, f);
    load8(dy + row * lddy + vx * 8, d);
    if (accum) load8(accum + row * ldacc + vx * 8, ac);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (f[e] - mu[e]) * rs[e];
      float dz = d[e];
      if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
      float r = dz * k1[e] - k2[e] - xh * k3[e];
      if (accum) r += ac[e];
      f[e] = r;
    }
    store8(dx + row * lddx + vx * 8, f);
  }
}

template <typename T>
static int gn_bwd_t(const GnBwdArgs& a, hipStream_t st) {
  const Gn1Geom g1 = gn1_geom(a.B, a.HW, a.C, a.G, (int)sizeof(T), true);
  if (g1.ok) {
    if (a.silu) gn1_bwd_launch<T, true>(a, g1, st); else gn1_bwd_launch<T, false>(a, g1, st);
    CL_CHECK_LAUNCH();
    return CL_OK;
  }
  const GnGeom g = gn_geom(a.B, a.HW, a.C);
  if (!a.dgamma && gn_two_pass_ok(g, a.C, a.G)) {
    dim3 grid(g.nchunk, a.B);
    const int lds = std::max(g.threads * 64, a.C * 8);
#define GN_BWD_G(S)                                                                                                \
    hipLaunchKernelGGL((gn_bwd_gpartial_kernel<T, S>), grid, dim3(g.threads), lds, st, (const T*)a.x, a.ldx,        \
                       (const T*)a.dy, a.lddy, a.HW, a.C, a.G, g.VX, g.PY, g.ppc, g.nchunk, a.gamma, a.beta, a.stats, \
                       a.ws);                                                                                       \
    hipLaunchKernelGGL((gn_bwd_gapply_kernel<T, S>), grid, dim3(g.threads), 0, st, (const T*)a.x, a.ldx,            \
                       (const T*)a.dy, a.lddy, (const T*)a.accum, a.ldacc, (T*)a.dx, a.lddx, a.HW, a.C, a.G, g.VX,  \
                       g.PY, g.ppc, g.nchunk, a.gamma, a.beta, a.stats, a.ws);
    if (a.silu) { GN_BWD_G(true) } else { GN_BWD_G(false) }
#undef GN_BWD_G
    CL_CHECK_LAUNCH();
    return CL_OK;
  }
  float* partial = a.ws;
  float* bcoef = a.ws + (long)a.B * g.nchunk * a.C * 2;
  dim3 grid(g.nchunk, a.B);
#define GN_BWD_LAUNCH(S)                                                                                       \
  hipLaunchKernelGGL((gn_bwd_partial_kernel<T, S>), grid, dim3(g.threads), g.threads * 64, st, (const T*)a.x,   \
                     a.ldx, (const T*)a.dy, a.lddy, a.HW, a.C, a.G, g.VX, g.PY, g.ppc, g.nchunk, a.gamma,       \
                     a.beta, a.stats, partial);                                                                \
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(a.G, a.B), dim3(256), 0, st,                                    \
                     partial, g.nchunk, a.HW, a.C, a.G, a.gamma, a.stats, bcoef, a.dgamma, a.dbeta);            \
  hipLaunchKernelGGL((gn_bwd_apply_kernel<T, S>), grid, dim3(g.threads), 0, st, (const T*)a.x, a.ldx,           \
                     (const T*)a.dy, a.lddy, (const T*)a.accum, a.ldacc, (T*)a.dx, a.lddx, a.HW, a.C, a.G,      \
                     g.VX, g.PY, g.ppc, a.gamma, a.beta, a.stats, bcoef);
  if (a.silu) { GN_BWD_LAUNCH(true) } else { GN_BWD_LAUNCH(false) }
#undef GN_BWD_LAUNCH
  CL_CHECK_LAUNCH();
  return CL_OK;
}

int gn_bwd(const GnBwdArgs& a, int dtype, hipStream_t st) {
  if (a.C % 8 || a.C % a.G || a.ldx % 8 || a.lddy % 8 || a.lddx % 8 || a.C > 8192) return CL_EINVAL;
  if (a.C / 8 > 320 && (a.C / 8) % 320) return CL_EINVAL;
  if (a.accum && a.ldacc % 8) return CL_EINVAL;
  if ((a.dgamma == nullptr) != (a.dbeta == nullptr)) return CL_EINVAL;
  if (gnc_bwd(a, dtype, st) == CL_OK) return CL_OK;
  return dtype == CL_BF16 ? gn_bwd_t<bf16_t>(a, st) : gn_bwd_t<float>(a, st);
}

// ------------------------------------------------------------------ LayerNorm

static constexpr int LN_MAXV = 3;  // vectors of 8 per lane -> D <= 1536 (SD1.5: 320 / 640 / 1280)

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y,
                                                     long ldy, int M, int D, float eps,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta,
                                                     float* __restrict__ stats /*[M][2]*/) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D8 = D / 8;
  for (long row = (long)blockIdx.x * 4 + wave; row < M; row += (long)gridDim.x * 4) {
    float f[LN_MAXV][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      const int v = lane + 64 * k;
      if (v < D8) {
        load8(x + row * ldx + v * 8, f[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[k][e];
      }
    }
    const float mean = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      const int v = lane + 64 * k;
      if (v < D8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = f[k][e] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0 && stats) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
#pragma unroll
    for (int k = 0; k < LN_MAXV; ++k) {
      const int v = lane + 64 * k;
      if (v < D8) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f[k][e] - mean) * rstd * gamma[v * 8 + e] + beta[v * 8 + e];
        store8(y + row * ldy + v * 8, o);
      }
    }
  }
}

// (Round 3 tried RPI = 4 / 2 / 1 rows per wave in flight in the FORWARD as the backward does: measured slower -- 21.2 us
// instead of 14.7 us for the 42 MB of a 64x64 LayerNorm, 1.04 instead of 0.72 ms per step -- and removed:
// profiles/r03_final/ln_rows_negative.txt.  With one row per wave the grid already keeps 32 waves per CU streaming.)
// dx = rstd * (dyh - mean(dyh) - xh * mean(dyh * xh)),  dyh = dy * gamma ; optional (+ accum)
// WG: dgamma += sum_rows dy * xh ; dbeta += sum_rows dy  (per-lane column sums -> LDS across the 4 waves
// -> block partial row, or one fp32 atomic per column per workgroup)
// NV = 16-byte vectors per lane (D <= 512 NV); RPI = rows per wave iteration, all kept in registers between
// the statistics pass and the output pass.  A row is only D * 2 bytes per operand (40 of 64 lanes at D = 320),
// so one row per wave leaves the kernel bound by the load -> reduce -> store latency chain; RPI rows in flight
// per wave (4 / 2 / 1 at NV = 1 / 2 / 3, ~96 data registers each way) is what fills the memory pipeline.
// A row past the end is clamped for its loads, contributes nothing and is not stored.
template <typename T, bool WG, int NV, int RPI>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy,
                                                     long lddy, const T* __restrict__ accum, long ldacc,
                                                     T* __restrict__ dx, long lddx, int M, int D,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ stats,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ partial) {
  // per-wave column sums (plain 16-byte LDS stores; LDS float atomics measured ~20 us per workgroup here)
  __shared__ __attribute__((aligned(16))) float red[WG ? 4 * 2 * NV * 512 : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int D8 = D / 8;
  float gsum[WG ? NV : 1][8], bsum[WG ? NV : 1][8];
  if (WG) {
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) { gsum[k][e] = 0.f; bsum[k][e] = 0.f; }
  }
  float gam[NV][8];
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const int v = lane + 64 * k;
#pragma unroll
    for (int e = 0; e < 8; ++e) gam[k][e] = v < D8 ? gamma[v * 8 + e] : 0.f;
  }
  const long rstride = (long)gridDim.x * 4;
  for (long row0 = (long)blockIdx.x * 4 + wave; row0 < M; row0 += RPI * rstride) {
    long rows[RPI];
    bool ok[RPI];
    float mean[RPI], rstd[RPI], c1[RPI], c2[RPI];
    float f[RPI][NV][8], d[RPI][NV][8], o[RPI][NV][8];
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      rows[r] = row0 + r * rstride;
      ok[r] = rows[r] < M;
      if (!ok[r]) rows[r] = row0;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lane + 64 * k;
        if (v < D8) {
          load8(x + rows[r] * ldx + v * 8, f[r][k]);
          load8(dy + rows[r] * lddy + v * 8, d[r][k]);
          if (accum) load8(accum + rows[r] * ldacc + v * 8, o[r][k]);
        }
      }
      mean[r] = stats[rows[r] * 2]; rstd[r] = stats[rows[r] * 2 + 1];
      c1[r] = 0.f; c2[r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const float live = ok[r] ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lane + 64 * k;
        if (v < D8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xh = (f[r][k][e] - mean[r]) * rstd[r];
            const float de = d[r][k][e] * live;
            if (WG) { gsum[k][e] += de * xh; bsum[k][e] += de; }
            const float dh = de * gam[k][e];
            c1[r] += dh; c2[r] += dh * xh;
            f[r][k][e] = xh; d[r][k][e] = dh;
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RPI; ++r) { c1[r] = wave_sum(c1[r]) / D; c2[r] = wave_sum(c2[r]) / D; }
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      if (!ok[r]) continue;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        const int v = lane + 64 * k;
        if (v < D8) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float g = rstd[r] * (d[r][k][e] - c1[r] - f[r][k][e] * c2[r]);
            o[r][k][e] = accum ? o[r][k][e] + g : g;
          }
          store8(dx + rows[r] * lddx + v * 8, o[r][k]);
        }
      }
    }
  }
  if (WG) {
    constexpr int WS = 2 * NV * 512;   // floats per wave: [dgamma NV*512 | dbeta NV*512]
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int v = lane + 64 * k;
      if (v < D8) {
        float4* pg = reinterpret_cast<float4*>(&red[wave * WS + v * 8]);
        float4* pb = reinterpret_cast<float4*>(&red[wave * WS + NV * 512 + v * 8]);
        pg[0] = make_float4(gsum[k][0], gsum[k][1], gsum[k][2], gsum[k][3]);
        pg[1] = make_float4(gsum[k][4], gsum[k][5], gsum[k][6], gsum[k][7]);
        pb[0] = make_float4(bsum[k][0], bsum[k][1], bsum[k][2], bsum[k][3]);
        pb[1] = make_float4(bsum[k][4], bsum[k][5], bsum[k][6], bsum[k][7]);
      }
    }
    __syncthreads();
    float* dst = partial ? partial + (long)blockIdx.x * 2 * D : nullptr;
    for (int c = threadIdx.x; c < D; c += 256) {
      const float g = (red[c] + red[WS + c]) + (red[2 * WS + c] + red[3 * WS + c]);
      const float b = (red[NV * 512 + c] + red[WS + NV * 512 + c]) +
                      (red[2 * WS + NV * 512 + c] + red[3 * WS + NV * 512 + c]);
      if (dst) { dst[c] = g; dst[D + c] = b; }   // block partials [grid][2 D]; ln_bwd_finish_kernel adds them up
      else { atomicAdd(dgamma + c, g); atomicAdd(dbeta + c, b); }
    }
  }
}

// dgamma[c] += sum_blocks partial[blk][c] ; dbeta[c] += sum_blocks partial[blk][D + c]
// 32 columns x 32 block lanes per 1024-thread workgroup (few columns: the depth of each lane's loop is what counts)
__global__ __launch_bounds__(1024) void ln_bwd_finish_kernel(const float* __restrict__ partial, int nblk, int D,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float red[1024];
  const int cl = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;   // column of the [2 D] partial row
  float s = 0.f;
  if (c < 2 * D)
#pragma unroll 8
    for (int k = kl; k < nblk; k += 32) s += partial[(long)k * 2 * D + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (kl == 0 && c < 2 * D) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) t += red[k * 32 + cl];
    if (c < D) dgamma[c] += t; else dbeta[c - D] += t;
  }
}

static int ln_grid(int M) {
  int g = (M + 3) / 4;
  return g < 2048 ? g : 2048;
}

int ln_fwd(const LnArgs& a, int dtype, hipStream_t st) {
  if (a.D % 8 || a.D > 64 * LN_MAXV * 8 || a.ldx % 8 || a.ldy % 8) return CL_EINVAL;
  if (dtype == CL_BF16)
    hipLaunchKernelGGL((ln_fwd_kernel<bf16_t>), dim3(ln_grid(a.M)), dim3(256), 0, st, (const bf16_t*)a.x, a.ldx,
                       (bf16_t*)a.y, a.ldy, a.M, a.D, a.eps, a.gamma, a.beta, a.stats);
  else
    hipLaunchKernelGGL((ln_fwd_kernel<float>), dim3(ln_grid(a.M)), dim3(256), 0, st, (const float*)a.x, a.ldx,
                       (float*)a.y, a.ldy, a.M, a.D, a.eps, a.gamma, a.beta, a.stats);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

int ln_bwd(const LnBwdArgs& a, int dtype, hipStream_t st) {
  if (a.D % 8 || a.D > 64 * LN_MAXV * 8 || a.ldx % 8 || a.lddy % 8 || a.lddx % 8) return CL_EINVAL;
  if ((a.dgamma == nullptr) != (a.dbeta == nullptr)) return CL_EINVAL;
  const int rpi = a.D <= 512 ? 4 : a.D <= 1024 ? 2 : 1;   // rows per wave iteration (see ln_bwd_kernel)
  int grid = (a.M + 4 * rpi - 1) / (4 * rpi);
  if (grid > 2048) grid = 2048;
  float* partial = nullptr;
  if (a.dgamma) {
    // column sums: block partials through the stream's registered scratch + a finishing kernel when there is
    // one (grid 1024); otherwise fp32 atomics with the grid bounded to 512 (same-address contention)
    void* wsp = nullptr; long wsb = 0;
    gemm_get_workspace_for(st, &wsp, &wsb);
    if (grid > 1024) grid = 1024;
    if (wsp && (long)grid * 2 * a.D * 4 <= wsb) partial = (float*)wsp;
    else if (grid > 512) grid = 512;
  }
#define LN_BWD_LAUNCH2(TT, WG, NV, RPI)                                                                         \
  hipLaunchKernelGGL((ln_bwd_kernel<TT, WG, NV, RPI>), dim3(grid), dim3(256), 0, st, (const TT*)a.x, a.ldx,     \
                     (const TT*)a.dy, a.lddy, (const TT*)a.accum, a.ldacc, (TT*)a.dx, a.lddx, a.M, a.D, a.gamma, \
                     a.stats, a.dgamma, a.dbeta, partial)
#define LN_BWD_LAUNCH(TT, WG)                                                                                   \
  do {                                                                                                          \
    if (a.D <= 512) LN_BWD_LAUNCH2(TT, WG, 1, 4);                                                               \
    else if (a.D <= 1024) LN_BWD_LAUNCH2(TT, WG, 2, 2);                                                         \
    else LN_BWD_LAUNCH2(TT, WG, 3, 1);                                                                          \
  } while (0)
  if (dtype == CL_BF16) { if (a.dgamma) LN_BWD_LAUNCH(bf16_t, true); else LN_BWD_LAUNCH(bf16_t, false); }
  else { if (a.dgamma) LN_BWD_LAUNCH(float, true); else LN_BWD_LAUNCH(float, false); }
#undef LN_BWD_LAUNCH2
#undef LN_BWD_LAUNCH
  if (partial)
    hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((2 * a.D + 31) / 32), dim3(1024), 0, st, partial, grid, a.D,
                       a.dgamma, a.dbeta);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

}  // namespace cl
