// Fused attention backward for gfx950 (data-gradient of CrossAttention.forward,
// ldm/modules/attention.py:163-194), recomputing P from the saved log-sum-exp.
//
//   delta[q]  = sum_d dO[q,d] O[q,d]
//   dKV kernel (one 64-key block per workgroup, 16 keys per wave, loop over query tiles):
//     S  = Q K^T   -> P = exp2(S*c - LSE[q])        C layout: col = key (lane&15), rows = queries
//     dP = dO V^T  -> dS = P o (dP - delta[q]) * scale
//     dV^T[d][kv] += dO^T . P      dK^T[d][kv] += Q^T . dS     (P / dS feed the MFMA B operand
//                                                               directly from their accumulators)
//   dQ kernel (64 queries per workgroup, 16 per wave, loop over key tiles), S^T form as in forward:
//     S^T = K Q^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta[q]) * scale,  dQ^T[d][q] += K^T . dS^T
//
// The second-stage A operands need the contraction index contiguous, i.e. Q^T, dO^T (over
// queries) and K^T (over keys); those [B][inner][n_pad] copies are produced by the batched
// transpose kernel (HBM-bound, a few % of the attention time).  Tiles are staged with
// global_load_lds, single buffered.
#include "attn_common.h"

namespace cl {

// ---------------------------------------------------------------- delta
template <typename T>
__global__ void attn_delta_kernel(const T* __restrict__ O, long ldo, const T* __restrict__ dO, long lddo,
                                  float* __restrict__ delta, int lse_stride, int B, int H, int N, int DH) {
  const long total = (long)B * H * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % N); const long bh = i / N; const int h = (int)(bh % H); const int b = (int)(bh / H);
    const T* o = O + ((long)b * N + q) * ldo + (long)h * DH;
    const T* d = dO + ((long)b * N + q) * lddo + (long)h * DH;
    float acc = 0.f;
    for (int e = 0; e < DH; e += 8) {
      float a[8], c[8]; load8(o + e, a); load8(d + e, c);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += a[k] * c[k];
    }
    delta[bh * lse_stride + q] = acc;
  }
}

// ---------------------------------------------------------------- dK / dV
template <typename T, int DH, int BQ>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnBwdArgs p) {
  constexpr int EB = AttnTraits<T>::EB;
  constexpr int CPR = DH * EB / 16, KSTEPS = (CPR + 3) / 4;
  constexpr int QF = BQ / 16, DN = (DH + 15) / 16;
  constexpr int TROW = BQ * EB, TCPR = TROW / 16;         // transposed tile rows (queries contiguous)
  constexpr int RT_BYTES = BQ * CPR * 16;                  // row-major Q / dO tile
  constexpr int TT_BYTES = DN * 16 * TROW;                 // Q^T / dO^T tile
  constexpr int RI = BQ * CPR / 64, TI = DH * TCPR / 64;   // glds instructions
  constexpr int PF = PFrag<T>::FRAGS, PSTEPS = QF / PF;
  static_assert((BQ * CPR) % 64 == 0 && (DH * TCPR) % 64 == 0, "tile must be whole glds instructions");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sQ = smem; char* const sdO = sQ + RT_BYTES; char* const sQt = sdO + RT_BYTES; char* const sdOt = sQt + TT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lq = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int kv_w = blockIdx.x * 64 + wave * 16;   // this wave's 16 keys
  const float sl2 = p.scale * 1.4426950408889634f;

  // K and V fragments (B operands: col = key lq, k = d chunk) live in registers for the whole loop
  u32x4_t kb[KSTEPS], vb[KSTEPS];
  {
    const int kr = min(kv_w + lq, p.Nkv - 1);
    const char* kp = (const char*)p.K + (((long)b * p.Nkv + kr) * p.ldk + (long)h * DH) * EB;
    const char* vp = (const char*)p.V + (((long)b * p.Nkv + kr) * p.ldv + (long)h * DH) * EB;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = 4 * ks + g;
      kb[ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(kp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
      vb[ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(vp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
  }
  const char* qbase = (const char*)p.Q + ((long)h * DH) * EB;
  const char* dobase = (const char*)p.dO + ((long)h * DH) * EB;
  const char* qtbase = (const char*)p.Qt + (((long)b * p.H + h) * DH) * (long)p.n_pad * EB;
  const char* dotbase = (const char*)p.dOt + (((long)b * p.H + h) * DH) * (long)p.n_pad * EB;
  const float* lse = p.LSE + ((long)b * p.H + h) * p.lse_stride;
  const float* dlt = p.Delta + ((long)b * p.H + h) * p.lse_stride;

  f32x4_t dvt[DN], dkt[DN];
#pragma unroll
  for (int i = 0; i < DN; ++i) { dvt[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dkt[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t aQ = lds0, adO = aQ + RT_BYTES, aQt = adO + RT_BYTES, adOt = aQt + TT_BYTES;
  const int ntiles = (p.N + BQ - 1) / BQ;
  for (int t = 0; t < ntiles; ++t) {
    const int q0 = t * BQ;
    __syncthreads();  // previous tile fully consumed
    for (int ii = wave; ii < 2 * RI + 2 * TI; ii += 4) {
      if (ii < 2 * RI) {
        const bool second = ii >= RI; const int i2 = second ? ii - RI : ii;
        const int q = i2 * 64 + lane; const int r = q / CPR, c = q - r * CPR;
        const int qr = min(q0 + r, p.N - 1);
        if (!second) glds16(qbase + (((long)b * p.N + qr) * p.ldq) * EB + c * 16, sQ + i2 * 1024);
        else glds16(dobase + (((long)b * p.N + qr) * p.lddo) * EB + c * 16, sdO + i2 * 1024);
      } else {
        const int i3 = ii - 2 * RI; const bool second = i3 >= TI; const int i2 = second ? i3 - TI : i3;
        const int q = i2 * 64 + lane; const int d = q / TCPR, c = (q - d * TCPR) ^ tile_swz<TROW>(d);
        if (!second) glds16(qtbase + ((long)d * p.n_pad + q0) * EB + c * 16, sQt + i2 * 1024);
        else glds16(dotbase + ((long)d * p.n_pad + q0) * EB + c * 16, sdOt + i2 * 1024);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- S = Q K^T and dP = dO V^T  (rows = queries 16*qf + 4g + r, col = key lq)
    f32x4_t ps[QF], ds[QF];
#pragma unroll
    for (int qf = 0; qf < QF; ++qf) {
      u32x4_t qa[KSTEPS], da[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int c = 4 * ks + g;
        qa[ks] = (c < CPR) ? lds_read_b128(aQ + ((qf * 16 + lq) * CPR + c) * 16) : u32x4_t{0u, 0u, 0u, 0u};
        da[ks] = (c < CPR) ? lds_read_b128(adO + ((qf * 16 + lq) * CPR + c) * 16) : u32x4_t{0u, 0u, 0u, 0u};
      }
      lds_wait();
      f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) { Mma<T>::run(qa[ks], kb[ks], s); Mma<T>::run(da[ks], vb[ks], dp); }
      const int qrow = q0 + qf * 16 + 4 * g;
      const float4 l4 = *reinterpret_cast<const float4*>(lse + qrow);
      const float4 d4 = *reinterpret_cast<const float4*>(dlt + qrow);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = qrow + r < p.N;
        const float pr = ok ? __builtin_amdgcn_exp2f(s[r] * sl2 - lv[r]) : 0.f;
        ps[qf][r] = pr;
        ds[qf][r] = ok ? pr * (dp[r] - dv[r]) * p.scale : 0.f;
      }
    }
    // ---- dV^T += dO^T . P ;  dK^T += Q^T . dS
    u32x4_t pb[PSTEPS], sb[PSTEPS];
#pragma unroll
    for (int s = 0; s < PSTEPS; ++s) { pb[s] = PFrag<T>::make(&ps[s * PF]); sb[s] = PFrag<T>::make(&ds[s * PF]); }
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      u32x4_t oa[PSTEPS], qa[PSTEPS];
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s) {
        oa[s] = PFrag<T>::read_a(adOt + (i * 16 + lq) * TROW, s, g, tile_swz<TROW>(lq));
        qa[s] = PFrag<T>::read_a(aQt + (i * 16 + lq) * TROW, s, g, tile_swz<TROW>(lq));
      }
      lds_wait();
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s) { Mma<T>::run(oa[s], pb[s], dvt[i]); Mma<T>::run(qa[s], sb[s], dkt[i]); }
    }
  }
  // ---- store dK / dV rows (key kv_w + lq, 4 consecutive d per lane)
  const int kr = kv_w + lq;
  if (kr < p.Nkv) {
    T* dkp = reinterpret_cast<T*>(p.dK) + ((long)b * p.Nkv + kr) * p.lddk + (long)h * DH;
    T* dvp = reinterpret_cast<T*>(p.dV) + ((long)b * p.Nkv + kr) * p.lddv + (long)h * DH;
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      const int d0 = i * 16 + 4 * g;
      if (d0 < DH) {
        float a[4] = {dkt[i][0], dkt[i][1], dkt[i][2], dkt[i][3]};
        float c[4] = {dvt[i][0], dvt[i][1], dvt[i][2], dvt[i][3]};
        store4(dkp + d0, a);
        store4(dvp + d0, c);
      }
    }
  }
}

// ---------------------------------------------------------------- dQ
template <typename T, int DH, int BKV>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdArgs p) {
  constexpr int EB = AttnTraits<T>::EB;
  constexpr int CPR = DH * EB / 16, KSTEPS = (CPR + 3) / 4;
  constexpr int KVF = BKV / 16, DN = (DH + 15) / 16;
  constexpr int TROW = BKV * EB, TCPR = TROW / 16;
  constexpr int RT_BYTES = BKV * CPR * 16, TT_BYTES = DN * 16 * TROW;
  constexpr int RI = BKV * CPR / 64, TI = DH * TCPR / 64;
  constexpr int PF = PFrag<T>::FRAGS, PSTEPS = KVF / PF;
  static_assert((BKV * CPR) % 64 == 0 && (DH * TCPR) % 64 == 0, "tile must be whole glds instructions");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sK = smem; char* const sV = sK + RT_BYTES; char* const sKt = sV + RT_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lq = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q_w = blockIdx.x * 64 + wave * 16;
  const float sl2 = p.scale * 1.4426950408889634f;
  const int qrow = q_w + lq;
  const int qr = min(qrow, p.N - 1);

  u32x4_t qb[KSTEPS], ob[KSTEPS];
  {
    const char* qp = (const char*)p.Q + (((long)b * p.N + qr) * p.ldq + (long)h * DH) * EB;
    const char* op = (const char*)p.dO + (((long)b * p.N + qr) * p.lddo + (long)h * DH) * EB;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = 4 * ks + g;
      qb[ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(qp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
      ob[ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(op + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
  }
  const float lse_q = p.LSE[((long)b * p.H + h) * p.lse_stride + qr];
  const float dlt_q = p.Delta[((long)b * p.H + h) * p.lse_stride + qr];
  const char* kbase = (const char*)p.K + ((long)h * DH) * EB;
  const char* vbase = (const char*)p.V + ((long)h * DH) * EB;
  const char* ktbase = (const char*)p.Kt + (((long)b * p.H + h) * DH) * (long)p.nkv_pad * EB;

  f32x4_t dqt[DN];
#pragma unroll
  for (int i = 0; i < DN; ++i) dqt[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t aK = lds0, aV = aK + RT_BYTES, aKt = aV + RT_BYTES;
  const int ntiles = (p.Nkv + BKV - 1) / BKV;
  for (int t = 0; t < ntiles; ++t) {
    const int kv0 = t * BKV;
    __syncthreads();
    for (int ii = wave; ii < 2 * RI + TI; ii += 4) {
      if (ii < 2 * RI) {
        const bool second = ii >= RI; const int i2 = second ? ii - RI : ii;
        const int q = i2 * 64 + lane; const int r = q / CPR, c = q - r * CPR;
        const int kr = min(kv0 + r, p.Nkv - 1);
        if (!second) glds16(kbase + (((long)b * p.Nkv + kr) * p.ldk) * EB + c * 16, sK + i2 * 1024);
        else glds16(vbase + (((long)b * p.Nkv + kr) * p.ldv) * EB + c * 16, sV + i2 * 1024);
      } else {
        const int i2 = ii - 2 * RI;
        const int q = i2 * 64 + lane; const int d = q / TCPR, c = (q - d * TCPR) ^ tile_swz<TROW>(d);
        glds16(ktbase + ((long)d * p.nkv_pad + kv0) * EB + c * 16, sKt + i2 * 1024);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4_t dst[KVF];
#pragma unroll
    for (int kf = 0; kf < KVF; ++kf) {
      u32x4_t ka[KSTEPS], va[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int c = 4 * ks + g;
        ka[ks] = (c < CPR) ? lds_read_b128(aK + ((kf * 16 + lq) * CPR + c) * 16) : u32x4_t{0u, 0u, 0u, 0u};
        va[ks] = (c < CPR) ? lds_read_b128(aV + ((kf * 16 + lq) * CPR + c) * 16) : u32x4_t{0u, 0u, 0u, 0u};
      }
      lds_wait();
      f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) { Mma<T>::run(ka[ks], qb[ks], s); Mma<T>::run(va[ks], ob[ks], dp); }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool ok = kv0 + kf * 16 + 4 * g + r < p.Nkv;
        const float pr = ok ? __builtin_amdgcn_exp2f(s[r] * sl2 - lse_q) : 0.f;
        dst[kf][r] = ok ? pr * (dp[r] - dlt_q) * p.scale : 0.f;
      }
    }
    u32x4_t sb[PSTEPS];
#pragma unroll
    for (int s = 0; s < PSTEPS; ++s) sb[s] = PFrag<T>::make(&dst[s * PF]);
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      u32x4_t ka[PSTEPS];
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s) ka[s] = PFrag<T>::read_a(aKt + (i * 16 + lq) * TROW, s, g, tile_swz<TROW>(lq));
      lds_wait();
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s) Mma<T>::run(ka[s], sb[s], dqt[i]);
    }
  }
  if (qrow < p.N) {
    T* dqp = reinterpret_cast<T*>(p.dQ) + ((long)b * p.N + qrow) * p.lddq + (long)h * DH;
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      const int d0 = i * 16 + 4 * g;
      if (d0 < DH) {
        float a[4] = {dqt[i][0], dqt[i][1], dqt[i][2], dqt[i][3]};
        store4(dqp + d0, a);
      }
    }
  }
}

// ---------------------------------------------------------------- host side
template <typename T, int DH>
static int launch_bwd(const AttnBwdArgs& a, hipStream_t st) {
  constexpr int EB = AttnTraits<T>::EB;
  constexpr int BT = (EB == 4 && DH >= 80) ? 32 : 64;   // tile length along the looped dimension
  constexpr int CPR = DH * EB / 16, DN = (DH + 15) / 16;
  constexpr int LDS_DKV = 2 * BT * CPR * 16 + 2 * DN * 16 * BT * EB;
  constexpr int LDS_DQ = 2 * BT * CPR * 16 + DN * 16 * BT * EB;
  static bool attr_set = false;
  if (!attr_set) {
    if (LDS_DKV > 65536 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<T, DH, BT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV) != hipSuccess)
      return CL_ELAUNCH;
    if (LDS_DQ > 65536 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<T, DH, BT>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ) != hipSuccess)
      return CL_ELAUNCH;
    attr_set = true;
  }
  const long total = (long)a.B * a.H * a.N;
  int dgrid = (int)((total + 255) / 256); if (dgrid > 2048) dgrid = 2048;
  hipLaunchKernelGGL((attn_delta_kernel<T>), dim3(dgrid), dim3(256), 0, st, (const T*)a.O, a.ldo, (const T*)a.dO,
                     a.lddo, a.Delta, a.lse_stride, a.B, a.H, a.N, a.DH);
  if (a.dK) {
    dim3 grid((a.Nkv + 63) / 64, a.H, a.B);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, DH, BT>), grid, dim3(256), LDS_DKV, st, a);
  }
  dim3 gridq((a.N + 63) / 64, a.H, a.B);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, DH, BT>), gridq, dim3(256), LDS_DQ, st, a);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

template <typename T>
static int dispatch_bwd(const AttnBwdArgs& a, hipStream_t st) {
  switch (a.DH) {
    case 8: return launch_bwd<T, 8>(a, st);
    case 16: return launch_bwd<T, 16>(a, st);
    case 32: return launch_bwd<T, 32>(a, st);
    case 40: return launch_bwd<T, 40>(a, st);
    case 80: return launch_bwd<T, 80>(a, st);
    case 160: return launch_bwd<T, 160>(a, st);
    default: return CL_EINVAL;
  }
}

int attn_delta(const AttnBwdArgs& a, hipStream_t st) {
  const long total = (long)a.B * a.H * a.N;
  int dgrid = (int)((total + 255) / 256); if (dgrid > 2048) dgrid = 2048;
  hipLaunchKernelGGL((attn_delta_kernel<bf16_t>), dim3(dgrid), dim3(256), 0, st, (const bf16_t*)a.O, a.ldo,
                     (const bf16_t*)a.dO, a.lddo, a.Delta, a.lse_stride, a.B, a.H, a.N, a.DH);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

int attn_bwd(const AttnBwdArgs& a, int dtype, hipStream_t st) {
  if (a.q_prescaled) return CL_EINVAL;   // the pre-scaled-Q contract is the transpose-free bf16 kernels' (attention_tr.hip)
  const int eb = dtype == CL_BF16 ? 2 : 4;
  if ((a.ldq * eb) % 16 || (a.ldk * eb) % 16 || (a.ldv * eb) % 16 || (a.lddo * eb) % 16 || (a.ldo * eb) % 16)
    return CL_EINVAL;
  if ((a.lddq * eb) % 16 || a.n_pad % 64 || a.nkv_pad % 64 || a.n_pad < a.N || a.nkv_pad < a.Nkv ||
      a.lse_stride % 64 || a.lse_stride < a.N)
    return CL_EINVAL;
  if ((a.dK == nullptr) != (a.dV == nullptr)) return CL_EINVAL;
  if (a.dK && ((a.lddk * eb) % 16 || (a.lddv * eb) % 16)) return CL_EINVAL;
  return dtype == CL_BF16 ? dispatch_bwd<bf16_t>(a, st) : dispatch_bwd<float>(a, st);
}

}  // namespace cl
