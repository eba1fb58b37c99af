// Fused (flash-style) attention forward for gfx950: CrossAttention.forward
// (ldm/modules/attention.py:163-194) without materialising the (B*8, N, Nkv)
// score matrix the reference writes to HBM.
//
//   per (b, h, 64*QW query rows): loop over BKV-key tiles
//     S^T[kv][q] = K . Q^T          (MFMA; A = K rows from LDS, B = Q rows held in registers)
//     online softmax over kv, fp32 (the reference forces fp32 for QK^T/softmax, :171-179);
//       in the S^T layout every lane owns ONE query (q = lane & 15), so running max / sum /
//       rescale factor are per-lane scalars: the row reduction is two shuffles, no LDS
//     O^T[d][q] += V^T . P^T        (MFMA; A = V^T rows from LDS, B = P straight from the S^T
//                                    accumulator registers -- no layout change needed)
//
// K and V^T tiles are staged HBM->LDS with global_load_lds (double buffered, one
// barrier per tile).  d_head 40/80/160 are handled without padding in LDS: K
// rows are 80/160/320 bytes, and fragments past d_head are zeroed in registers.
// The same code runs in fp32 (parity mode) through Mma<float>.
#include "attn_common.h"

namespace cl {

template <typename T, int DH, int QW, int BKV>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnFwdArgs p) {
  constexpr int EB = AttnTraits<T>::EB;
  constexpr int CPR = DH * EB / 16;          // 16-byte chunks per K/Q row
  constexpr int KSTEPS = (CPR + 3) / 4;      // 64-byte K steps over d_head
  constexpr int KVF = BKV / 16;              // kv fragments per tile
  constexpr int DN = (DH + 15) / 16;         // d fragments of the output
  constexpr int VROW = BKV * EB;             // bytes per V^T tile row
  constexpr int VCPR = VROW / 16;
  constexpr int KT_BYTES = BKV * CPR * 16;
  constexpr int VT_BYTES = DN * 16 * VROW;
  constexpr int STAGE = KT_BYTES + VT_BYTES;
  constexpr int KI = BKV * CPR / 64, VI = DH * VCPR / 64;  // glds instructions per tile
  constexpr int PF = PFrag<T>::FRAGS;
  constexpr int PSTEPS = KVF / PF;
  static_assert((BKV * CPR) % 64 == 0 && (DH * VCPR) % 64 == 0, "tile must be whole glds instructions");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, lq = lane & 15;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (64 * QW) + wave * (16 * QW);
  const float sl2 = p.scale * 1.4426950408889634f;

  // ---- Q fragments (B operand of S^T = K.Q^T), straight from HBM into registers
  u32x4_t qf[QW][KSTEPS];
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    int row = q0 + f * 16 + lq;
    row = min(row, p.N - 1);
    const char* qp = (const char*)p.Q + (((long)b * p.N + row) * p.ldq + (long)h * DH) * EB;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int c = 4 * ks + g;
      qf[f][ks] = (c < CPR) ? *reinterpret_cast<const u32x4_t*>(qp + c * 16) : u32x4_t{0u, 0u, 0u, 0u};
    }
  }

  const char* kbase = (const char*)p.K + ((long)h * DH) * EB;
  const char* vbase = (const char*)p.Vt + (((long)b * p.H + h) * DH) * (long)p.nkv_pad * EB;

  auto issue = [&](int tile, int buf) {
    char* kt = smem + buf * STAGE;
    char* vt = kt + KT_BYTES;
    const int kv0 = tile * BKV;
    for (int ii = wave; ii < KI + VI; ii += 4) {
      if (ii < KI) {
        const int q = ii * 64 + lane;
        const int r = q / CPR, c = q - r * CPR;
        const int kr = min(kv0 + r, p.Nkv - 1);
        glds16(kbase + (((long)b * p.Nkv + kr) * p.ldk) * EB + c * 16, kt + ii * 1024);
      } else {
        const int q = (ii - KI) * 64 + lane;
        const int d = q / VCPR, c = (q - d * VCPR) ^ tile_swz<VROW>(d);
        glds16(vbase + ((long)d * p.nkv_pad + kv0) * EB + c * 16, vt + (ii - KI) * 1024);
      }
    }
  };

  f32x4_t ot[DN][QW];
#pragma unroll
  for (int i = 0; i < DN; ++i)
#pragma unroll
    for (int f = 0; f < QW; ++f) ot[i][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run[QW], l_run[QW];
#pragma unroll
  for (int f = 0; f < QW; ++f) { m_run[f] = -1e30f; l_run[f] = 0.f; }

  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int ntiles = (p.Nkv + BKV - 1) / BKV;
  issue(0, 0);
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 < ntiles) issue(t + 1, buf ^ 1);
    const uint32_t kt = lds0 + buf * STAGE, vt = kt + KT_BYTES;

    // ---- S^T = K . Q^T
    f32x4_t st[KVF][QW];
#pragma unroll
    for (int kf = 0; kf < KVF; ++kf) {
      u32x4_t ka[KSTEPS];
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int c = 4 * ks + g;
        ka[ks] = (c < CPR) ? lds_read_b128(kt + ((kf * 16 + lq) * CPR + c) * 16) : u32x4_t{0u, 0u, 0u, 0u};
      }
      lds_wait();
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        st[kf][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) Mma<T>::run(ka[ks], qf[f][ks], st[kf][f]);
      }
    }

    // ---- online softmax (lane owns query lq of each q fragment; keys 16*kf + 4*g + r)
    const int kv0 = t * BKV;
    const bool tail = kv0 + BKV > p.Nkv;
#pragma unroll
    for (int f = 0; f < QW; ++f) {
      float mx = -1e30f;
#pragma unroll
      for (int kf = 0; kf < KVF; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = st[kf][f][r] * sl2;
          if (tail && kv0 + kf * 16 + 4 * g + r >= p.Nkv) s = -INFINITY;
          st[kf][f][r] = s;
          mx = fmaxf(mx, s);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run[f], mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);
      m_run[f] = m_new;
      float ls = 0.f;
#pragma unroll
      for (int kf = 0; kf < KVF; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(st[kf][f][r] - m_new);
          st[kf][f][r] = e;
          ls += e;
        }
      l_run[f] = l_run[f] * alpha + ls;
#pragma unroll
      for (int i = 0; i < DN; ++i) ot[i][f] *= alpha;
    }

    // ---- O^T += V^T . P^T
    u32x4_t pb[PSTEPS][QW];
#pragma unroll
    for (int s = 0; s < PSTEPS; ++s)
#pragma unroll
      for (int f = 0; f < QW; ++f) {
        f32x4_t tmp[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) tmp[k] = st[s * PF + k][f];
        pb[s][f] = PFrag<T>::make(tmp);
      }
#pragma unroll
    for (int i = 0; i < DN; ++i) {
      u32x4_t va[PSTEPS];
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s) va[s] = PFrag<T>::read_a(vt + (i * 16 + lq) * VROW, s, g, tile_swz<VROW>(lq));
      lds_wait();
#pragma unroll
      for (int s = 0; s < PSTEPS; ++s)
#pragma unroll
        for (int f = 0; f < QW; ++f) Mma<T>::run(va[s], pb[s][f], ot[i][f]);
    }
  }

  // ---- epilogue: normalise, store O rows (4 consecutive d per lane), log-sum-exp
#pragma unroll
  for (int f = 0; f < QW; ++f) {
    float l = l_run[f];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    const int row = q0 + f * 16 + lq;
    if (row < p.N) {
      T* op = reinterpret_cast<T*>(p.O) + ((long)b * p.N + row) * p.ldo + (long)h * DH;
#pragma unroll
      for (int i = 0; i < DN; ++i) {
        const int d0 = i * 16 + 4 * g;
        if (d0 < DH) {
          float v[4] = {ot[i][f][0] * inv, ot[i][f][1] * inv, ot[i][f][2] * inv, ot[i][f][3] * inv};
          store4(op + d0, v);
        }
      }
      if (p.LSE && g == 0) p.LSE[((long)b * p.H + h) * p.lse_stride + row] = m_run[f] + __builtin_amdgcn_logf(l);
    }
  }
}

template <typename T, int DH, int QW, int BKV>
static int launch_fwd(const AttnFwdArgs& a, hipStream_t st) {
  constexpr int EB = AttnTraits<T>::EB;
  constexpr int CPR = DH * EB / 16, DN = (DH + 15) / 16;
  constexpr int LDS = 2 * (BKV * CPR * 16 + DN * 16 * BKV * EB);
  static bool attr_set = false;
  if (!attr_set) {
    if (LDS > 65536 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<T, DH, QW, BKV>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
      return CL_ELAUNCH;
    attr_set = true;
  }
  dim3 grid((a.N + 64 * QW - 1) / (64 * QW), a.H, a.B);
  hipLaunchKernelGGL((attn_fwd_kernel<T, DH, QW, BKV>), grid, dim3(256), LDS, st, a);
  CL_CHECK_LAUNCH();
  return CL_OK;
}

template <typename T, int DH>
static int dispatch_qw(const AttnFwdArgs& a, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    const long blocks128 = (long)((a.N + 127) / 128) * a.H * a.B;
    if (blocks128 >= 512) return launch_fwd<T, DH, 2, 64>(a, st);
    return launch_fwd<T, DH, 1, 64>(a, st);
  } else {
    return launch_fwd<T, DH, 1, (DH >= 80 ? 32 : 64)>(a, st);
  }
}

template <typename T>
static int dispatch_dh(const AttnFwdArgs& a, hipStream_t st) {
  switch (a.DH) {
    case 8: return dispatch_qw<T, 8>(a, st);
    case 16: return dispatch_qw<T, 16>(a, st);
    case 32: return dispatch_qw<T, 32>(a, st);
    case 40: return dispatch_qw<T, 40>(a, st);
    case 80: return dispatch_qw<T, 80>(a, st);
    case 160: return dispatch_qw<T, 160>(a, st);
    default: return CL_EINVAL;
  }
}

int attn_fwd(const AttnFwdArgs& a, int dtype, hipStream_t st) {
  if (a.q_prescaled) return CL_EINVAL;   // the pre-scaled-Q contract is the transpose-free bf16 kernels' (attention_tr.hip)
  const int eb = dtype == CL_BF16 ? 2 : 4;
  if ((a.ldq * eb) % 16 || (a.ldk * eb) % 16 || a.nkv_pad % 64 || a.nkv_pad < a.Nkv || a.Nkv < 1 || a.N < 1)
    return CL_EINVAL;
  if ((a.ldo * eb) % 16) return CL_EINVAL;
  return dtype == CL_BF16 ? dispatch_dh<bf16_t>(a, st) : dispatch_dh<float>(a, st);
}

}  // namespace cl
