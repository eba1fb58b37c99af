// GroupNorm(32)(+SiLU), forward and backward, as ONE launch and ONE pass over HBM at the levels whose groups span more pixels
// than a workgroup can hold (64x64, 32x32): the reference's normalisation (ldm/modules/diffusionmodules/util.py:217-219,
// openaimodel.py:201-203,225-229) reads x, reduces over (H, W, C / G), and applies -- the two-launch form of norm.hip does that
// as partial sums | apply, three passes over a tensor (read, read, write) and two dependent launches; the register-resident
// form of norm.hip (one workgroup per (sample, whole groups)) reads 80-byte pieces of 640-byte rows and runs on B * C / 40
// workgroups.  Here:
//
//   grid (S, B): workgroup (s, b) owns PIXEL SLAB s of sample b -- HW / S pixels x ALL C channels, whole rows, so every load and
//   store is a full-line 16-byte-per-lane stream -- and keeps it in registers (bf16-packed, NV 16-byte vectors per lane);
//   pass 1: per-channel partial sums of the slab -> per-group partial sums -> ws[b][s][g][2];
//   the S workgroups of a sample meet at a counter (all of them are resident: the launcher checks the grid against the
//   occupancy; the partials travel as agent-scope atomic stores / loads -- the XCDs' L2s are not coherent with each other);
//   pass 2: every workgroup sums the S partials in the same fixed order (bit-identical statistics in all of them, no atomics on
//   the data path), forms mean / rstd in fp64 exactly as norm.hip does, and normalises its slab out of registers.
//
// HBM traffic: the tensor once in, once out (backward: x and dy in, dx out).  The counters are self-resetting (the last
// workgroup to LEAVE zeroes them), live in device globals, and therefore assume the GroupNorms of a process are launched on one
// stream at a time (they are: cldm/ and ldm/ run the UNet on a single stream, captured graphs are linear chains).  A workgroup
// that polls 2^16 times without seeing its sample complete gives up, raises g_gnc_timeouts and proceeds with what it has --
// a wrong result that the parity tests see, instead of a hung GPU.
#include <algorithm>
#include "norm.h"
#include "gemm.h"

namespace cl {

// MEASURED (profiles/r06_gn/coop_vs_other_forms.txt, MI355X): this form is NOT faster than the two-launch form -- (32768, 320)
// forward 24.8 us vs 25.7, backward 40.9 vs 36.6; (8192, 640) forward 20.1 vs 18.0 -- because the four dependent trips to memory of
// the meeting (partials out, counter up, counter seen, partials in; ~2 us each across XCDs) cost what the saved second read of
// a 21 MB tensor costs at ~3 TB/s.  It is therefore OFF by default (cl_debug_groupnorm_coop(1) / CTRLORA_GN_COOP=1 turn it on;
// tests/test_gpu_groupnorm_coop.py keeps it correct).
int g_gn_coop = 0;

namespace {

constexpr int GNC_MAXB = 4096;
__device__ unsigned g_gnc_arrive[GNC_MAXB];
__device__ unsigned g_gnc_leave[GNC_MAXB];
__device__ unsigned g_gnc_timeouts;

struct PackB {                 // 8 bf16
  uint4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void get(float f[8]) const {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
  }
};

// The S workgroups of sample b meet.  What they exchange (the partial sums) is written and read with agent-scope atomic
// stores / loads -- write-through / cache-bypassing accesses -- so that no L2 write-back or invalidate is needed around the
// counter (a release / acquire FENCE pair here costs ~20 us: buffer_wbl2 writes back every dirty line of the XCD's L2, the
// previous kernel's output included).  The writers wait for their stores (gnc_stores_done) before the workgroup's barrier;
// thread 0 then arrives, polls, leaves.
__device__ __forceinline__ void gnc_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void gnc_sample_barrier(int b, int S) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&g_gnc_arrive[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int polls = 0;
    while (__hip_atomic_load(&g_gnc_arrive[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)S) {
      __builtin_amdgcn_s_sleep(2);
      if (++polls > (1 << 16)) { atomicAdd(&g_gnc_timeouts, 1u); break; }
    }
    asm volatile("" ::: "memory");
    // the last one to leave resets both counters (nobody is polling any more: everybody has left)
    const unsigned d = __hip_atomic_fetch_add(&g_gnc_leave[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == (unsigned)S - 1) {
      __hip_atomic_store(&g_gnc_leave[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&g_gnc_arrive[b], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
}

// n floats that other workgroups wrote (agent-scope stores) -> LDS, every thread four loads in flight at a time.  (A loop of
// __hip_atomic_load is compiled to one load, one wait, ... : 64 dependent trips to memory cost ~25 us.)
__device__ __forceinline__ void gnc_gather(const float* __restrict__ src, float* dst, int n) {
  for (int base = 0; base < n; base += 4 * blockDim.x) {
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + k * blockDim.x + threadIdx.x;
      const float* a = src + (i < n ? i : 0);
      asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[k]) : "v"(a) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = base + k * blockDim.x + threadIdx.x;
      if (i < n) dst[i] = v[k];
    }
  }
  __syncthreads();
}

// per-thread channel sums (thread = (py, vx): always the same 8 channels) -> per-channel sums of the workgroup in chs[C][2]
__device__ __forceinline__ void gnc_channel_sums(const float s[8], const float q[8], float* red /*[T][16]*/, float* chs /*[C][2]*/,
                                                 int C, int VX, int PY) {
  float* r = red + threadIdx.x * 16;
#pragma unroll
  for (int e = 0; e < 8; ++e) { r[e] = s[e]; r[8 + e] = q[e]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int vx = c >> 3, e = c & 7;
    float a = 0.f, b = 0.f;
    for (int py = 0; py < PY; ++py) { a += red[(py * VX + vx) * 16 + e]; b += red[(py * VX + vx) * 16 + 8 + e]; }
    chs[2 * c] = a; chs[2 * c + 1] = b;
  }
  __syncthreads();
}

template <bool SILU, int NV>
__global__ __launch_bounds__(512) void gnc_fwd_kernel(const bf16_t* __restrict__ x, long ldx, bf16_t* __restrict__ y, long ldy, int HW,
                                                      int C, int G, int VX, int PY, int S, int PPS, float eps,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ stats, float* __restrict__ part /*[B][S][G][2]*/) {
  extern __shared__ float sm[];
  float* red = sm;                              // [T][16]
  float* chs = red + max((int)blockDim.x * 16, S * G * 2);   // [C + S][2]
  float* gst = chs + (C + S) * 2;               // [G][2]
  const int t = threadIdx.x, vx = t % VX, py = t / VX;
  const int b = blockIdx.y, sl = blockIdx.x, cg = C / G;
  const long row0 = (long)b * HW + (long)sl * PPS;
  PackB d[NV];
  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = py + i * PY;
    if (p < PPS) d[i].load(x + (row0 + p) * ldx + vx * 8);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = py + i * PY;
    if (p < PPS) {
      float f[8]; d[i].get(f);
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
    }
  }
  gnc_channel_sums(s, q, red, chs, C, VX, PY);
  if (t < G) {
    float a = 0.f, c2 = 0.f;
    for (int c = t * cg; c < (t + 1) * cg; ++c) { a += chs[2 * c]; c2 += chs[2 * c + 1]; }
    float* o = part + (((long)b * S + sl) * G + t) * 2;
    __hip_atomic_store(o, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 1, c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gnc_stores_done();
  }
  gnc_sample_barrier(b, S);
  gnc_gather(part + (long)b * S * G * 2, red, S * G * 2);       // (red is free again: [S][G][2])
  if (t < G) {
    double Sm = 0, Q = 0;
    for (int k = 0; k < S; ++k) { Sm += (double)red[(k * G + t) * 2]; Q += (double)red[(k * G + t) * 2 + 1]; }
    const double n = (double)HW * cg, mean = Sm / n;
    double var = Q / n - mean * mean;
    if (var < 0) var = 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    gst[2 * t] = (float)mean; gst[2 * t + 1] = (float)rstd;
    if (sl == 0) {
      stats[((long)b * G + t) * 2] = (float)mean;
      stats[((long)b * G + t) * 2 + 1] = (float)rstd;
    }
  }
  __syncthreads();
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vx * 8 + e, gi = c / cg;
    const double scd = (double)gst[2 * gi + 1] * (double)gamma[c];
    sc[e] = (float)scd;
    sh[e] = (float)((double)beta[c] - (double)gst[2 * gi] * scd);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = py + i * PY;
    if (p < PPS) {
      float f[8]; d[i].get(f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = f[e] * sc[e] + sh[e];
        f[e] = SILU ? silu_f(z) : z;
      }
      store8(y + (row0 + p) * ldy + vx * 8, f);
    }
  }
}

// backward: dz = dy * silu'(.) ; per channel s = sum dz, q = sum dz * xhat ; dx = rstd (gamma dz - (S1 + xhat S2) / n) with the
// gamma-weighted group sums S1 = sum gamma s, S2 = sum gamma q (norm.hip gn1_bwd_kernel: the same arithmetic)
template <bool SILU, int NV>
__global__ __launch_bounds__(512) void gnc_bwd_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                      const bf16_t* __restrict__ accum, long ldacc, bf16_t* __restrict__ dx, long lddx,
                                                      int HW, int C, int G, int VX, int PY, int S, int PPS,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ stats, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, float* __restrict__ part /*[B][S][G][2]*/,
                                                      float* __restrict__ cpart /*[B][S][C][2] (trainable norms)*/) {
  extern __shared__ float sm[];
  float* red = sm;
  float* chs = red + max((int)blockDim.x * 16, S * G * 2);
  float* gst = chs + (C + S) * 2;
  const int t = threadIdx.x, vx = t % VX, py = t / VX;
  const int b = blockIdx.y, sl = blockIdx.x, cg = C / G;
  const long row0 = (long)b * HW + (long)sl * PPS;
  PackB dxv[NV], ddv[NV];
  float ga[8], be[8], mu[8], rs[8], s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vx * 8 + e, gi = c / cg;
    ga[e] = gamma[c]; be[e] = beta[c];
    mu[e] = stats[((long)b * G + gi) * 2]; rs[e] = stats[((long)b * G + gi) * 2 + 1];
    s[e] = 0.f; q[e] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = py + i * PY;
    if (p < PPS) {
      dxv[i].load(x + (row0 + p) * ldx + vx * 8);
      ddv[i].load(dy + (row0 + p) * lddy + vx * 8);
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = py + i * PY;
    if (p < PPS) {
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
  gnc_channel_sums(s, q, red, chs, C, VX, PY);
  if (dgamma) {
    float* o = cpart + ((long)b * S + sl) * C * 2;
    for (int c = t; c < C; c += blockDim.x) {
      __hip_atomic_store(o + 2 * c, chs[2 * c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(o + 2 * c + 1, chs[2 * c + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gnc_stores_done();
  }
  if (t < G) {
    double s1 = 0, s2 = 0;
    for (int c = t * cg; c < (t + 1) * cg; ++c) { s1 += (double)gamma[c] * chs[2 * c]; s2 += (double)gamma[c] * chs[2 * c + 1]; }
    float* o = part + (((long)b * S + sl) * G + t) * 2;
    __hip_atomic_store(o, (float)s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 1, (float)s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    gnc_stores_done();
  }
  gnc_sample_barrier(b, S);
  gnc_gather(part + (long)b * S * G * 2, red, S * G * 2);       // (red is free again: [S][G][2])
  if (t < G) {
    double s1 = 0, s2 = 0;
    for (int k = 0; k < S; ++k) { s1 += (double)red[(k * G + t) * 2]; s2 += (double)red[(k * G + t) * 2 + 1]; }
    gst[2 * t] = (float)s1; gst[2 * t + 1] = (float)s2;
  }
  if (dgamma) {
    // this workgroup's share of the sample's channels (C / S of them, rounded up): sums over the S slabs in fixed order, then
    // one float atomic per channel and sample (as the other forms).  Thread (k, j) fetches slab k's pair of channel c0 + j.
    const int per = (C + S - 1) / S, c0 = sl * per, nc = min(C, c0 + per) - c0;
    __syncthreads();
    if (nc > 0) {
      for (int i = t; i < S * nc; i += blockDim.x) {
        const int k = i / nc, j = i - k * nc;
        const float* o = cpart + ((long)b * S + k) * C * 2 + 2 * (c0 + j);
        float v0, v1;
        asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %2, off offset:4 sc1" : "=&v"(v0), "=&v"(v1) : "v"(o) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        chs[2 * i] = v0; chs[2 * i + 1] = v1;                   // (chs [C][2] is free again; S * nc <= C + S)
      }
    }
    __syncthreads();
    if (t < nc) {
      float a = 0.f, g2 = 0.f;
      for (int k = 0; k < S; ++k) { a += chs[2 * (k * nc + t)]; g2 += chs[2 * (k * nc + t) + 1]; }
      atomicAdd(dbeta + c0 + t, a);
      atomicAdd(dgamma + c0 + t, g2);
    }
  }
  __syncthreads();
  const double n = (double)HW * cg;
  float k1[8], k2[8], k3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = vx * 8 + e, gi = c / cg;
    const double rstd = rs[e];
    k1[e] = (float)(rstd * ga[e]);
    k2[e] = (float)(rstd * (double)gst[2 * gi] / n);
    k3[e] = (float)(rstd * (double)gst[2 * gi + 1] / n);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = py + i * PY;
    if (p < PPS) {
      float f[8], d[8], ac[8]; dxv[i].get(f); ddv[i].get(d);
      if (accum) load8(accum + (row0 + p) * ldacc + vx * 8, ac);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (f[e] - mu[e]) * rs[e];
        float dz = d[e];
        if (SILU) dz *= dsilu_f(xh * ga[e] + be[e]);
        float r = dz * k1[e] - k2[e] - xh * k3[e];
        if (accum) r += ac[e];
        f[e] = r;
      }
      store8(dx + (row0 + p) * lddx + vx * 8, f);
    }
  }
}

struct GncGeom { int VX, PY, T, S, PPS, NV, lds; bool ok; };

// S: the fewest slabs per sample (a power of two dividing HW) that gives every CU a workgroup, subject to the slab fitting
// nvmax vectors per lane; taken only where the other forms lose (groups spanning >= 1024 pixels)
GncGeom gnc_geom(int B, int HW, int C, int G, int nvmax, long need_per_ws, long have_ws) {
  GncGeom g{}; g.ok = false;
  if (!g_gn_coop || g_gn_three_pass || C % 8 || C % G || G > 512 || B > GNC_MAXB || HW < 1024) return g;
  g.VX = C / 8;
  if (g.VX > 512) return g;
  g.PY = 512 / g.VX;
  g.T = g.VX * g.PY;
  if (g.T < G || g.T < 64) return g;
  int S = 1;
  while (S < 64 && (B * S < 256 || (HW / S + g.PY - 1) / g.PY > nvmax)) S *= 2;
  if (HW % S) return g;
  g.S = S; g.PPS = HW / S;
  const int nv = (g.PPS + g.PY - 1) / g.PY;
  if (nv > nvmax || S < 2) return g;
  g.NV = nv <= 4 ? 4 : nv <= 8 ? 8 : nv <= 12 ? 12 : 16;
  g.lds = (std::max(g.T * 16, S * G * 2) + (C + S) * 2 + G * 2) * 4;
  if ((long)B * S * need_per_ws > have_ws) return g;
  g.ok = true;
  return g;
}

// every workgroup of the grid resident at once?  (the occupancy query is remembered per kernel instance and launch shape)
template <typename K>
bool gnc_resident(K kern, int threads, int lds, long wgs) {
  struct Seen { int threads, lds; long cap; };
  static Seen seen[8];
  static int nseen = 0;
  for (int i = 0; i < nseen; ++i)
    if (seen[i].threads == threads && seen[i].lds == lds) return wgs <= seen[i].cap;
  int dev = 0, cus = 0, per = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, threads, lds) != hipSuccess) return false;
  const long cap = (long)per * cus;
  if (nseen < 8) seen[nseen++] = Seen{threads, lds, cap};
  return wgs <= cap;
}

}  // namespace

unsigned gnc_timeouts() {
  unsigned v = 0;
  (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_gnc_timeouts), sizeof(v));
  return v;
}

// CL_EINVAL = not this form's case (the caller goes on to the other forms)
int gnc_fwd(const GnArgs& a, int dtype, hipStream_t st) {
  if (dtype != CL_BF16) return CL_EINVAL;
  const GncGeom g = gnc_geom(a.B, a.HW, a.C, a.G, 16, (long)a.G * 2, gn_ws_floats(a.B, a.HW, a.C));
  if (!g.ok) return CL_EINVAL;
  dim3 grid(g.S, a.B);
#define GNC_FWD(SL, NVV)                                                                                                     \
  {                                                                                                                          \
    auto kern = gnc_fwd_kernel<SL, NVV>;                                                                                     \
    if (g.lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds) != hipSuccess) \
      return CL_EINVAL;                                                                                                      \
    if (!gnc_resident(kern, g.T, g.lds, (long)g.S * a.B)) return CL_EINVAL;                                                  \
    hipLaunchKernelGGL(kern, grid, dim3(g.T), g.lds, st, (const bf16_t*)a.x, a.ldx, (bf16_t*)a.y, a.ldy, a.HW, a.C, a.G, g.VX, \
                       g.PY, g.S, g.PPS, a.eps, a.gamma, a.beta, a.stats, a.ws);                                             \
  }
#define GNC_FWD_NV(SL) switch (g.NV) { case 4: GNC_FWD(SL, 4) break; case 8: GNC_FWD(SL, 8) break; case 12: GNC_FWD(SL, 12) break; default: GNC_FWD(SL, 16) break; }
  if (a.silu) { GNC_FWD_NV(true) } else { GNC_FWD_NV(false) }
#undef GNC_FWD_NV
#undef GNC_FWD
  CL_CHECK_LAUNCH();
  return CL_OK;
}

int gnc_bwd(const GnBwdArgs& a, int dtype, hipStream_t st) {
  if (dtype != CL_BF16) return CL_EINVAL;
  const long per = (long)a.G * 2 + (a.dgamma ? (long)a.C * 2 : 0);
  const GncGeom g = gnc_geom(a.B, a.HW, a.C, a.G, 12, per, gn_ws_floats(a.B, a.HW, a.C));
  if (!g.ok) return CL_EINVAL;
  dim3 grid(g.S, a.B);
  float* part = a.ws;
  float* cpart = a.ws + (long)a.B * g.S * a.G * 2;
#define GNC_BWD(SL, NVV)                                                                                                     \
  {                                                                                                                          \
    auto kern = gnc_bwd_kernel<SL, NVV>;                                                                                     \
    if (g.lds > 48 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, g.lds) != hipSuccess) \
      return CL_EINVAL;                                                                                                      \
    if (!gnc_resident(kern, g.T, g.lds, (long)g.S * a.B)) return CL_EINVAL;                                                  \
    hipLaunchKernelGGL(kern, grid, dim3(g.T), g.lds, st, (const bf16_t*)a.x, a.ldx, (const bf16_t*)a.dy, a.lddy,             \
                       (const bf16_t*)a.accum, a.ldacc, (bf16_t*)a.dx, a.lddx, a.HW, a.C, a.G, g.VX, g.PY, g.S, g.PPS, a.gamma, \
                       a.beta, a.stats, a.dgamma, a.dbeta, part, cpart);                                                     \
  }
#define GNC_BWD_NV(SL) switch (g.NV) { case 4: GNC_BWD(SL, 4) break; case 8: GNC_BWD(SL, 8) break; default: GNC_BWD(SL, 12) break; }
  if (a.silu) { GNC_BWD_NV(true) } else { GNC_BWD_NV(false) }
#undef GNC_BWD_NV
#undef GNC_BWD
  CL_CHECK_LAUNCH();
  return CL_OK;
}

}  // namespace cl
