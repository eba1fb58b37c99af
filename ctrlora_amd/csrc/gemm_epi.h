// Epilogue shared by the tile kernels (gemm.hip) and the loader / consumer kernel (gemm_w4.hip): bias, time-embedding
// row-bias (openaimodel.py:272), SiLU, alpha / alpha_n, beta * residual (openaimodel.py:274, cldm/cldm.py:41), fp32 / bf16 /
// atomic store -- applied to 8 consecutive columns of one output row.
#pragma once
#include "gemm.h"

namespace cl {

struct EpiArgs {
  const float* bias; const void* rowbias; long ldrb; int rows_per_batch;
  const void* residual; long ldr; float alpha, beta; int act;
  void* C; long ldc; int out_f32; int atomic; int M, N; int alpha_n;
  int ph_mq, ph_win;   // phase-decomposed conv modes (gemm.h): rows per phase, source width; 0 = rows are output rows
};

// apply the epilogue to 8 consecutive columns of one row and store
template <typename T>
__device__ __forceinline__ void epilogue8(const EpiArgs& e, float v[8], int grow, int gcol) {
  if (e.ph_mq) {   // internal row [phase][b][y][x] -> output pixel (2y + a, 2x + b) of the interleaved grid
    const int ph = grow / e.ph_mq, r = grow - ph * e.ph_mq;
    const int q = r / e.ph_win, x = r - q * e.ph_win;
    grow = (q * 2 + (ph >> 1)) * (2 * e.ph_win) + 2 * x + (ph & 1);
  }
  if (e.bias) {
    const float4 b0 = *reinterpret_cast<const float4*>(e.bias + gcol);
    const float4 b1 = *reinterpret_cast<const float4*>(e.bias + gcol + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
    v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (e.rowbias) {
    float rb[8];
    load8(reinterpret_cast<const T*>(e.rowbias) + (long)(grow / e.rows_per_batch) * e.ldrb + gcol, rb);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += rb[i];
  }
  if (e.act == ACT_SILU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = silu_f(v[i]);
  }
  const float al = (e.alpha_n > 0 && gcol >= e.alpha_n) ? 1.0f : e.alpha;     // (8 columns never straddle alpha_n)
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= al;
  if (e.residual) {
    float rs[8];
    load8(reinterpret_cast<const T*>(e.residual) + (long)grow * e.ldr + gcol, rs);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += e.beta * rs[i];
  }
  if (e.atomic) {
    float* dst = reinterpret_cast<float*>(e.C) + (long)grow * e.ldc + gcol;
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(dst + i, v[i]);
  } else if (e.out_f32) {
    store8(reinterpret_cast<float*>(e.C) + (long)grow * e.ldc + gcol, v);
  } else {
    store8(reinterpret_cast<T*>(e.C) + (long)grow * e.ldc + gcol, v);
  }
}

// the same minus bias / row-bias (the caller has added them), for the loader / consumer kernel: the residual's 16 bytes were
// fetched ahead (vmcnt retires in order -- a load issued behind a tile's stores would wait for them to drain), and the output
// element offset is 32-bit (one VGPR per unit instead of a pointer pair; the launcher checks the range)
__device__ __forceinline__ void epilogue8_tail_bf16(const EpiArgs& e, float v[8], unsigned off_c, int gcol, uint4 rs) {
  if (e.act == ACT_SILU) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = silu_f(v[i]);
  }
  const float al = (e.alpha_n > 0 && gcol >= e.alpha_n) ? 1.0f : e.alpha;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] *= al;
  if (e.residual) {
    const uint32_t w[4] = {rs.x, rs.y, rs.z, rs.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] += e.beta * __uint_as_float(w[i] << 16);
      v[2 * i + 1] += e.beta * __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  if (e.out_f32) store8(reinterpret_cast<float*>(e.C) + off_c, v);
  else {
    store8(reinterpret_cast<bf16_t*>(e.C) + off_c, v);
  }
}

__device__ __forceinline__ EpiArgs epi_of(const GemmParams& p) {
  EpiArgs e;
  e.bias = p.bias; e.rowbias = p.rowbias; e.ldrb = p.ldrb; e.rows_per_batch = p.rows_per_batch;
  e.residual = p.residual; e.ldr = p.ldr; e.alpha = p.alpha; e.beta = p.beta; e.act = p.act;
  e.C = p.C; e.ldc = p.ldc; e.out_f32 = p.out_f32; e.atomic = p.atomic; e.M = p.M; e.N = p.N; e.alpha_n = p.alpha_n;
  e.ph_mq = gemm_phase_mode(p.mode) ? p.B * p.Hin * p.Win : 0; e.ph_win = p.Win;
  return e;
}

}  // namespace cl
