// maxsim_backward.h -- backward of the all-pairs MaxSim operator (training form: the reference differentiates
// BaseModel.score through torch autograd, colbert/modeling/colbert_model.py:87-96, materialising the
// [q, d, m, n] similarity tensor).  Here the forward records arg-max indices (ReducerArg) and the backward routes
// gradients through them:
//     dQ[q,m,:] = q_mask[q,m] * sum_d g[q,d] * d_mask[d,i] * D[d,i,:]          i = argmax[q,d,m]
//     dD[d,i,:] = d_mask[d,i] * sum_{(q,m): argmax[q,d,m] == i} g[q,d] * q_mask[q,m] * Q[q,m,:]
// Both are gather-reduce problems over rows that sit in L2 / Infinity Cache.  Indices and coefficients are loaded 64
// at a time into lanes and walked with v_readlane, so the row loads of consecutive items are independent and stay
// in flight together (a scalar index load per item would serialise on memory latency).  fp32 accumulation.
#pragma once
#include "maxsim_common.h"

namespace maxsim {

__device__ __forceinline__ float lane_f32(float v, int l) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), l));
}

// dQ: one workgroup per (q, m); thread t owns dims t, t + 256, ... (h <= 1024).
template <int DT>
__global__ void __launch_bounds__(256) k_maxsim_bwd_dq(const void* __restrict__ D, const void* __restrict__ q_mask,
                                                       const void* __restrict__ d_mask, int mask_dtype,
                                                       const int32_t* __restrict__ argmax,
                                                       const float* __restrict__ grad, float* __restrict__ dQ, int nd,
                                                       int Lq, int Ld, int h) {
  const int qm = blockIdx.x;  // q * Lq + m
  const int q = qm / Lq;
  const int m = qm - q * Lq;
  const int lane = threadIdx.x & 63;
  const float qs = mask_dtype != MAXSIM_MASK_NONE ? load_mask(q_mask, mask_dtype, qm) : 1.0f;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int d0 = 0; d0 < nd; d0 += 64) {
    // lane j describes doc d0 + j: row of its arg-max token and the coefficient g * d_mask
    const int d = d0 + lane;
    int64_t row = 0;
    float g = 0.0f;
    if (d < nd) {
      const int i = argmax[((int64_t)q * nd + d) * Lq + m];
      row = (int64_t)d * Ld + i;
      g = grad[(int64_t)q * nd + d];
      if (mask_dtype != MAXSIM_MASK_NONE) g *= load_mask(d_mask, mask_dtype, row);
    }
    // 4 docs at a time: their row loads are independent and in flight together; docs past the end have g = 0, row 0
    for (int j0 = 0; j0 < 64; j0 += 4) {
      if (d0 + j0 >= nd) break;
      float gj[4];
      int64_t base[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        gj[t] = lane_f32(g, j0 + t);
        base[t] = (((int64_t)__builtin_amdgcn_readlane((int)(row >> 32), j0 + t) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)row, j0 + t)) * h;
      }
      float x[4][4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = threadIdx.x + 256 * u;
          x[t][u] = k < h ? load_elem<DT>(D, base[t] + k) : 0.0f;
        }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fmaf(gj[t], x[t][u], acc[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int k = threadIdx.x + 256 * u;
    if (k < h) dQ[(int64_t)qm * h + k] = acc[u] * qs;
  }
}

// dD: one workgroup (16 waves) per (doc d, 64-dim chunk); the doc's [Ld][64] fp32 gradient slab lives in LDS and
// is written out once -- no global atomics, no read-modify-write of dD.  LDS float atomics are very slow on gfx950
// (~170 cycles per wave-instruction measured), so the slab rows are OWNED: wave w accumulates only the items whose
// arg-max token n has n % 16 == w, with plain ds_read / ds_write.  Every wave scans all (q, m) items 64 at a time
// (indices in lanes), picks its own with a ballot and walks them 4 at a time so the 4 query-row loads are in flight
// together.  The sum order is fixed, so dD is bitwise reproducible.  LDS = Ld * 256 B (Ld <= 600).
template <int DT>
__global__ void __launch_bounds__(1024) k_maxsim_bwd_dd_lds(const void* __restrict__ Q, const void* __restrict__ q_mask,
                                                            const void* __restrict__ d_mask, int mask_dtype,
                                                            const int32_t* __restrict__ argmax,
                                                            const float* __restrict__ grad, float* __restrict__ dD,
                                                            int nq, int nd, int Lq, int Ld, int h, int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  float* slab = (float*)lds;  // [Ld][64]
  const int d = blockIdx.x / nchunk;
  const int k0 = (blockIdx.x - d * nchunk) * 64;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < Ld * 64; i += 1024) slab[i] = 0.0f;
  __syncthreads();
  const int k = min(k0 + lane, h - 1);  // lanes past h (last chunk of a non-multiple-of-64 h) add into unused columns
  const int nitem = nq * Lq;            // items (q, m) in row-major order
  for (int p0 = 0; p0 < nitem; p0 += 64) {
    const int pi = p0 + lane;
    int idx = 0;
    float coef = 0.0f;
    if (pi < nitem) {
      const int q = pi / Lq, m = pi - q * Lq;
      idx = argmax[((int64_t)q * nd + d) * Lq + m];
      coef = grad[(int64_t)q * nd + d];
      if (mask_dtype != MAXSIM_MASK_NONE)
        coef *= load_mask(q_mask, mask_dtype, pi) * load_mask(d_mask, mask_dtype, (int64_t)d * Ld + idx);
    }
    uint64_t own = __ballot(((idx & 15) == wave) && coef != 0.0f);
    while (own) {
      int jj[4];
      float x[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        jj[t] = own ? (int)__builtin_ctzll(own) : -1;
        own = own ? (own & (own - 1)) : own;
        x[t] = jj[t] >= 0 ? load_elem<DT>(Q, (int64_t)(p0 + jj[t]) * h + k) : 0.0f;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (jj[t] >= 0) {
          const float cj = lane_f32(coef, jj[t]);
          float* cell = slab + __builtin_amdgcn_readlane(idx, jj[t]) * 64 + lane;
          *cell = fmaf(cj, x[t], *cell);
        }
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Ld * 64; i += 1024) {
    const int n = i >> 6, kk = k0 + (i & 63);
    if (kk < h) dD[((int64_t)d * Ld + n) * h + kk] = slab[i];
  }
}

// dD fallback for very long documents (slab does not fit in LDS): global float atomics into a zeroed dD.
template <int DT>
__global__ void __launch_bounds__(256) k_maxsim_bwd_dd_atomic(const void* __restrict__ Q,
                                                              const void* __restrict__ q_mask,
                                                              const void* __restrict__ d_mask, int mask_dtype,
                                                              const int32_t* __restrict__ argmax,
                                                              const float* __restrict__ grad, float* __restrict__ dD,
                                                              int nd, int Lq, int Ld, int h) {
  const int64_t qd = blockIdx.x;  // q * nd + d
  const int q = (int)(qd / nd);
  const int d = (int)(qd - (int64_t)q * nd);
  const float g = grad[qd];
  if (g == 0.0f) return;
  for (int m = 0; m < Lq; ++m) {
    const int i = argmax[qd * Lq + m];
    float coef = g;
    if (mask_dtype != MAXSIM_MASK_NONE)
      coef *= load_mask(q_mask, mask_dtype, (int64_t)q * Lq + m) * load_mask(d_mask, mask_dtype, (int64_t)d * Ld + i);
    if (coef == 0.0f) continue;
    const int64_t src = ((int64_t)q * Lq + m) * h;
    float* dst = dD + ((int64_t)d * Ld + i) * h;
    for (int k = threadIdx.x; k < h; k += 256) atomicAdd(dst + k, coef * load_elem<DT>(Q, src + k));
  }
}

}  // namespace maxsim
