// maxsim_backward.h -- backward of the all-pairs MaxSim operator (training form: the reference differentiates
// BaseModel.score through torch autograd, colbert/modeling/colbert_model.py:87-96, materialising the
// [q, d, m, n] similarity tensor).  Here the forward records arg-max indices (ReducerArg) and the backward routes
// gradients through them:
//     dQ[q,m,:] = q_mask[q,m] * sum_d g[q,d] * d_mask[d,i] * D[d,i,:]          i = argmax[q,d,m]
//     dD[d,i,:] += d_mask[d,i] * g[q,d] * q_mask[q,m] * Q[q,m,:]               for every (q,m) with argmax == i
// fp32 accumulation whatever the input type; dD uses float atomics (sum order, hence the last bits, may vary from
// run to run -- MI355X_MICROARCH "Global float atomics").
#pragma once
#include "maxsim_common.h"

namespace maxsim {

template <int DT>
__global__ void __launch_bounds__(256) k_maxsim_bwd_dq(const void* __restrict__ D, const void* __restrict__ q_mask,
                                                       const void* __restrict__ d_mask, int mask_dtype,
                                                       const int32_t* __restrict__ argmax,
                                                       const float* __restrict__ grad, float* __restrict__ dQ, int nd,
                                                       int Lq, int Ld, int h) {
  const int qm = blockIdx.x;  // q * Lq + m
  const int q = qm / Lq;
  const int m = qm - q * Lq;
  const float qs = mask_dtype != MAXSIM_MASK_NONE ? load_mask(q_mask, mask_dtype, qm) : 1.0f;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // h <= 1024 = 4 x 256 threads
  for (int d = 0; d < nd; ++d) {
    const int i = argmax[((int64_t)q * nd + d) * Lq + m];
    float g = grad[(int64_t)q * nd + d];
    if (mask_dtype != MAXSIM_MASK_NONE) g *= load_mask(d_mask, mask_dtype, (int64_t)d * Ld + i);
    const int64_t row = ((int64_t)d * Ld + i) * h;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = threadIdx.x + 256 * j;
      if (k < h) acc[j] = fmaf(g, load_elem<DT>(D, row + k), acc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = threadIdx.x + 256 * j;
    if (k < h) dQ[(int64_t)qm * h + k] = acc[j] * qs;
  }
}

template <int DT>
__global__ void __launch_bounds__(256) k_maxsim_bwd_dd(const void* __restrict__ Q, const void* __restrict__ q_mask,
                                                       const void* __restrict__ d_mask, int mask_dtype,
                                                       const int32_t* __restrict__ argmax,
                                                       const float* __restrict__ grad, float* __restrict__ dD, int nd,
                                                       int Lq, int Ld, int h) {
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
