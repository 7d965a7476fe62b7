// maxsim_generic.h -- correctness kernel for any h / Lq / doc length / element type (one workgroup per pair).
#pragma once
#include "maxsim_common.h"

namespace maxsim {

// =============================================================================================
// Generic kernel: any h / Lq / doc length / element type.  One workgroup per (query, candidate).
// Correctness path for shapes the MFMA kernels do not cover (e.g. the reference's 2x2x3 KAT).
// =============================================================================================
template <int DT, int MODE>
__global__ void __launch_bounds__(256) k_maxsim_generic(KARGS_DECL) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  float* smax = (float*)lds;  // [Lq]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int qi = blockIdx.x / p.ncand;
  const int c = blockIdx.x - qi * p.ncand;
  const Doc d = load_doc<MODE>(p, qi, c);
  const int h = p.h;
  int qlen = p.Lq;
  if (MODE == MODE_RERANK && p.q_len) qlen = min(qlen, p.q_len[qi]);
  const bool masked = MODE == MODE_DENSE && p.mask_dtype != MAXSIM_MASK_NONE;

  for (int mq = wave; mq < p.Lq; mq += 4) {
    float best = NEG_INF;
    int bestn = 0;
    const bool live = q_token_live<MODE>(p, qi, mq, qlen);
    float qs = 1.0f;
    if (masked) qs = load_mask(p.q_mask, p.mask_dtype, (int64_t)qi * p.Lq + mq);
    const int64_t qbase = ((int64_t)qi * p.Lq + mq) * h;
    if (live && d.kind == 0) {
      for (int nn = lane; nn < d.len; nn += 64) {
        float ds = 1.0f;
        if (masked) ds = load_mask(p.d_mask, p.mask_dtype, d.row0 + nn);
        const int64_t dbase = (d.row0 + nn) * h;
        float acc = 0.0f;
        for (int k = 0; k < h; ++k) {
          float qv = load_q(p.Q, p.q_dtype, qbase + k);
          float dv = load_elem<DT>(p.index, dbase + k);
          if (masked) { qv *= qs; dv *= ds; }
          acc = fmaf(qv, dv, acc);
        }
        if (acc > best) { best = acc; bestn = nn; }  // strict: the first maximal token wins (torch.max)
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int on = __shfl_xor(bestn, o);
      const bool take = (ob > best) || (ob == best && on < bestn);
      best = take ? ob : best;
      bestn = take ? on : bestn;
    }
    if (MODE == MODE_DENSE && p.argmax && lane == 0) p.argmax[((int64_t)qi * p.ncand + c) * p.Lq + mq] = bestn;
    if (d.floor0) best = fmaxf(best, 0.0f);
    if (!live) best = 0.0f;  // dropped query token contributes nothing
    if (lane == 0) smax[mq] = best;
  }
  __syncthreads();
  if (wave == 0) {
    float s = 0.0f;
    for (int mq = lane; mq < p.Lq; mq += 64) s += smax[mq];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) p.scores[(int64_t)qi * p.ncand + c] = d.kind == 0 ? s : (d.kind == 1 ? 0.0f : NEG_INF);
  }
}

}  // namespace maxsim
