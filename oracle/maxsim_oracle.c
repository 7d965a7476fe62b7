/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference MaxSim operator, independent of torch/BLAS.
 *
 * Follows colbert/modeling/BaseModel.py:39-46 of wuyaoxuehun/colbert literally:
 *   D = D * d_mask[..., None]            (:41)   fp32 multiply
 *   Q = Q * q_mask[..., None]            (:42)
 *   simmat[q,d,m,n] = sum_h Q[q,m,h] * D[d,n,h]   (:43)  fp32 products, accumulated in double here so that the
 *                                                        result bounds every fp32 summation order
 *   scores_match = simmat.max(-1)        (:44)
 *   scores = scores_match.sum(-1)        (:45)
 * and the ragged closed form the fused kernel computes (colbert/ranking/colbert_ranker.py:88-118: real tokens of the
 * candidate, max floored at 0 iff the reference would have padded the doc).
 * Only tests/ may load this (oracle/__init__.py); never the product path.
 */
#include <math.h>
#include <stdint.h>

void oracle_score_dense(const float* Q, const float* D, const float* q_mask, const float* d_mask, int nq, int nd,
                        int Lq, int Ld, int h, double* out) {
  for (int q = 0; q < nq; ++q)
    for (int d = 0; d < nd; ++d) {
      double total = 0.0;
      for (int m = 0; m < Lq; ++m) {
        const float qs = q_mask[(int64_t)q * Lq + m];
        double best = -INFINITY;
        for (int n = 0; n < Ld; ++n) {
          const float ds = d_mask[(int64_t)d * Ld + n];
          double acc = 0.0;
          for (int k = 0; k < h; ++k) {
            const float a = Q[((int64_t)q * Lq + m) * h + k] * qs;  /* fp32 product, as the reference */
            const float b = D[((int64_t)d * Ld + n) * h + k] * ds;
            acc += (double)a * (double)b;
          }
          if (acc > best) best = acc;
        }
        total += best;
      }
      out[(int64_t)q * nd + d] = total;
    }
}

/* index [n_tokens, h] fp32; one query Q [Lq, h]; candidate list pids[n]; out[n] */
void oracle_rerank_one(const float* index, const int64_t* tok_offsets, const int32_t* doclens, const int32_t* pad_len,
                       const float* Q, int Lq, int h, const int64_t* pids, int n, double* out) {
  for (int i = 0; i < n; ++i) {
    const int64_t pid = pids[i];
    const int64_t off = tok_offsets[pid];
    const int len = doclens[pid];
    if (len == 0) { out[i] = 0.0; continue; }
    const int floor0 = pad_len && pad_len[pid] > len;
    double total = 0.0;
    for (int m = 0; m < Lq; ++m) {
      double best = floor0 ? 0.0 : -INFINITY;
      for (int t = 0; t < len; ++t) {
        double acc = 0.0;
        for (int k = 0; k < h; ++k) acc += (double)Q[(int64_t)m * h + k] * (double)index[(off + t) * h + k];
        if (acc > best) best = acc;
      }
      total += best;
    }
    out[i] = total;
  }
}
