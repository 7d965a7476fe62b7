// maxsim_candidates.h -- candidate-side glue: ANN embedding ids -> per-query unique pid lists, on the GPU.
// Replaces ColbertIndex.embedding_ids_to_pids (reference: colbert/ranking/colbert_ranker.py:212-229: emb2pid
// lookup, .tolist(), per-query set() in a Pool(16)) and the 4-byte-per-token emb2pid table of build_emb2pid
// (:163-174): the pid of a token row is found by binary search in the doclens prefix sum the ranker already holds.
#pragma once
#include "maxsim_common.h"

namespace maxsim {

// One workgroup per query.  keys: P (power of two >= n) uint32 in LDS, 0xFFFFFFFF = dropped id.
// Output: the query's distinct pids in ascending order, then -1 padding; out_count[q] = number of distinct pids.
__global__ void __launch_bounds__(1024) k_unique_pids(const int64_t* __restrict__ emb_ids, int n, int P,
                                                      const int64_t* __restrict__ tok_offsets, int64_t n_docs,
                                                      int64_t n_tokens, int64_t* __restrict__ out_pids,
                                                      int32_t* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  uint32_t* keys = (uint32_t*)lds;            // [P]
  uint32_t* scan = keys + P;                  // [blockDim.x]
  const int q = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < P; i += nt) {
    uint32_t key = 0xFFFFFFFFu;
    if (i < n) {
      const int64_t e = emb_ids[(int64_t)q * n + i];
      if (e >= 0 && e < n_tokens) {            // FAISS pads missing neighbours with -1
        // pid = (number of docs whose first row is <= e) - 1, skipping empty docs that start at the same row
        int64_t lo = 0, hi = n_docs;           // invariant: tok_offsets[lo'] <= e for lo' < lo ; > e for >= hi
        while (lo < hi) {
          const int64_t mid = (lo + hi) >> 1;
          if (tok_offsets[mid] <= e) lo = mid + 1; else hi = mid;
        }
        key = (uint32_t)(lo - 1);
      }
    }
    keys[i] = key;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (P >> 1); i += nt) {
        const int lo = ((i / stride) * (stride << 1)) + (i % stride);
        const int hi = lo + stride;
        const bool asc = ((lo & size) == 0);
        const uint32_t a = keys[lo], b = keys[hi];
        if (asc ? (a > b) : (a < b)) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  // distinct keys: each thread owns a contiguous run of P / nt elements
  const int per = (P + nt - 1) / nt;
  const int b0 = tid * per, b1 = min(P, b0 + per);
  uint32_t cnt = 0;
  for (int i = b0; i < b1; ++i) {
    const uint32_t k = keys[i];
    cnt += (k != 0xFFFFFFFFu && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
  }
  scan[tid] = cnt;
  __syncthreads();
  for (int off = 1; off < nt; off <<= 1) {     // inclusive Hillis-Steele scan
    const uint32_t v = tid >= off ? scan[tid - off] : 0u;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  const uint32_t total = scan[nt - 1];
  uint32_t pos = scan[tid] - cnt;
  for (int i = b0; i < b1; ++i) {
    const uint32_t k = keys[i];
    if (k != 0xFFFFFFFFu && (i == 0 || keys[i - 1] != k)) out_pids[(int64_t)q * n + pos++] = (int64_t)k;
  }
  for (int i = (int)total + tid; i < n; i += nt) out_pids[(int64_t)q * n + i] = -1;
  if (tid == 0) out_count[q] = (int32_t)total;
}

}  // namespace maxsim
