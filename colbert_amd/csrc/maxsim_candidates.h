// maxsim_candidates.h -- candidate-side glue: ANN embedding ids -> per-query unique pid lists, on the GPU.
// Replaces ColbertIndex.embedding_ids_to_pids (reference: colbert/ranking/colbert_ranker.py:212-229: emb2pid
// lookup, .tolist(), per-query set() in a Pool(16)) and the 4-byte-per-token emb2pid table of build_emb2pid
// (:163-174): the pid of a token row is found by binary search in the doclens prefix sum the ranker already holds.
#pragma once
#include "maxsim_common.h"
#include "maxsim_sort.h"

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
  // pid of a token row e = (number of docs whose first row is <= e) - 1 (skipping empty docs that start at the same row):
  // a binary search in the doclens prefix sum, ~20 dependent 8-byte loads for a million docs -- and with the ids in ANN
  // order every lane of a load instruction hits a different cache line: 16384 x 20 loads cost the CU's address pipeline
  // ~0.2 ms of a 0.46 ms launch.  The pid is monotone in the token row, so when the rows fit 32 bits the ROWS are sorted
  // first and looked up afterwards: neighbouring lanes then walk neighbouring table entries (the same few lines per
  // instruction), and the distinct step below sees the same sorted pids.
  const bool rows_first = n_tokens < 0xFFFFFFFFll;
  auto pid_of = [&](int64_t e) {
    int64_t lo = 0, hi = n_docs;               // invariant: tok_offsets[lo'] <= e for lo' < lo ; > e for >= hi
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (tok_offsets[mid] <= e) lo = mid + 1; else hi = mid;
    }
    return (uint32_t)(lo - 1);
  };
  for (int i = tid; i < P; i += nt) {
    uint32_t key = 0xFFFFFFFFu;
    if (i < n) {
      const int64_t e = emb_ids[(int64_t)q * n + i];
      if (e >= 0 && e < n_tokens) key = rows_first ? (uint32_t)e : pid_of(e);   // FAISS pads missing neighbours with -1
    }
    keys[i] = key;
  }
  __syncthreads();
  // bitonic network: keys in registers where the workgroup is full (P >= 2048), else all in LDS; size and stride are powers
  // of two: shifts and masks, not the integer divisions `i / stride` costs
  if (nt == 1024 && P == 16384) bitonic_sort_regs<16, uint32_t, false>(keys, P, tid);
  else if (nt == 1024 && P == 8192) bitonic_sort_regs<8, uint32_t, false>(keys, P, tid);
  else if (nt == 1024 && P == 4096) bitonic_sort_regs<4, uint32_t, false>(keys, P, tid);
  else if (nt == 1024 && P == 2048) bitonic_sort_regs<2, uint32_t, false>(keys, P, tid);
  else
  for (int size = 2; size <= P; size <<= 1) {
    for (int ls = 31 - __builtin_clz(size) - 1; ls >= 0; --ls) {
      const int stride = 1 << ls;
      for (int i = tid; i < (P >> 1); i += nt) {
        const int lo = ((i >> ls) << (ls + 1)) | (i & (stride - 1));
        const int hi = lo + stride;
        const bool asc = ((lo & size) == 0);
        const uint32_t a = keys[lo], b = keys[hi];
        if (asc ? (a > b) : (a < b)) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  if (rows_first) {  // sorted token rows -> their (sorted) pids, in place.  A thread runs its searches G at a time in
    // lock-step (every round issues G independent loads, no branches): the chain it waits for is ~20 loads, not 20 x P / nt
    constexpr int G = 8;
    int steps = 1;
    while ((1ll << steps) <= n_docs) ++steps;   // rounds until every bracket [lo, hi) is empty
    const int64_t last = n_docs > 0 ? n_docs - 1 : 0;
    for (int i0 = tid; i0 < P; i0 += G * nt) {
      int64_t e[G], lo[G], hi[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int i = i0 + g * nt;
        const uint32_t k32 = i < P ? keys[i] : 0xFFFFFFFFu;
        e[g] = k32 == 0xFFFFFFFFu ? -1 : (int64_t)k32;
        lo[g] = 0;
        hi[g] = e[g] >= 0 ? n_docs : 0;
      }
      for (int s = 0; s < steps; ++s) {
        int64_t mid[G], off[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          mid[g] = (lo[g] + hi[g]) >> 1;
          off[g] = tok_offsets[mid[g] < last ? mid[g] : last];   // (a closed bracket re-reads a valid entry and ignores it)
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const bool open = lo[g] < hi[g], right = off[g] <= e[g];
          lo[g] = (open && right) ? mid[g] + 1 : lo[g];
          hi[g] = (open && !right) ? mid[g] : hi[g];
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int i = i0 + g * nt;
        if (i < P && e[g] >= 0) keys[i] = (uint32_t)(lo[g] - 1);
      }
    }
    __syncthreads();
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
