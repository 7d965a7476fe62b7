// maxsim_shard.h -- small index-side kernels: the packed per-doc descriptor table and the doc-shard candidate filter.
#pragma once
#include "maxsim_common.h"

namespace maxsim {

// One 16-byte row per doc: {int64 first token row, int32 doclen, int32 bucket stride}.  The rerank kernels then fetch
// ONE random cache line per candidate instead of three (tok_offsets / doclens / pad_len live in separate arrays).
__global__ void __launch_bounds__(256) k_build_doc_table(const int64_t* __restrict__ tok_offsets,
                                                         const int32_t* __restrict__ doclens,
                                                         const int32_t* __restrict__ pad_len, int64_t n_docs,
                                                         int4* __restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_docs) return;
  const int64_t off = tok_offsets[i];
  const int len = doclens[i];
  int4 r;
  r.x = (int)(uint32_t)off;
  r.y = (int)(uint32_t)((uint64_t)off >> 32);
  r.z = len;
  r.w = pad_len ? pad_len[i] : len;
  table[i] = r;
}

// Doc-sharded rerank: every rank receives the same GLOBAL candidate lists; this rank owns the pid range
// [lo, hi).  One workgroup per query moves the query's in-range candidates to the front of its row, in list order
// (a stable partition: the per-query top-k tie order -- lower list position first -- survives sharding), as LOCAL pids
// (pid - lo) next to the global ones; the rest of the row is -1 padding.  The rerank kernel then streams dense rows
// and its trailing all-padding waves retire at once.  No host sync: the row width stays ncand.
__global__ void __launch_bounds__(256) k_shard_candidates(const int64_t* __restrict__ cand, int ncand, int64_t lo,
                                                          int64_t hi, int64_t* __restrict__ out_local,
                                                          int64_t* __restrict__ out_global,
                                                          int32_t* __restrict__ out_count) {
  __shared__ int wave_tot[4];
  __shared__ int base_s;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t* row = cand + (int64_t)q * ncand;
  int64_t* ol = out_local + (int64_t)q * ncand;
  int64_t* og = out_global ? out_global + (int64_t)q * ncand : nullptr;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int c0 = 0; c0 < ncand; c0 += 256) {
    const int c = c0 + tid;
    const int64_t pid = c < ncand ? row[c] : -1;
    const bool in = pid >= lo && pid < hi;
    const uint64_t bal = __ballot(in);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int wbase = base_s;
    for (int w = 0; w < wave; ++w) wbase += wave_tot[w];
    if (in) {
      ol[wbase + before] = pid - lo;
      if (og) og[wbase + before] = pid;
    }
    __syncthreads();
    if (tid == 0) base_s += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
  }
  const int total = base_s;
  for (int i = total + tid; i < ncand; i += 256) {
    ol[i] = -1;
    if (og) og[i] = -1;
  }
  if (tid == 0 && out_count) out_count[q] = total;
}

}  // namespace maxsim
