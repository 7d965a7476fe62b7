// maxsim_sort.h -- device-side building blocks of the top-k: orderable score keys and the workgroup sort for short lists
// (templates / inline only: included by the top-k kernels and by the streaming kernel's fused epilogue).
#pragma once
#include "maxsim_common.h"

namespace maxsim {

// key = orderable(score) << 32 | ~position  -> descending sort = score desc, position asc.
__device__ __forceinline__ uint32_t orderable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// =============================================================================================
// Workgroup sort for short lists (the reference's online call ranks 1000 candidates, colbert_ranker.py:128):
// 256 threads hold R keys each (element e = tid * R + r), P = 256 R <= 2048 keys, bitonic network, DESCENDING.
// Compare-exchange partners at distance < R sit in the same thread (registers), at distance < 64 R in the same wave
// (lane exchange, no barrier), only the last log2(4) distances cross waves through LDS: 3 barrier pairs for P = 1024
// against the 55 of a plain LDS bitonic sort (19 us -> ~3 us for one query, which is what a single rank_forward waits for).
// =============================================================================================
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
  const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, mask);
  const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), mask);
  return ((uint64_t)hi << 32) | lo;
}

template <int R>
__device__ __forceinline__ void wg_sort_desc(uint64_t (&k)[R], uint64_t* lds /* [256 R] */, int tid) {
  constexpr int P = 256 * R;
#pragma unroll
  for (int size = 2; size <= P; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride < R) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (r & stride) continue;
          const int e = tid * R + r;
          const bool desc = (e & size) == 0;
          const uint64_t a = k[r], b = k[r | stride];
          const bool sw = desc ? (a < b) : (a > b);
          k[r] = sw ? b : a;
          k[r | stride] = sw ? a : b;
        }
      } else if (stride < 64 * R) {
        const int lx = stride / R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int e = tid * R + r;
          const uint64_t o = shfl_xor_u64(k[r], lx);
          // the pair's lower element keeps the larger key in a descending block, the smaller in an ascending one
          const bool keep_max = ((e & size) == 0) == ((e & stride) == 0);
          const bool o_gt = o > k[r];
          k[r] = (keep_max == o_gt) ? o : k[r];
        }
      } else {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) lds[tid * R + r] = k[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int e = tid * R + r;
          const uint64_t o = lds[e ^ stride];
          const bool keep_max = ((e & size) == 0) == ((e & stride) == 0);
          const bool o_gt = o > k[r];
          k[r] = (keep_max == o_gt) ? o : k[r];
        }
      }
    }
  }
}

// Top-k of one query's score row by the workgroup sort: loads the row, sorts, writes the first k (score, pid) pairs.
// `lds` needs 256 R keys (8 bytes each).  All 256 threads of the workgroup must call it.
template <int R>
__device__ __forceinline__ void wg_topk_row(const float* __restrict__ srow, const int64_t* __restrict__ prow, int ncand,
                                            int k, float* __restrict__ out_s, int64_t* __restrict__ out_p,
                                            uint64_t* lds, int tid) {
  uint64_t key[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid * R + r;
    key[r] = i < ncand ? (((uint64_t)orderable(srow[i]) << 32) | (uint32_t)(~(uint32_t)i)) : 0ull;  // 0 < every real key
  }
  wg_sort_desc<R>(key, lds, tid);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = tid * R + r;
    if (i < k) {
      float s = NEG_INF;
      int64_t pid = -1;
      if (i < ncand) {
        const int pos = (int)(~(uint32_t)key[r]);
        s = unorderable((uint32_t)(key[r] >> 32));
        pid = prow ? prow[pos] : (int64_t)pos;
      }
      out_s[i] = s;
      out_p[i] = pid;
    }
  }
}

}  // namespace maxsim
