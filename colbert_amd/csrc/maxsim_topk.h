// maxsim_topk.h -- per-query top-k and small utility kernels.
#pragma once
#include "maxsim_common.h"
#include "maxsim_sort.h"

namespace maxsim {

template <int R>
__global__ void __launch_bounds__(256) k_topk_small(const float* __restrict__ scores, const int64_t* __restrict__ pids,
                                                    int ncand, int k, float* __restrict__ out_s,
                                                    int64_t* __restrict__ out_p) {
  __shared__ uint64_t lds[256 * R];
  const int q = blockIdx.x;
  wg_topk_row<R>(scores + (int64_t)q * ncand, pids ? pids + (int64_t)q * ncand : nullptr, ncand, k,
                 out_s + (int64_t)q * k, out_p + (int64_t)q * k, lds, threadIdx.x);
}

// =============================================================================================
// Long lists (2048 < ncand <= 16384 = the reference's BSIZE): bitonic sort of the keys in LDS, one workgroup per query.
// =============================================================================================
__global__ void __launch_bounds__(1024) k_topk(const float* __restrict__ scores, const int64_t* __restrict__ pids,
                                               int ncand, int k, int P, float* __restrict__ out_s,
                                               int64_t* __restrict__ out_p) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  uint64_t* keys = (uint64_t*)lds;  // [P]
  const int q = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < P; i += nt) {
    uint64_t key = 0;  // below every real key (orderable(-inf) = 0x007fffff > 0)
    if (i < ncand) key = ((uint64_t)orderable(scores[(int64_t)q * ncand + i]) << 32) | (uint32_t)(~(uint32_t)i);
    keys[i] = key;
  }
  __syncthreads();
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = tid; i < (P >> 1); i += nt) {
        int lo = ((i / stride) * (stride << 1)) + (i % stride);
        int hi = lo + stride;
        bool desc = ((lo & size) == 0);  // descending overall
        uint64_t a = keys[lo], b = keys[hi];
        bool swap = desc ? (a < b) : (a > b);
        if (swap) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < k; i += nt) {
    float s = NEG_INF;
    int64_t pid = -1;
    if (i < ncand) {
      uint64_t key = keys[i];
      int pos = (int)(~(uint32_t)key);
      s = unorderable((uint32_t)(key >> 32));
      pid = pids ? pids[(int64_t)q * ncand + pos] : (int64_t)pos;
    }
    out_s[(int64_t)q * k + i] = s;
    out_p[(int64_t)q * k + i] = pid;
  }
}

__global__ void k_fill(float* out, int64_t n, float v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}

}  // namespace maxsim
