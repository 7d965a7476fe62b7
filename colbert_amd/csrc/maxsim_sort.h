// maxsim_sort.h -- workgroup-wide bitonic sort with the keys in registers (used by the ids -> distinct pids kernel,
// maxsim_candidates.h, and the long-row top-k, maxsim_topk.h).
#pragma once
#include "maxsim_common.h"

namespace maxsim {

// The bitonic network over P = 1024 E keys with E keys per thread in REGISTERS (thread t owns elements t E .. t E + E - 1):
// strides below E are compare-exchanges inside a thread (no memory at all), strides below 64 E cross-lane exchanges inside a
// wave (one ds_bpermute per key, no barrier), and only the strides from 64 E up -- 10 of the 105 passes at P = 16384 -- go
// through the LDS array with workgroup barriers.  The all-LDS network moves ~13 MB through LDS per 16384-key sort and was
// ~145 us of a one-query launch's ~210.  K: uint32_t or uint64_t keys; DESC: descending order.
template <int E, typename K, bool DESC>
__device__ __forceinline__ void bitonic_sort_regs(K* keys, int P, int tid) {
  K k[E];
#pragma unroll
  for (int j = 0; j < E; ++j) k[j] = keys[tid * E + j];
  for (int size = 2; size <= P; size <<= 1) {
    int stride = size >> 1;
    if (stride >= 64 * E) {  // across waves: through LDS
#pragma unroll
      for (int j = 0; j < E; ++j) keys[tid * E + j] = k[j];
      __syncthreads();
      for (; stride >= 64 * E; stride >>= 1) {
        const int ls = 31 - __builtin_clz(stride);
        for (int i = tid; i < (P >> 1); i += 1024) {
          const int lo = ((i >> ls) << (ls + 1)) | (i & (stride - 1));
          const int hi = lo + stride;
          const bool asc = ((lo & size) == 0) != DESC;
          const K a = keys[lo], b = keys[hi];
          if (asc ? (a > b) : (a < b)) { keys[lo] = b; keys[hi] = a; }
        }
        __syncthreads();
      }
#pragma unroll
      for (int j = 0; j < E; ++j) k[j] = keys[tid * E + j];
    }
    for (; stride >= E; stride >>= 1) {  // across the lanes of a wave
      const int lx = stride / E;
      const bool is_lo = (tid & lx) == 0;
      const bool asc = (((tid * E) & size) == 0) != DESC;  // (size >= 2 stride >= 2 E: the same for the thread's E keys)
      const bool keep_min = asc == is_lo;
#pragma unroll
      for (int j = 0; j < E; ++j) {
        K v;
        if constexpr (sizeof(K) == 8) v = (K)__shfl_xor((long long)k[j], lx); else v = (K)__shfl_xor((int)k[j], lx);
        k[j] = keep_min ? (k[j] < v ? k[j] : v) : (k[j] < v ? v : k[j]);
      }
    }
#pragma unroll
    for (int s = E / 2; s >= 1; s >>= 1) {  // inside the thread (s is a compile-time number in every unrolled copy)
      if (s <= (size >> 1)) {
#pragma unroll
        for (int j = 0; j < E; ++j) {
          if ((j & s) == 0) {
            const bool asc = (((tid * E + j) & size) == 0) != DESC;
            const K a = k[j], b = k[j | s];
            const K mn = a < b ? a : b, mx = a < b ? b : a;
            k[j] = asc ? mn : mx;
            k[j | s] = asc ? mx : mn;
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < E; ++j) keys[tid * E + j] = k[j];
  __syncthreads();
}


}  // namespace maxsim
