// maxsim_topk.h -- per-query top-k (colbert_ranker.py:128-130) and small utility kernels.
#pragma once
#include "maxsim_common.h"
#include "maxsim_sort.h"

namespace maxsim {

// key = orderable(score) << 32 | ~position  -> descending key order = score desc, position asc.  Keys are unique.
__device__ __forceinline__ uint32_t orderable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

// =============================================================================================
// Short lists (ncand <= 2048; the reference's online call ranks ~1000 candidates): RANK BY COUNTING.  The rank of a
// candidate is the number of keys above its own; keys are unique, so ranks are a permutation and every candidate with
// rank < k writes its own output slot: no sorting network, no rounds, no cross-workgroup communication.  A workgroup
// ranks 16 candidates of one query, 16 lanes per candidate (each lane compares against every 16th key of the row, held
// in LDS; the 16 partial counts are added with DPP): ncand / 16 independent workgroups per query.
// Why not sort: a sort of 1024 keys is ~55 DEPENDENT exchange steps on one workgroup and took 15-19 us whatever the
// network (LDS bitonic 19 us, register/lane-exchange bitonic 16 us, radix selection + counting 14 us: the steps'
// latency, not their work, is the cost) -- half the GPU time of a whole rank_forward.  Counting is n^2 / 2 comparisons but
// all of them independent: ~3 us for one query, and it scales over the chip for a batch (256 x 1000: 16k workgroups).
// done_flag (optional, host-visible): the last workgroup of the launch stores `ticket` to it after every output of
// the launch is written (counter: one zeroed int32, left zero), so that a host thread can poll instead of calling
// hipStreamSynchronize.
// =============================================================================================
constexpr int TOPK_CAND_PER_WG = 16;

// The ranking step of k_topk_count: `keys` (LDS, n16 entries, 0 past the n real
// ones) -> the 16 candidates of group g each count the keys above their own and, when that rank is < k, write
// (score, pid) to output slot `rank`.  COHERENT: the host may read the outputs while the kernel is still running
// (system-scope write-through stores).  256 threads.
template <bool COHERENT>
__device__ __forceinline__ void topk_rank_group(const uint64_t* keys, int n, int n16, int g, int k,
                                                const int64_t* __restrict__ pid_row, float* __restrict__ out_s,
                                                int64_t* __restrict__ out_p) {
  const int tid = threadIdx.x;
  const int c = g * TOPK_CAND_PER_WG + (tid >> 4), part = tid & 15;
  const uint64_t mine = keys[max(min(c, n16 - 1), 0)];
  int above = 0;
#pragma unroll 4
  for (int j = part; j < n16; j += 16) above += keys[j] > mine ? 1 : 0;
  // sum over the 16 lanes of the candidate (one DPP row): xor 1, xor 2, then the two mirrors
  above += __builtin_amdgcn_update_dpp(0, above, 0xB1, 0xF, 0xF, false);
  above += __builtin_amdgcn_update_dpp(0, above, 0x4E, 0xF, 0xF, false);
  above += __builtin_amdgcn_update_dpp(0, above, 0x141, 0xF, 0xF, false);
  above += __builtin_amdgcn_update_dpp(0, above, 0x140, 0xF, 0xF, false);
  if (part == 0 && c < n && above < k) {
    const int pos = (int)(~(uint32_t)mine);
    const float s = unorderable((uint32_t)(mine >> 32));
    const int64_t pid = pid_row ? pid_row[pos] : (int64_t)pos;
    if constexpr (COHERENT) {
      __hip_atomic_store(out_s + above, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(out_p + above, pid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
      out_s[above] = s;
      out_p[above] = pid;
    }
  }
}

__global__ void __launch_bounds__(256) k_topk_count(const float* __restrict__ scores, const int64_t* __restrict__ pids,
                                                    int ncand, int k, float* __restrict__ out_s,
                                                    int64_t* __restrict__ out_p, int groups, int32_t* counter,
                                                    uint32_t* done_flag, uint32_t ticket,
                                                    const int32_t* __restrict__ counts, int gstride) {
  __shared__ uint64_t keys[2048];
  const int tid = threadIdx.x;
  // `groups` workgroups per query; workgroup g ranks the candidate groups g, g + gstride, ... (gstride == groups: one each.
  // Counted rows of a big batch take fewer workgroups per row than the row is wide: most of a 1000-wide row of a doc
  // shard's share is padding, and 63 workgroups per row that load the keys to find they have nothing to rank cost more
  // than 8 that loop -- 2048 rows x 125 live: 61 -> ~25 us)
  const int q = blockIdx.x / groups, g = blockIdx.x - q * groups;
  const int row_w = ncand;  // row pitch of scores / pids
  // counted rows: only the first counts[q] slots of the row are candidates (the rest is (-inf, -1) padding, which is also
  // what the slots [n, k) of the output receive) -- the groups past them have nothing to rank
  if (counts) {
    ncand = min(max(counts[q], 0), row_w);
    if (g > 0 && g * TOPK_CAND_PER_WG >= ncand) return;  // (counts are never combined with done_flag)
  }
  const float* srow = scores + (int64_t)q * row_w;
  const int n16 = (ncand + 15) & ~15;
  for (int i = tid; i < n16; i += 256)
    keys[i] = i < ncand ? (((uint64_t)orderable(srow[i]) << 32) | (uint32_t)(~(uint32_t)i)) : 0ull;  // 0 < every real key
  if (n16 == 0 && tid == 0) keys[0] = 0ull;  // an empty counted row: topk_rank_group reads (and ignores) keys[0]
  __syncthreads();
  const int64_t* const pid_row = pids ? pids + (int64_t)q * row_w : nullptr;
  if (done_flag)  // the host may read the outputs while the kernel is still running: system-scope write-through
    topk_rank_group<true>(keys, ncand, n16, g, k, pid_row, out_s + (int64_t)q * k, out_p + (int64_t)q * k);
  else
    for (int gg = g; gg == g || gg * TOPK_CAND_PER_WG < ncand; gg += gstride)
      topk_rank_group<false>(keys, ncand, n16, gg, k, pid_row, out_s + (int64_t)q * k, out_p + (int64_t)q * k);
  // slots k' in [ncand, k) (k > ncand): (-inf, -1), written by the query's first workgroup
  if (g == 0)
    for (int i = ncand + tid; i < k; i += 256) {
      out_s[(int64_t)q * k + i] = NEG_INF;
      out_p[(int64_t)q * k + i] = -1;
    }
  if (done_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's outputs are acknowledged ...
    __syncthreads();                                   // ... and so are the workgroup's
    if (tid == 0 &&
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
      __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // left zero for the next launch
      __hip_atomic_store(done_flag, ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// =============================================================================================
// Long lists (2048 < ncand <= 16384 = the reference's BSIZE): bitonic sort of the keys in LDS, one workgroup per query.
// =============================================================================================
// counts (optional): counted rows -- only the first counts[q] slots of row q are candidates (the rest is (-inf, -1) padding,
// which is also what the output slots past them receive): the row is sorted as the next power of two >= its count, e.g. a
// 16384-wide row of ANN pids with ~2000 distinct candidates as 2048 keys (36 exchange stages instead of 105).
__global__ void __launch_bounds__(1024) k_topk(const float* __restrict__ scores, const int64_t* __restrict__ pids,
                                               int ncand, int k, int P, float* __restrict__ out_s,
                                               int64_t* __restrict__ out_p, const int32_t* __restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  uint64_t* keys = (uint64_t*)lds;  // [P]
  const int q = blockIdx.x;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int row_w = ncand;
  if (counts) {
    ncand = min(max(counts[q], 0), row_w);
    int Pq = 2;
    while (Pq < ncand) Pq <<= 1;
    P = min(P, Pq);
  }
  for (int i = tid; i < P; i += nt) {
    uint64_t key = 0;  // below every real key (orderable(-inf) = 0x007fffff > 0)
    if (i < ncand) key = ((uint64_t)orderable(scores[(int64_t)q * row_w + i]) << 32) | (uint32_t)(~(uint32_t)i);
    keys[i] = key;
  }
  __syncthreads();
  // descending bitonic sort: in registers (maxsim_sort.h) where the row fills the workgroup, else in LDS
  if (nt == 1024 && P == 16384) bitonic_sort_regs<16, uint64_t, true>(keys, P, tid);
  else if (nt == 1024 && P == 8192) bitonic_sort_regs<8, uint64_t, true>(keys, P, tid);
  else if (nt == 1024 && P == 4096) bitonic_sort_regs<4, uint64_t, true>(keys, P, tid);
  else if (nt == 1024 && P == 2048) bitonic_sort_regs<2, uint64_t, true>(keys, P, tid);
  else
  for (int size = 2; size <= P; size <<= 1) {
    for (int ls = 31 - __builtin_clz(size) - 1; ls >= 0; --ls) {
      const int stride = 1 << ls;
      for (int i = tid; i < (P >> 1); i += nt) {
        int lo = ((i >> ls) << (ls + 1)) | (i & (stride - 1));
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
      pid = pids ? pids[(int64_t)q * row_w + pos] : (int64_t)pos;
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
