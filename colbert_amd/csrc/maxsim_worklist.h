// maxsim_worklist.h -- the dense work list of counted candidate rows.
//
// A candidate matrix whose rows are only partly live -- one rank's share of a doc-sharded step (maxsim_shard_candidates:
// ~1/N of every 1000-wide row), or the distinct pids of an ANN search (maxsim_embedding_ids_to_pids) -- is a bad fit for
// the static (query, chunk) grid of the streaming kernels: most of the grid is all-padding workgroups, the last live
// chunk of every row leaves waves of its workgroup without docs (the workgroup keeps its LDS until its one busy wave is
// done), and nothing of this is known on the host without a synchronisation.  Here the device builds, from the per-row
// live counts, a list of WAVE ITEMS (query, first slot, docs) -- every item about as many docs as a wave's stream should
// hold, a row's docs dealt evenly over its items -- and the LIST form of the streaming kernel (maxsim_stream.h) runs a
// fixed grid whose waves walk that list: item = wave id, + waves in the grid, ...  No host sync, no all-padding
// workgroups, every wave of every workgroup busy.
//
// Layout of the work list (device memory, int32 words):
//   [0] number of wave items   [1] docs per item the builder settled on   [2] live candidates in all rows   [3..15] 0
//   [16 .. 16 + nq]            item_start[q] (exclusive prefix sum of the rows' item counts; [nq] = word 0)
//   then, 16-byte aligned:     items[] as int2 {query, first slot | docs << 20}
// The h = 128 kernels take WAVE items (<= 64 docs, a wave's stream); the LDS-query kernel for wider rows, whose waves share
// the staged query, takes WORKGROUP items (<= 64 docs per wave: docs per item = waves x docs per wave) and deals an item's
// docs evenly over its waves.
#pragma once
#include "maxsim_common.h"

namespace maxsim {

constexpr int WL_HEADER_WORDS = 16;
constexpr int WL_SLOT_BITS = 20;  // first slot < 2^20; docs per item < 2^11
__host__ __device__ inline int64_t worklist_items_word(int nq) { return (WL_HEADER_WORDS + (int64_t)nq + 1 + 3) & ~(int64_t)3; }

__device__ __forceinline__ int wl_row_count(const int32_t* __restrict__ counts, int q, int ncand) {
  return min(max(counts[q], 0), ncand);
}

__device__ __forceinline__ void worklist_fill_row(const int32_t* __restrict__ counts, int q, int nq, int ncand,
                                                  int32_t* __restrict__ wl, float* __restrict__ scores, int fill_tail,
                                                  int D, int base, int tid, int nt);

// One workgroup: (1) the number of live candidates -> docs per wave item D (the host's target D0 = a ~1.4 k-token stream,
// halved while the launch would have fewer than `min_items` items: a small launch is better off with more, shorter
// streams -- the rule pick_docs_per_wave applies on the host to static grids; then, for lists of 2-4 rounds of the
// kernel's `slots` resident wave (workgroup) slots, the cut that wastes the least of the last round --
// refine_docs_per_wave's reasoning, on the device); (2) exclusive scan of ceil(count / D).
// fused_scores != nullptr (a handful of queries: the online driver serving one query at a time): this workgroup also does
// k_worklist_fill's work -- one launch and one dependent launch gap less on a ~100 us call.
static __global__ void __launch_bounds__(1024) k_worklist_scan(const int32_t* __restrict__ counts, int nq, int ncand, int D0,
                                                        int Dmax, int min_items, int slots, int32_t* __restrict__ wl,
                                                        float* __restrict__ fused_scores = nullptr) {
  __shared__ long long red[16];
  __shared__ int wsum[16];
  __shared__ int carry_s, D_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long long live = 0;
  for (int q = tid; q < nq; q += 1024) live += wl_row_count(counts, q, ncand);
  for (int o = 32; o > 0; o >>= 1) live += __shfl_xor(live, o);
  if (lane == 0) red[wave] = live;
  __syncthreads();
  if (tid == 0) {
    long long t = 0;
    for (int w = 0; w < 16; ++w) t += red[w];
    int D = D0 < 1 ? 1 : (D0 > Dmax ? Dmax : D0);
    while (D > 1 && (t + D - 1) / D < min_items) D = (D + 1) / 2;
    if (slots > 0 && nq > 0 && t > 0) {
      // mid-size lists (2-4 rounds of the resident slots): the cut with the smallest rounds x docs per item among D / 2 ..
      // 2 D, taken when it saves >= 10 % -- with the rows' AVERAGE count standing in for every row (a doc shard's rows
      // hold ~1000 / N each; for rows of very different lengths the estimate is rough and any D is still correct)
      const long long cbar = (t + nq - 1) / nq;
      auto rounds = [&](int d) { return ((long long)nq * ((cbar + d - 1) / d) + slots - 1) / slots; };
      const long long r0 = rounds(D);
      if (r0 >= 2 && r0 <= 4) {
        long long best_cost = r0 * D;
        int best = D;
        const int hi = 2 * D < Dmax ? 2 * D : Dmax, lo = D / 2 > 1 ? D / 2 : 1;
        for (int d = hi; d >= lo; --d) {
          const long long c = rounds(d) * d;
          if (c < best_cost) { best_cost = c; best = d; }
        }
        if (best_cost * 10 <= r0 * D * 9) D = best;
      }
    }
    D_s = D;
    carry_s = 0;
    wl[1] = D;
    wl[2] = (int32_t)(t > 0x7fffffffLL ? 0x7fffffffLL : t);
    for (int i = 3; i < WL_HEADER_WORDS; ++i) wl[i] = 0;
  }
  __syncthreads();
  const int D = D_s;
  int32_t* const item_start = wl + WL_HEADER_WORDS;
  for (int q0 = 0; q0 < nq; q0 += 1024) {
    const int q = q0 + tid;
    const int w = q < nq ? (wl_row_count(counts, q, ncand) + D - 1) / D : 0;
    int incl = w;  // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o);
      incl += lane >= o ? up : 0;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int base = carry_s;
    for (int k = 0; k < wave; ++k) base += wsum[k];
    if (q < nq) item_start[q] = base + incl - w;
    __syncthreads();
    if (tid == 1023) carry_s = base + incl;
    __syncthreads();
  }
  if (tid == 0) {
    item_start[nq] = carry_s;
    wl[0] = carry_s;
  }
  if (fused_scores != nullptr) {
    __syncthreads();                     // (item_start was written by this workgroup: visible after the barrier)
    for (int q = 0; q < nq; ++q) worklist_fill_row(counts, q, nq, ncand, wl, fused_scores, 1, D, item_start[q], tid, 1024);
  }
}

// A row's items (its docs dealt evenly: item j of w takes slots [j c / w, (j + 1) c / w)) and the -inf tail of the row's
// scores (slots past the live count are padding slots: what the static kernels write there), by `nt` threads.
__device__ __forceinline__ void worklist_fill_row(const int32_t* __restrict__ counts, int q, int nq, int ncand,
                                                  int32_t* __restrict__ wl, float* __restrict__ scores, int fill_tail,
                                                  int D, int base, int tid, int nt) {
  const int c = wl_row_count(counts, q, ncand);
  const int w = (c + D - 1) / D;
  int2* const items = (int2*)(wl + worklist_items_word(nq));
  for (int j = tid; j < w; j += nt) {
    const int b = (int)((long long)j * c / w), e = (int)((long long)(j + 1) * c / w);
    items[base + j] = make_int2(q, b | ((e - b) << WL_SLOT_BITS));
  }
  if (fill_tail) {
    float* const row = scores + (int64_t)q * ncand;
    // 16-byte stores over the aligned middle of the tail, single floats at its two ends
    const int a0 = min(ncand, c + (int)((4 - (((uintptr_t)(row + c) >> 2) & 3)) & 3));
    const int a1 = a0 + ((ncand - a0) & ~3);
    for (int i = c + tid; i < a0; i += nt) row[i] = NEG_INF;
    const float4 ninf = make_float4(NEG_INF, NEG_INF, NEG_INF, NEG_INF);
    for (int i = a0 + 4 * tid; i < a1; i += 4 * nt) *(float4*)(row + i) = ninf;
    for (int i = a1 + tid; i < ncand; i += nt) row[i] = NEG_INF;
  }
}

// One workgroup of 256 threads per query.
static __global__ void __launch_bounds__(256) k_worklist_fill(const int32_t* __restrict__ counts, int nq, int ncand,
                                                      int32_t* __restrict__ wl, float* __restrict__ scores,
                                                      int fill_tail) {
  const int q = blockIdx.x;
  worklist_fill_row(counts, q, nq, ncand, wl, scores, fill_tail, wl[1], wl[WL_HEADER_WORDS + q], threadIdx.x, 256);
}

}  // namespace maxsim
