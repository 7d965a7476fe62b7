// maxsim_candidates.h -- candidate-side glue: ANN embedding ids -> per-query unique pid lists, on the GPU.
// Replaces ColbertIndex.embedding_ids_to_pids (reference: colbert/ranking/colbert_ranker.py:212-229: emb2pid
// lookup, .tolist(), per-query set() in a Pool(16)) and the 4-byte-per-token emb2pid table of build_emb2pid
// (:163-174): the pid of a token row comes from a 64x coarser table (one entry per 64 token rows, k_build_row_blocks)
// and a search of the doclens prefix sum inside that block -- ~3 loads -- or, without the table, a binary search over
// the whole prefix sum (~20 loads).
#pragma once
#include "maxsim_common.h"
#include "maxsim_sort.h"

namespace maxsim {

constexpr int kRowBlockShift = 6;  // one row-block table entry per 64 token rows
constexpr uint32_t kNoKey = 0xFFFFFFFFu;

// row_blocks[b], b = 0 .. nblocks - 1, describes the 64 token rows from b * 64 on (8 bytes per entry):
//   bits  0..31  the doc that holds row b * 64 (the LAST doc whose first row is <= it: empty docs that start at the same
//                row come before the one that owns it)
//   bits 32..38  0 = the whole block belongs to that doc; else the offset (1..63) inside the block at which the NEXT doc
//                (doc + 1) starts -- the common case for docs of 64 tokens and more
//   bit  39      the block holds more than two docs, or its second doc is not doc + 1 (empty docs in between): look the row
//                up in tok_offsets between this entry's doc and the next entry's
// row_blocks[nblocks] = {n_docs - 1}: the upper end of the last block's search.  One 8-byte load answers a lookup in the
// common case (the 4-byte form of round 4's first version took two table loads + a prefix-sum load per id: 33 of the 67 us
// of a 256-query launch went into those divergent loads).
__global__ void __launch_bounds__(256) k_build_row_blocks(const int64_t* __restrict__ tok_offsets, int64_t n_docs,
                                                          int64_t n_tokens, int64_t nblocks,
                                                          uint64_t* __restrict__ row_blocks) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b > nblocks) return;
  auto doc_of = [&](int64_t e) {  // number of docs whose first row is <= e, minus one
    int64_t lo = 0, hi = n_docs;  // invariant: tok_offsets[i] <= e for i < lo ; > e for i >= hi
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (tok_offsets[mid] <= e) lo = mid + 1; else hi = mid;
    }
    return lo > 0 ? lo - 1 : 0;
  };
  if (b == nblocks) {
    row_blocks[b] = (uint64_t)(uint32_t)(n_docs > 0 ? n_docs - 1 : 0);
    return;
  }
  const int64_t r0 = b << kRowBlockShift;
  const int64_t r1 = (r0 + (1 << kRowBlockShift) < n_tokens ? r0 + (1 << kRowBlockShift) : n_tokens) - 1;  // last row of the block
  const int64_t d0 = doc_of(r0), d1 = doc_of(r1);
  uint64_t ent = (uint64_t)(uint32_t)d0;
  if (d1 == d0 + 1) ent |= (uint64_t)(tok_offsets[d1] - r0) << 32;   // (1..63: d1's first row lies inside the block, past r0)
  else if (d1 != d0) ent |= 1ull << 39;
  row_blocks[b] = ent;
}

// The token row (relative to id_base) behind slot i of query q, or -1: FAISS pads missing neighbours with -1, rows of
// another shard fall outside [0, n_tokens), and the neighbours of a dropped query token (tok_keep[q, i / ids_per_token]
// == 0: keep_nonzero, training_utils.py:48-53) do not count.
__device__ __forceinline__ int64_t candidate_row(const int64_t* __restrict__ emb_ids, int q, int n, int i, int64_t id_base,
                                                 const uint8_t* __restrict__ tok_keep, int ids_per_token, int64_t n_tokens) {
  if (i >= n) return -1;
  if (tok_keep != nullptr) {   // ids_per_token < 0: -(log2 of it) - 1, a shift instead of the division (faiss_depth is 2^k as a rule)
    const int tok = ids_per_token < 0 ? i >> (-ids_per_token - 1) : i / ids_per_token;
    const int ntok = ids_per_token < 0 ? n >> (-ids_per_token - 1) : n / ids_per_token;
    if (tok_keep[(int64_t)q * ntok + tok] == 0) return -1;
  }
  const int64_t e = emb_ids[(int64_t)q * n + i] - id_base;
  return (e >= 0 && e < n_tokens) ? e : -1;
}

// Ascending distinct pids of keys[0 .. P) (kNoKey = nothing) by sorting ALL P keys: the path of a query whose ids hit
// more distinct docs than the hash set below holds.  1024 threads, P in {2048 .. 16384}.
__device__ __forceinline__ void unique_by_full_sort(uint32_t* keys, uint32_t* scan, int q, int n, int P, const int64_t* __restrict__ emb_ids,
                                                 int64_t id_base, const uint8_t* __restrict__ tok_keep, int ids_per_token,
                                                 const int64_t* __restrict__ tok_offsets, int64_t n_docs, int64_t n_tokens,
                                                 int64_t* __restrict__ out_pids, int32_t* __restrict__ out_count) {
  const int tid = threadIdx.x;
  constexpr int nt = 1024;
  // The pid is monotone in the token row, so when the rows fit 32 bits the ROWS are sorted first and looked up afterwards:
  // neighbouring lanes then walk neighbouring table entries (the same few cache lines per load instruction) instead of
  // 64 different lines, and the distinct step below sees the same sorted pids.
  const bool rows_first = n_tokens < 0xFFFFFFFFll;
  auto pid_of = [&](int64_t e) {
    int64_t lo = 0, hi = n_docs;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (tok_offsets[mid] <= e) lo = mid + 1; else hi = mid;
    }
    return (uint32_t)(lo - 1);
  };
  for (int i = tid; i < P; i += nt) {
    const int64_t e = candidate_row(emb_ids, q, n, i, id_base, tok_keep, ids_per_token, n_tokens);
    keys[i] = e < 0 ? kNoKey : (rows_first ? (uint32_t)e : pid_of(e));
  }
  __syncthreads();
  if (P == 16384) bitonic_sort_regs<16, uint32_t, false>(keys, P, tid);
  else if (P == 8192) bitonic_sort_regs<8, uint32_t, false>(keys, P, tid);
  else if (P == 4096) bitonic_sort_regs<4, uint32_t, false>(keys, P, tid);
  else bitonic_sort_regs<2, uint32_t, false>(keys, P, tid);
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
        const uint32_t k32 = i < P ? keys[i] : kNoKey;
        e[g] = k32 == kNoKey ? -1 : (int64_t)k32;
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
  const int per = P / nt;
  const int b0 = tid * per, b1 = b0 + per;
  uint32_t cnt = 0;
  for (int i = b0; i < b1; ++i) {
    const uint32_t k = keys[i];
    cnt += (k != kNoKey && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
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
    if (k != kNoKey && (i == 0 || keys[i - 1] != k)) out_pids[(int64_t)q * n + pos++] = (int64_t)k;
  }
  for (int i = (int)total + tid; i < n; i += nt) out_pids[(int64_t)q * n + i] = -1;
  if (tid == 0) out_count[q] = (int32_t)total;
}

// One workgroup of 1024 threads per query.  The ANN result of a query names each of its docs several times (32 tokens x
// faiss_depth 512 = 16384 ids over a few thousand docs), and the reference wants the SET (colbert_ranker.py:234): the ids
// are looked up and de-duplicated FIRST, in an LDS hash set (open addressing, one ds_cmpst per probe), and only the
// distinct pids are sorted -- e.g. a 2048-key register network instead of the 16384-key one (105 passes) that sorting
// every id costs.  The set has TS = min(16384, 2 P) slots; it is compacted in place (every thread holds its slots in
// registers across a barrier), so the keys to sort re-use the table's memory.  A query that fills the set beyond what
// linear probing handles (a probe chain > 64: ~14000 distinct docs of 16384 ids) takes unique_by_full_sort instead: same
// result, old speed.
// Output: the query's distinct pids in ascending order, then -1 padding; out_count[q] = number of distinct pids.
// LDS: buf[TS] u32 (TS >= P: the full-sort path re-uses it as keys[P]), scan[1024] u32, flags.
__global__ void __launch_bounds__(1024) k_unique_pids(const int64_t* __restrict__ emb_ids, int n, int P, int log_ts,
                                                      int64_t id_base, const uint8_t* __restrict__ tok_keep, int ids_per_token,
                                                      const int64_t* __restrict__ tok_offsets, int64_t n_docs,
                                                      int64_t n_tokens, const uint64_t* __restrict__ row_blocks,
                                                      int64_t* __restrict__ out_pids, int32_t* __restrict__ out_count) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int nt = 1024;
  const int TS = 1 << log_ts;
  uint32_t* table = (uint32_t*)lds;            // [TS]
  uint32_t* keys = table;                      // the compacted set, in place
  uint32_t* scan = table + TS;                 // [1024]
  uint32_t* flags = scan + nt;                 // [0] = overflow
  const int q = blockIdx.x, tid = threadIdx.x;
#ifdef MAXSIM_STAMP_UNIQUE   // timing builds only: 100 MHz stamps of workgroup 0's phases behind out_count[gridDim.x] (a buffer the probe makes longer)
  uint64_t* const stamp_out = (uint64_t*)(out_count + ((gridDim.x + 1) & ~1u));
#define UQ_STAMP(i) do { if (q == 0 && tid == 0) stamp_out[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define UQ_STAMP(i)
#endif
  UQ_STAMP(0);
  for (int i = tid; i < TS; i += nt) table[i] = kNoKey;
  if (tid == 0) flags[0] = 0;
  __syncthreads();
  UQ_STAMP(1);

  // ---- look up and insert.  A thread's ids (i = tid, tid + 1024, ...: 16 of a 16384-id query) are loaded ALL AT ONCE, then
  // their row-block entries all at once: the workgroup's chain is one id-load latency + one table-load latency (two rounds
  // of 8 paid both twice: stamped at 13.1 us of a 28 us one-query launch).  The rest -- the rare search inside a block with
  // several docs, the LDS inserts -- runs in halves of G = 8 to stay inside the 128 registers a 1024-thread workgroup gets.
  constexpr int G = 8, R = 16;
  const int64_t last = n_docs - 1;
  for (int i0 = tid; i0 < n; i0 += R * nt) {
    int64_t ea[R];
    uint64_t enta[R];
#pragma unroll
    for (int g = 0; g < R; ++g) ea[g] = candidate_row(emb_ids, q, n, i0 + g * nt, id_base, tok_keep, ids_per_token, n_tokens);
#pragma unroll
    for (int g = 0; g < R; ++g) enta[g] = (row_blocks != nullptr && ea[g] >= 0) ? row_blocks[ea[g] >> kRowBlockShift] : 0ull;
#pragma unroll
    for (int half = 0; half < R / G; ++half) {
      int64_t e[G];
      uint32_t lo[G], hi[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        e[g] = ea[half * G + g];
        lo[g] = 0;
        hi[g] = e[g] >= 0 ? (uint32_t)last : 0u;
        if (row_blocks != nullptr && e[g] >= 0) {
          const uint64_t ent = enta[half * G + g];
          const uint32_t d0 = (uint32_t)ent, bnd = (uint32_t)(ent >> 32) & 127u;
          lo[g] = d0;
          if ((ent >> 39) & 1) hi[g] = (uint32_t)row_blocks[(e[g] >> kRowBlockShift) + 1];   // rare: search between the two entries
          else lo[g] = hi[g] = d0 + ((bnd != 0 && ((uint32_t)e[g] & 63u) >= bnd) ? 1u : 0u);
        }
      }
      // the doc of row e: the LAST pid in [lo, hi] whose first row is <= e (tok_offsets[lo] <= e holds on entry)
      bool open = false;
#pragma unroll
      for (int g = 0; g < G; ++g) open |= lo[g] < hi[g];
      while (__any(open)) {
        uint32_t mid[G];
        int64_t off[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          mid[g] = lo[g] + ((hi[g] - lo[g] + 1) >> 1);
          off[g] = lo[g] < hi[g] ? tok_offsets[mid[g]] : 0;
        }
        open = false;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          if (lo[g] < hi[g]) {
            if (off[g] <= e[g]) lo[g] = mid[g]; else hi[g] = mid[g] - 1;
          }
          open |= lo[g] < hi[g];
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (e[g] < 0) continue;
        const uint32_t pid = lo[g];
        uint32_t h = (pid * 2654435761u) >> (32 - log_ts);
        int probes = 0;
        for (;;) {
          const uint32_t prev = atomicCAS(&table[h], kNoKey, pid);
          if (prev == kNoKey || prev == pid) break;
          h = (h + 1) & (TS - 1);
          if (++probes > 64) { flags[0] = 1; break; }   // the set is (nearly) full: this query takes the full sort
        }
      }
    }
  }
  __syncthreads();
  UQ_STAMP(2);
  if (flags[0] != 0) {
    __syncthreads();
    unique_by_full_sort((uint32_t*)lds, scan, q, n, P, emb_ids, id_base, tok_keep, ids_per_token, tok_offsets, n_docs, n_tokens,
                        out_pids, out_count);
    return;
  }

  // ---- compact the set in place: thread t owns table slots t, t + 1024, ... (spt = 4 / 8 / 16 of them).  (Slots
  // [t * spt, (t + 1) * spt) read with a lane stride of spt words: 32 lanes on one LDS bank at spt = 16 -- stamped at 5.2 us
  // of a 28 us one-query launch; the order of the compacted keys does not matter, they are sorted next.)
  const int spt = TS / nt;
  uint32_t v[16];
  uint32_t cnt = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    v[j] = j < spt ? table[j * nt + tid] : kNoKey;
    cnt += v[j] != kNoKey ? 1u : 0u;
  }
  uint32_t incl = cnt;                          // inclusive scan inside the wave, then over the 16 waves' totals
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += u;
  }
  if ((tid & 63) == 63) scan[tid >> 6] = incl;
  __syncthreads();                              // (also: every thread has read its slots)
  uint32_t base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < nt / 64; ++w) {
    const uint32_t u = scan[w];
    base += w < (tid >> 6) ? u : 0u;
    total += u;
  }
  uint32_t pos = base + incl - cnt;
#pragma unroll
  for (int j = 0; j < 16; ++j)
    if (v[j] != kNoKey) keys[pos++] = v[j];
  int P2 = 1024;
  while (P2 < (int)total) P2 <<= 1;
  for (int i = (int)total + tid; i < P2; i += nt) keys[i] = kNoKey;
  __syncthreads();
  UQ_STAMP(3);
  // ---- sort the distinct pids (they are distinct: nothing to drop afterwards)
  if (total > 1) {
    if (P2 == 1024) bitonic_sort_regs<1, uint32_t, false>(keys, P2, tid);
    else if (P2 == 2048) bitonic_sort_regs<2, uint32_t, false>(keys, P2, tid);
    else if (P2 == 4096) bitonic_sort_regs<4, uint32_t, false>(keys, P2, tid);
    else if (P2 == 8192) bitonic_sort_regs<8, uint32_t, false>(keys, P2, tid);
    else bitonic_sort_regs<16, uint32_t, false>(keys, P2, tid);
  }
  UQ_STAMP(4);
  for (int i = tid; i < n; i += nt) out_pids[(int64_t)q * n + i] = i < (int)total ? (int64_t)keys[i] : (int64_t)-1;
  if (tid == 0) out_count[q] = (int32_t)total;
  UQ_STAMP(5);
}


}  // namespace maxsim
