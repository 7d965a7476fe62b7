// maxsim_stream_dyn.h -- the short-doc rerank kernel (k_maxsim_stream_f32h: fp32 index, dim 128, docs of a few tokens --
// the 8-token multi-view config) as a PERSISTENT launch whose waves take their work from queues.
//
// Why: a wave of the static kernel takes 64 docs = 256 KB, a workgroup lives ~80 us, and a 256-query launch is two
// rounds of such workgroups on the chip's 512 slots.  Stamped per workgroup (s_memrealtime at entry and exit): the same
// work takes 71 .. 107 us depending on WHERE it runs (a slot that is slow once is slow again: with two fixed items per
// wave the launch takes 2 x the slowest slot, 0.26 ms), so the second round drains for 35 us at ~40 % occupancy and the
// slots the first round frees refill unevenly -- about 20 us of a 190 us launch.  Smaller static workgroups even that out
// but each pays its start-up chain (candidate -> descriptor row -> first tile: three dependent memory latencies) in the
// open (docs per wave 64 / 32 / 16: 0.199 / 0.210 / 0.220 ms).
//
// Here the grid is two workgroups per CU, once.  An ITEM is `dpi` docs of one query (16: 64 KB of 8-token docs, ~10 us of
// a wave's stream).  Every wave runs: take an item, stream it with the static kernel's tile loop, write its scores, repeat
// -- with the next items' descriptors fetched ahead (the queue counter's return value, the candidate ids and the
// descriptor rows of three consecutive items are in flight while the current one streams, each requested one item before
// it is needed), so an item starts from registers.  Fast slots take more items; the launch ends within one item of the
// last byte.
//
// Queues: ONE atomic counter cannot serve this (same-address atomics serialise at ~10 ns each: tools/micro/
// scalar_atomic_check.hip reads 15 us per returning atomic with one counter under load, 1 us with 8, 0.3 us with 64).
// There are 64: queue x holds the items with id % 64 == x and is drained by the 32 waves of the workgroups with
// blockIdx % 64 == x (workgroup ids are dealt round-robin over the XCDs: a queue's waves sit on one XCD) -- dynamic among
// those waves, static across queues.  Item id = query * items_per_query + chunk.  The counters live in a per-stream slot of a small device
// table (tu_stream_dyn.hip); the wave that finishes last zeroes them for the stream's next launch.
#pragma once
#include "maxsim_stream.h"

namespace maxsim {

// load_doc_lanes<MODE_RERANK> (maxsim_stream.h) in three steps, each of which can run an item ahead of the next: the
// candidate id and the packed descriptor row are fetched RAW -- unconditionally, from clamped addresses, with no use of
// the value at the place of the load: a use (a select, a range test, a branch join) makes the compiler wait for the load
// on the spot.
__device__ __forceinline__ int64_t dyn_fetch_pid(const Params& p, int qi, int c0, int lane) {
  return p.cand[(int64_t)qi * p.ncand + min(c0 + lane, p.ncand - 1)];
}
__device__ __forceinline__ int4 dyn_fetch_row(const Params& p, int64_t pid) {
  const bool ok = pid >= 0 && pid < p.n_docs;
  return ((const int4*)p.doc_table)[ok ? pid : 0];
}
__device__ __forceinline__ DocLanes dyn_decode(const Params& p, int64_t pid, int4 r, int ndoc, int lane) {
  DocLanes d;
  bool ok = lane < ndoc && pid >= 0 && pid < p.n_docs;
  const int64_t off = (int64_t)(((uint64_t)(uint32_t)r.y << 32) | (uint32_t)r.x);
  const int len = r.z, pad = r.w;
  ok = ok && off >= 0 && len >= 0 && off + len <= p.n_tokens;  // defensive: never stream outside the matrix
  const int kind = !ok ? 2 : (len == 0 ? 1 : 0);
  d.row0 = kind == 0 ? (uint32_t)off : 0u;
  d.len = kind == 0 ? len : 0;
  d.flags = kind | ((ok && pad > len) ? 4 : 0);
  return d;
}

#ifndef MAXSIM_DYN_QUEUES
#define MAXSIM_DYN_QUEUES 64
#endif
constexpr int DYN_QUEUES = MAXSIM_DYN_QUEUES;  // counters [0, Q): items taken per queue; counter [Q]: waves finished

// p.dpw = docs per item (<= 64), p.nchunk = items per query, p.argmax = the nine counters (the arg-max pointer is unused
// in rerank mode; this keeps the argument list the one every stream kernel shares).
template <int WAVES, int NCB, int NT = 2>  // NCB 16-column query blocks: 1 (Lq <= 16) or 2 (Lq <= 32)
__global__ void __launch_bounds__(WAVES * 64) k_maxsim_stream_f32h_dyn(KARGS_DECL) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  constexpr int ROWB = 512, HT = 16 * ROWB, NDMA = 8;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  char* const wlds = lds + wave * (NT * HT);
  const int n16 = lane & 15, kq = lane >> 4;
  const char* const tok = (const char*)p.index;
  const int xq = (int)blockIdx.x & (DYN_QUEUES - 1);
  int* const taken = p.argmax + xq;
  int* const finished = p.argmax + DYN_QUEUES;
  const int dpi = p.dpw, ipq = p.nchunk;
  const int total = p.nq * ipq;

  // item id -> (query, first candidate slot, docs); ids past the end are empty items
  struct Item {
    int id, q, c0, nd;
  };
  auto decode = [&](int taken_before) {  // the k-th item of this wave's queue
    Item t;
    t.id = taken_before * DYN_QUEUES + xq;
    const bool ok = t.id < total;
    t.q = ok ? t.id / ipq : 0;
    t.c0 = ok ? (t.id - t.q * ipq) * dpi : 0;
    t.nd = ok ? max(0, min(dpi, p.ncand - t.c0)) : 0;
    return t;
  };
  // take one item: a SCALAR atomic (s_atomic_add with return).  A vector atomic's return sits in the in-order vmcnt queue
  // in front of the item's tiles, and a returning agent-scope atomic takes microseconds on this part: every item's first
  // counted wait waited for it (+0.05 .. 0.15 ms per launch, however many queues).  The scalar one is counted in lgkmcnt;
  // its value is first looked at after the next wait_lgkmcnt0() -- the tile loop has one per tile -- and the explicit one
  // at the end of an item.
  auto take_issue = [&]() {
    int v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(v) : "s"(taken) : "memory");
    return v;
  };
  auto take_value = [&](int v) {  // (after a wait_lgkmcnt0)
    asm volatile("" : "+s"(v));
    return v;
  };

  // the pipeline while item A streams: B has its candidate ids and descriptor rows (requested an item ago, decoded when A
  // is done), C its candidate ids (its rows are requested now), D its id (its candidate ids are requested now), E's id is
  // taken now
  int r0 = take_issue(), r1 = take_issue(), r2 = take_issue(), rawD = take_issue();
  wait_lgkmcnt0();
  Item A = decode(take_value(r0)), B = decode(take_value(r1)), C_ = decode(take_value(r2));
  rawD = take_value(rawD);
  int64_t pidB = dyn_fetch_pid(p, B.q, B.c0, lane);
  int64_t pidC = dyn_fetch_pid(p, C_.q, C_.c0, lane);
  int4 rowB = dyn_fetch_row(p, pidB);
  DocLanes dlA;
  {
    const int64_t pidA = dyn_fetch_pid(p, A.q, A.c0, lane);
    dlA = dyn_decode(p, pidA, dyn_fetch_row(p, pidA), A.nd, lane);
  }

  f32x4 qv[8 * NCB];  // lane (n, kq) holds Q[16 cb + n][16 j + 4 kq + t] in qv[8 cb + j][t]
  int q_loaded = -1;
  while (A.id < total) {
    const int qi = A.q, c_begin = A.c0, ndoc = A.nd;
    const DocLanes dl = dlA;
    // the query tile (L2), when the wave moves to another query: requested before the item's first tiles, so that the first
    // counted wait below covers it and leaves the second tile in flight -- and before the prefetches, so that the waits
    // inside (q_len / q_mask, when given) do not cover them
    const bool q_new = q_loaded != qi;
    bool q_live[NCB];
    if (q_new) {
      q_loaded = qi;
      int qlen = p.Lq;
      if (p.q_len) qlen = min(qlen, p.q_len[qi]);
      const bool qf32 = p.q_dtype == MAXSIM_F32;
      int64_t qo[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int qt = p.q_tok0 + 16 * cb + n16;
        q_live[cb] = q_token_live<MODE_RERANK>(p, qi, qt, qlen);
        qo[cb] = ((int64_t)qi * p.Lq + (q_live[cb] ? qt : 0)) * 128;
      }
      // (raw loads only, and the dtype branch outside the loops: the dropped tokens' rows are zeroed after the item's first
      //  tiles are on their way -- a select or a branch join right behind a load is a wait at this point)
      if (qf32) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int j = 0; j < 8; ++j) qv[8 * cb + j] = *(const f32x4*)((const float*)p.Q + qo[cb] + 16 * j + 4 * kq);
      } else {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int t = 0; t < 4; ++t) qv[8 * cb + j][t] = load_q(p.Q, p.q_dtype, qo[cb] + 16 * j + 4 * kq + t);
      }
    }
    // ---- the next items, one step each: ordinary loads and one atomic, issued BEFORE this item's LDS-DMA traffic (they
    //      are older than every tile the counted waits below wait for) and not looked at until this item is done.  The
    //      loads whose addresses need values fetched an item ago come first, the atomic last: the wait for those values
    //      must not cover it.
    const Item D = decode(rawD);
    const int4 rowC = dyn_fetch_row(p, pidC);
    const int64_t pidD = dyn_fetch_pid(p, D.q, D.c0, lane);
    rawD = take_issue();

    Cursor F, C;
    F.init(dl, ndoc);
    C = F;
    int nissued = 0, nconsumed = 0;
    bool prev_issued = false;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const TileMap t = fill_tile<16>(F, dl, n16);
      if (t.kind != 0) {
        issue_rows<NDMA, 2, 32, false, CPOL_STREAM>(tok, (uint32_t)ROWB, 0u, wlds + j * HT, t, lane);
        ++nissued;
      }
      prev_issued = t.kind != 0;
    }
    if (q_new) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int j = 0; j < 8; ++j) qv[8 * cb + j] = q_live[cb] ? qv[8 * cb + j] : (f32x4)(0.0f);
    }
    float* const srow = p.scores + (int64_t)qi * p.ncand + c_begin;
    if (nissued == 0) {  // all padding slots / empty docs
      if (lane < ndoc) srow[lane] = (p.accum ? srow[lane] : 0.0f) + ((dl.flags & 3) == 1 ? 0.0f : NEG_INF);
    } else {
      ReducerH<NCB> red;
      red.init();
      int buf = 0;
      while (nconsumed < nissued) {
        __builtin_amdgcn_s_setprio(0);
        if (prev_issued) wait_vmcnt<NDMA * (NT - 1)>(); else wait_vmcnt<0>();
        u32x4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = *(const u32x4*)(wlds + buf * HT + n16 * ROWB + 16 * ((4 * j + kq) ^ n16));
        wait_lgkmcnt0();
        {
          const TileMap t = fill_tile<16>(F, dl, n16);
          if (t.kind != 0) {
            issue_rows<NDMA, 2, 32, false, CPOL_STREAM>(tok, (uint32_t)ROWB, 0u, wlds + buf * HT, t, lane);
            ++nissued;
          }
          prev_issued = t.kind != 0;
        }
        buf = (buf + 1 == NT) ? 0 : buf + 1;
        __builtin_amdgcn_s_setprio(3);

        f32x4 acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4)(0.0f);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, a[j])[t], qv[8 * cb + j][t], acc[cb], 0, 0, 0);
        float sv[NCB][4];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
          for (int v = 0; v < 4; ++v) sv[cb][v] = acc[cb][v];
        red.reduce_tile(sv, C, dl, lane);
        ++nconsumed;
      }
      red.drain(C, dl, lane);
      if (lane < red.jdoc) srow[lane] = (p.accum ? srow[lane] : 0.0f) + red.myscore;
      __builtin_amdgcn_s_setprio(0);
    }
    wait_lgkmcnt0();
    rawD = take_value(rawD);
    dlA = dyn_decode(p, pidB, rowB, B.nd, lane);
    A = B; B = C_; C_ = D;
    pidB = pidC; rowB = rowC;
    pidC = pidD;
  }
  // the last wave out resets the counters for the stream's next launch (every wave has stopped taking items by then)
  // (no fence: a fence is a write-back of the XCD's L2, per wave; the counters are only ever touched by atomics, which meet
  //  in L2, and a wave's increment of `finished` is issued after its last take has returned -- the loop condition needs it)
  if (lane == 0) {
    const int nwaves = (int)gridDim.x * WAVES;
    if (atomicAdd(finished, 1) == nwaves - 1) {
#pragma unroll
      for (int i = 0; i <= DYN_QUEUES; ++i) atomicExch(p.argmax + i, 0);
    }
  }
}

}  // namespace maxsim
