// maxsim_common.h -- shared device-side definitions of libmaxsim (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "maxsim.h"

namespace maxsim {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr float NEG_INF = -__builtin_huge_valf();

enum : int { MODE_RERANK = 0, MODE_DENSE = 1 };

struct Params {
  // token matrix: the HBM-resident index (rerank) or D[nd,Ld,h] (dense)
  const void* index;
  int64_t n_tokens;
  const int64_t* tok_offsets;
  const int32_t* doclens;
  const int32_t* pad_len;
  const void* doc_table;  // optional packed descriptor rows {int64 first_row, int32 len, int32 pad_len} (maxsim_build_doc_table)
  int64_t n_docs;
  // queries
  const void* Q;
  int q_dtype;  // element type of Q (rerank: MAXSIM_F32 / F16 / BF16; dense: same as the token matrix)
  int q_tok0;   // first query token handled by this launch (queries longer than 32 tokens take one launch per 32)
  int accum;    // 1: add this launch's partial sums to `scores` (passes after the first)
  const int32_t* q_len;
  const int64_t* cand;
  int nq, ncand, Lq, h;
  float* scores;
  int32_t* argmax;  // dense training-form forward only: [nq, nd, Lq] arg-max doc token per query token, or NULL
  // dense: multiplicative masks of element type mask_dtype.  rerank: q_mask is NULL or a uint8 [nq, Lq] keep-predicate
  // (the reference's keep_nonzero, training_utils.py:48-53), d_mask unused
  const void* q_mask;
  const void* d_mask;
  int mask_dtype;
  int Ld;
  // scheduling
  int dpw;     // docs per workgroup
  int nchunk;  // ceil(ncand / dpw)
  int split;   // split-doc kernels: waves per doc (2 or 4); otherwise 1
  // dense work list (maxsim_worklist.h; counted candidate rows -- doc shards, ANN lists): NULL, or the device-built list of
  // (query, first slot, docs) wave items the LIST kernels walk instead of the static (query, chunk) grid
  const void* worklist;
  int uniform_len;  // host side only (launch choice): > 0 = every doc has exactly this many tokens, none is padded
};

// Kernel arguments: the read-only tables are passed as individual `const __restrict__` pointers (not inside
// a by-value struct) so that hipcc can prove them unclobbered and fetch wave-uniform metadata with SMEM
// (s_load, lgkmcnt) instead of VMEM -- a vector load in the hot loop would share vmcnt with the LDS-DMA
// stream and drain it at every document boundary.
struct Scalars {
  int64_t n_tokens, n_docs;
  int nq, ncand, Lq, h, mask_dtype, Ld, dpw, nchunk, q_dtype, q_tok0, accum, split;
};
#define KARGS_DECL                                                                                         \
  const void* __restrict__ a_index, const int64_t* __restrict__ a_tok_offsets,                              \
      const int32_t* __restrict__ a_doclens, const int32_t* __restrict__ a_pad_len,                         \
      const void* __restrict__ a_Q, const int32_t* __restrict__ a_q_len, const int64_t* __restrict__ a_cand, \
      float* __restrict__ a_scores, const void* __restrict__ a_q_mask, const void* __restrict__ a_d_mask,   \
      int32_t* __restrict__ a_argmax, const void* __restrict__ a_doc_table, const void* __restrict__ a_worklist, \
      const maxsim::Scalars sc
#define KARGS_TO_PARAMS                                                                                     \
  maxsim::Params p;                                                                                         \
  p.index = a_index; p.n_tokens = sc.n_tokens; p.tok_offsets = a_tok_offsets; p.doclens = a_doclens;        \
  p.pad_len = a_pad_len; p.n_docs = sc.n_docs; p.Q = a_Q; p.q_len = a_q_len; p.cand = a_cand;               \
  p.nq = sc.nq; p.ncand = sc.ncand; p.Lq = sc.Lq; p.h = sc.h; p.scores = a_scores; p.q_mask = a_q_mask;     \
  p.d_mask = a_d_mask; p.mask_dtype = sc.mask_dtype; p.Ld = sc.Ld; p.dpw = sc.dpw; p.nchunk = sc.nchunk;         \
  p.q_dtype = sc.q_dtype; p.argmax = a_argmax; p.q_tok0 = sc.q_tok0; p.accum = sc.accum; p.doc_table = a_doc_table;       \
  p.split = sc.split; p.worklist = a_worklist
#define KARGS_PASS(p)                                                                                       \
  (p).index, (p).tok_offsets, (p).doclens, (p).pad_len, (p).Q, (p).q_len, (p).cand, (p).scores, (p).q_mask, \
      (p).d_mask, (p).argmax, (p).doc_table, (p).worklist, maxsim::Scalars { (p).n_tokens, (p).n_docs, (p).nq, (p).ncand, (p).Lq, (p).h,             \
                                    (p).mask_dtype, (p).Ld, (p).dpw, (p).nchunk, (p).q_dtype, (p).q_tok0, (p).accum,       \
                                    (p).split }

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkmcnt0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

__device__ __forceinline__ float load_mask(const void* m, int mask_dtype, int64_t i) {
  switch (mask_dtype) {
    case MAXSIM_MASK_I64: return (float)((const int64_t*)m)[i];
    case MAXSIM_MASK_I32: return (float)((const int32_t*)m)[i];
    case MAXSIM_MASK_F32: return ((const float*)m)[i];
    case MAXSIM_MASK_U8: return (float)((const uint8_t*)m)[i];
    default: return 1.0f;
  }
}

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ float f16_to_f32(uint16_t b) {
  _Float16 h;
  __builtin_memcpy(&h, &b, 2);
  return (float)h;
}
__device__ __forceinline__ uint16_t f32_to_bf16_rn(float f) {  // finite inputs (L2-normalised embeddings)
  uint32_t u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float load_q(const void* Q, int q_dtype, int64_t i) {
  switch (q_dtype) {
    case MAXSIM_F16: return f16_to_f32(((const uint16_t*)Q)[i]);
    case MAXSIM_BF16: return bf16_to_f32(((const uint16_t*)Q)[i]);
    default: return ((const float*)Q)[i];
  }
}
template <int DT>
__device__ __forceinline__ float load_elem(const void* p, int64_t i) {
  if constexpr (DT == MAXSIM_F32) return ((const float*)p)[i];
  if constexpr (DT == MAXSIM_F16) return f16_to_f32(((const uint16_t*)p)[i]);
  return bf16_to_f32(((const uint16_t*)p)[i]);
}

// Doc metadata of one pid: first token row, length, bucket stride.  With the packed table this is ONE 16-byte load
// (one random cache line per candidate instead of three -- it matters when a doc is a few KB, e.g. the 8-token
// multi-view config, and it shortens a small launch's start-up).
struct DocMeta {
  int64_t off;
  int len, pad;
};
__device__ __forceinline__ DocMeta load_doc_meta(const Params& p, int64_t pid) {
  DocMeta m;
  if (p.doc_table) {
    const int4 r = ((const int4*)p.doc_table)[pid];
    m.off = (int64_t)(((uint64_t)(uint32_t)r.y << 32) | (uint32_t)r.x);
    m.len = r.z;
    m.pad = r.w;
  } else {
    m.off = p.tok_offsets[pid];
    m.len = p.doclens[pid];
    m.pad = p.pad_len ? p.pad_len[pid] : m.len;
  }
  return m;
}

// Is query token `qtok` of query `qi` scored?  Rerank: below q_len and kept by the uint8 q_mask predicate (both
// optional).  A dropped token is a zero query row: every similarity is exactly 0, so it adds 0 to the sum -- the same
// value the reference gets by removing the token before search() (keep_nonzero, training_utils.py:48-53).
template <int MODE>
__device__ __forceinline__ bool q_token_live(const Params& p, int qi, int qtok, int qlen) {
  bool live = qtok < qlen;
  if constexpr (MODE == MODE_RERANK) {
    if (p.q_mask) live = live && ((const uint8_t*)p.q_mask)[(int64_t)qi * p.Lq + (live ? qtok : 0)] != 0;
  }
  return live;
}

// Workgroup id -> (query [block], candidate chunk).  Queries are taken in groups of up to 64; inside a group the id runs
// over the queries first and the chunks second: ids (chunk 0 of 64 queries), (chunk 1 of the same 64), ...
// Why not simply id = query * nchunk + chunk: the hardware deals consecutive ids round-robin over the 8 XCDs and their
// CUs.  When only the FIRST chunks of every row hold real candidates -- a doc-sharded row after maxsim_shard_candidates:
// ~1/N of the slots live, the rest padding that retires at once -- and nchunk is a multiple of 8, query-major ids put
// every live workgroup on the same few XCDs / CUs (measured at N = 8: 27 ms instead of 4.2 ms for the same work).  Here
// a group's live workgroups are >= 64 consecutive ids, which spread evenly.  A group's 64 query tiles (1 MiB) stay in L2.
__device__ __forceinline__ void wg_to_work(int id, int nq, int nchunk, int& q, int& chunk) {
  constexpr int G = 64;
  const int per_group = G * nchunk;
  const int group = id / per_group;
  const int rem = id - group * per_group;
  const int gcur = min(G, nq - group * G);  // the last group may be smaller
  chunk = rem / gcur;
  q = group * G + (rem - chunk * gcur);
}

// ---------------------------------------------------------------------------------------------
// One candidate slot, wave-uniform.
struct Doc {
  int64_t row0;  // first token row in the token matrix
  int len;       // tokens to score (0 for kind != 0)
  int kind;      // 0 scored, 1 empty doc (score 0), 2 padding slot (score -inf)
  int floor0;    // 1: the reference padded this doc (pad_len > doclen) -> max floored at 0
};

template <int MODE>
__device__ __forceinline__ Doc load_doc(const Params& p, int qi, int c) {
  Doc d;
  if constexpr (MODE == MODE_DENSE) {
    d.row0 = (int64_t)c * p.Ld;
    d.len = p.Ld;
    d.kind = 0;
    d.floor0 = 0;
  } else {
    int64_t pid = uni64(p.cand[(int64_t)qi * p.ncand + c]);
    bool ok = pid >= 0 && pid < p.n_docs;
    int64_t safe = ok ? pid : 0;
    const DocMeta dm = load_doc_meta(p, safe);
    int64_t off = uni64(dm.off);
    int len = uni(dm.len);
    int pad = uni(dm.pad);
    // defensive: never stream outside the token matrix
    bool inb = off >= 0 && len >= 0 && off + len <= p.n_tokens;
    ok = ok && inb;
    d.kind = !ok ? 2 : (len == 0 ? 1 : 0);
    d.row0 = d.kind == 0 ? off : 0;
    d.len = d.kind == 0 ? len : 0;
    d.floor0 = pad > len;
  }
  return d;
}

}  // namespace maxsim
