// libmaxsim.so -- MI355X (gfx950 / CDNA4) kernels + C ABI for the ColBERT MaxSim rerank path.
//
// What is computed (reference: wuyaoxuehun/colbert, colbert/modeling/BaseModel.py:39-46 and
// colbert/ranking/colbert_ranker.py:88-118):
//     score(q, doc) = sum_m  max_n  <Q[q,m,:], D[doc,n,:]>
// Design: one wave64 = one independent token stream.  Doc-token rows are streamed
// HBM -> LDS with LDS-DMA (global_load_lds_dwordx4, full 128-B lines, XOR-swizzled on the SOURCE
// address so the MFMA operand reads are bank-conflict free), the query tile lives in VGPRs in MFMA
// B-operand layout, the h-contraction runs on f32-input MFMA (exact f32 fmaf chain) and the max over doc
// tokens / sum over query tokens happen in registers + a 6-step cross-lane reduce.  No workgroup barrier
// exists in the hot loop: each wave owns its LDS ring and paces it with counted s_waitcnt vmcnt.
//
// gfx950 only.  No CUDA, no hipify, no dual paths.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "maxsim.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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
  int64_t n_docs;
  // queries
  const void* Q;
  const int32_t* q_len;
  const int64_t* cand;
  int nq, ncand, Lq, h;
  float* scores;
  // dense only
  const void* q_mask;
  const void* d_mask;
  int mask_dtype;
  int Ld;
  // scheduling
  int dpw;     // docs per workgroup
  int nchunk;  // ceil(ncand / dpw)
};

// Kernel arguments: the read-only tables are passed as individual `const __restrict__` pointers (not inside
// a by-value struct) so that hipcc can prove them unclobbered and fetch wave-uniform metadata with SMEM
// (s_load, lgkmcnt) instead of VMEM -- a vector load in the hot loop would share vmcnt with the LDS-DMA ring
// and drain it at every document boundary.
struct Scalars {
  int64_t n_tokens, n_docs;
  int nq, ncand, Lq, h, mask_dtype, Ld, dpw, nchunk;
};
#define KARGS_DECL                                                                                         \
  const void* __restrict__ a_index, const int64_t* __restrict__ a_tok_offsets,                              \
      const int32_t* __restrict__ a_doclens, const int32_t* __restrict__ a_pad_len,                         \
      const void* __restrict__ a_Q, const int32_t* __restrict__ a_q_len, const int64_t* __restrict__ a_cand, \
      float* __restrict__ a_scores, const void* __restrict__ a_q_mask, const void* __restrict__ a_d_mask,   \
      const Scalars sc
#define KARGS_TO_PARAMS                                                                                     \
  Params p;                                                                                                 \
  p.index = a_index; p.n_tokens = sc.n_tokens; p.tok_offsets = a_tok_offsets; p.doclens = a_doclens;        \
  p.pad_len = a_pad_len; p.n_docs = sc.n_docs; p.Q = a_Q; p.q_len = a_q_len; p.cand = a_cand;               \
  p.nq = sc.nq; p.ncand = sc.ncand; p.Lq = sc.Lq; p.h = sc.h; p.scores = a_scores; p.q_mask = a_q_mask;     \
  p.d_mask = a_d_mask; p.mask_dtype = sc.mask_dtype; p.Ld = sc.Ld; p.dpw = sc.dpw; p.nchunk = sc.nchunk
#define KARGS_PASS(p)                                                                                       \
  (p).index, (p).tok_offsets, (p).doclens, (p).pad_len, (p).Q, (p).q_len, (p).cand, (p).scores, (p).q_mask, \
      (p).d_mask, Scalars { (p).n_tokens, (p).n_docs, (p).nq, (p).ncand, (p).Lq, (p).h, (p).mask_dtype,     \
                            (p).Ld, (p).dpw, (p).nchunk }

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
template <int DT>
__device__ __forceinline__ float load_elem(const void* p, int64_t i) {
  if constexpr (DT == MAXSIM_F32) return ((const float*)p)[i];
  if constexpr (DT == MAXSIM_F16) return f16_to_f32(((const uint16_t*)p)[i]);
  return bf16_to_f32(((const uint16_t*)p)[i]);
}

// ---------------------------------------------------------------------------------------------
// One candidate slot, wave-uniform.
struct Doc {
  int64_t row0;   // first token row in the token matrix
  int nfetch;     // rows streamed (>= 1 so the pipeline always has a tile)
  int kind;       // 0 scored, 1 empty doc (score 0), 2 padding slot (score -inf)
  int floor0;     // 1: the reference padded this doc (pad_len > doclen) -> max floored at 0
};

template <int MODE>
__device__ __forceinline__ Doc load_doc(const Params& p, int qi, int c) {
  Doc d;
  if constexpr (MODE == MODE_DENSE) {
    d.row0 = (int64_t)c * p.Ld;
    d.nfetch = p.Ld;
    d.kind = 0;
    d.floor0 = 0;
  } else {
    int64_t pid = uni64(p.cand[(int64_t)qi * p.ncand + c]);
    bool ok = pid >= 0 && pid < p.n_docs;
    int64_t safe = ok ? pid : 0;
    int64_t off = uni64(p.tok_offsets[safe]);
    int len = uni(p.doclens[safe]);
    int pad = p.pad_len ? uni(p.pad_len[safe]) : len;
    // defensive: never stream outside the token matrix
    bool inb = off >= 0 && len >= 0 && off + len <= p.n_tokens;
    ok = ok && inb;
    d.kind = !ok ? 2 : (len == 0 ? 1 : 0);
    d.row0 = d.kind == 0 ? off : 0;
    d.nfetch = d.kind == 0 ? len : 1;
    d.floor0 = pad > len;
  }
  return d;
}

template <int MODE, int WAVES>
struct TileIt {
  int c, c_end, t, ntile;
  bool valid;
  Doc d;
  __device__ __forceinline__ void init(const Params& p, int qi, int c0, int cend) {
    c = c0;
    c_end = cend;
    t = 0;
    valid = c < c_end;
    if (valid) {
      d = load_doc<MODE>(p, qi, c);
      ntile = (d.nfetch + 31) >> 5;
    } else {
      ntile = 0;
      d.row0 = 0; d.nfetch = 1; d.kind = 2; d.floor0 = 0;
    }
  }
  __device__ __forceinline__ void advance(const Params& p, int qi) {
    ++t;
    if (t >= ntile) {
      c += WAVES;
      t = 0;
      valid = c < c_end;
      if (valid) {
        d = load_doc<MODE>(p, qi, c);
        ntile = (d.nfetch + 31) >> 5;
      }
    }
  }
};

// =============================================================================================
// Flagship kernel: fp32 token matrix, h = 128, Lq <= 32.  f32-input MFMA 32x32x2 (exact f32).
//
// LDS per wave: a ring of NSLOT slabs of 4 KiB.  A slab = 32 token rows x 32 dims (128 B per row, one
// full cache line per row), written by 4 LDS-DMA instructions of 8 rows each; a 32-token tile = 4 slabs.
// Source chunk j of row m is stored at chunk position j ^ ((m >> 1) & 7), which makes the ds_read_b128
// operand reads conflict-free.  The fetch pointer runs exactly NSLOT slabs ahead of the consume pointer:
// the slab fetched at step c lands in the slot that step c has just read into registers.
//
// MFMA roles: A = doc tokens (row i = lane & 31, k = lane >> 5), B = query tokens (col j = lane & 31).
// The accumulator then holds, per lane, ONE query token and 16 doc tokens, so max-over-doc-tokens is an
// in-lane max over the 16 accumulator registers plus one exchange between the two lane halves.
// k-order of the fmaf chain: for slab s, u in 0..3, t in 0..3: dims 32s+8u+t then 32s+8u+4+t.
// =============================================================================================
template <int MODE, int WAVES, int NSLOT, int ABLATE = 0>  // ABLATE (diagnostic builds only): 1 = no MFMA, 2 = no DMA
__global__ void __launch_bounds__(WAVES * 64) k_maxsim_f32_h128(KARGS_DECL) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  constexpr int ROWB = 512;            // bytes per token row
  constexpr int SLAB = 4096;           // bytes per slab
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int qi = blockIdx.x / p.nchunk;
  const int chunk = blockIdx.x - qi * p.nchunk;
  const int c_begin = chunk * p.dpw;
  const int c_end = min(p.ncand, c_begin + p.dpw);
  char* const wlds = lds + wave * (NSLOT * SLAB);

  // ---- query tile into registers, MFMA B layout -------------------------------------------------
  const int n = lane & 31, kh = lane >> 5;
  f32x4 qv[16];
  {
    int qlen = p.Lq;
    if (MODE == MODE_RERANK && p.q_len) qlen = min(qlen, p.q_len[qi]);
    const bool live = n < qlen;
    float qs = 1.0f;
    if (MODE == MODE_DENSE && live && p.mask_dtype != MAXSIM_MASK_NONE)
      qs = load_mask(p.q_mask, p.mask_dtype, (int64_t)qi * p.Lq + n);
    const float* qrow = (const float*)p.Q + ((int64_t)qi * p.Lq + (live ? n : 0)) * 128 + 4 * kh;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      f32x4 v = *(const f32x4*)(qrow + 32 * (i >> 2) + 8 * (i & 3));
      if (MODE == MODE_DENSE) v *= qs;  // Q * q_mask[..., None], BaseModel.py:42
      qv[i] = live ? v : (f32x4)(0.0f);
    }
  }

  // ---- per-lane constants -----------------------------------------------------------------------
  // DMA: instruction i covers rows 8i + (lane >> 3); chunk position lane & 7
  const int drow = lane >> 3;                     // + 8 i
  const int dchunk = lane & 7;
  // operand read: row m = lane & 31, chunks 2u + kh
  const int m = lane & 31;
  const int rsw = (m >> 1) & 7;
  int rd[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) rd[u] = m * 128 + 16 * ((2 * u + kh) ^ rsw);

  TileIt<MODE, WAVES> F, C;
  F.init(p, qi, c_begin + wave, c_end);
  C = F;

  const char* const tok = (const char*)p.index;

  // per-lane byte offsets (relative to the doc's first row) of the 4 DMA row groups of a tile
  uint32_t foff[4] = {0, 0, 0, 0};
  const char* fbase = tok;
  auto fetch_tile_setup = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int mm = 8 * i + drow;
      int r = min(F.t * 32 + mm, F.d.nfetch - 1);  // rows past the doc end re-read its last token: max unchanged
      foff[i] = (uint32_t)r * ROWB + 16u * (uint32_t)(dchunk ^ ((mm >> 1) & 7));
    }
    fbase = tok + F.d.row0 * ROWB;
  };
  auto issue_slab = [&](int fs, int slot) {
    if (ABLATE == 2) return;
    char* l = wlds + slot * SLAB;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds(GPTR(fbase + (uint32_t)(foff[i] + (uint32_t)(fs * 128))), LPTR(l + i * 1024),
                                       16, 0, 0);
  };

  // ---- prologue: NSLOT slabs in flight ------------------------------------------------------------
  bool prev_issued = false;
#pragma unroll
  for (int j = 0; j < NSLOT; ++j) {
    const int fs = j & 3;
    if (fs == 0 && j > 0 && F.valid) F.advance(p, qi);
    if (F.valid) {
      if (fs == 0) fetch_tile_setup();
      issue_slab(fs, j);
    }
    prev_issued = F.valid;
  }

  float rmax = NEG_INF;
  float myscore = 0.0f;
  int jdoc = 0;
  int slot = 0;

  while (C.valid) {
    float mv = 1.0f;
    if (MODE == MODE_DENSE && p.mask_dtype != MAXSIM_MASK_NONE) {
      int r = min(C.t * 32 + m, C.d.nfetch - 1);
      mv = load_mask(p.d_mask, p.mask_dtype, C.d.row0 + r);
    }
    f32x16 acc = (f32x16)(0.0f);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // slabs c+1 .. c+NSLOT-1 were issued after this one iff the previous step issued
      if (prev_issued) wait_vmcnt<4 * (NSLOT - 1)>(); else wait_vmcnt<0>();
      const char* sl = wlds + slot * SLAB;
      f32x4 a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *(const f32x4*)(sl + rd[u]);
      wait_lgkmcnt0();  // operands are in registers: the slot may be overwritten
      const int fs = (s + NSLOT) & 3;
      if (fs == 0 && F.valid) {
        F.advance(p, qi);
        if (F.valid) fetch_tile_setup();
      }
      if (F.valid) issue_slab(fs, slot);
      prev_issued = F.valid;
      slot = (slot + 1 == NSLOT) ? 0 : slot + 1;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (MODE == MODE_DENSE) a[u] *= mv;  // D * d_mask[..., None], BaseModel.py:41
        if (ABLATE == 1) {
          asm volatile("" ::"v"(a[u]));
          continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], qv[s * 4 + u][t], acc, 0, 0, 0);
      }
    }
    // max over the 16 doc tokens this lane holds for its query token
    float t0 = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
    float t1 = fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7]));
    float t2 = fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11]));
    float t3 = fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15]));
    rmax = fmaxf(rmax, fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)));

    if (C.t == C.ntile - 1) {  // doc finished: combine lane halves, floor, sum over query tokens
      float v = fmaxf(rmax, __shfl_xor(rmax, 32));
      if (C.d.floor0) v = fmaxf(v, 0.0f);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 1);
      float sc = C.d.kind == 0 ? v : (C.d.kind == 1 ? 0.0f : NEG_INF);
      if (lane == jdoc) myscore = sc;
      ++jdoc;
      rmax = NEG_INF;
    }
    C.advance(p, qi);
  }

  if (lane < jdoc) {
    const int c = c_begin + wave + lane * WAVES;
    p.scores[(int64_t)qi * p.ncand + c] = myscore;
  }
}

// =============================================================================================
// Tile-granular variant of the flagship kernel: a whole 32-token tile (16 KiB, CONTIGUOUS in HBM: 32
// consecutive 512-B rows) is fetched by 16 LDS-DMA instructions of two full rows each, read into registers
// in one go (64 VGPRs of A operands), and the next tile's fetch is issued into the same LDS buffer before
// the 64 MFMAs start.  Every DMA instruction reads 1 KiB of consecutive addresses, so a tile is one
// sequential 16-KiB burst (DRAM-page friendly) instead of four strided passes.
// LDS image: row-major [32][512 B]; source chunk j of row m sits at chunk position j ^ (m & 15).
// =============================================================================================
template <int MODE, int WAVES, int NT, int ABLATE = 0>
__global__ void __launch_bounds__(WAVES * 64) k_maxsim_f32_h128_t(KARGS_DECL) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  constexpr int ROWB = 512;
  constexpr int TILE = 16384;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  const int qi = blockIdx.x / p.nchunk;
  const int chunk = blockIdx.x - qi * p.nchunk;
  const int c_begin = chunk * p.dpw;
  const int c_end = min(p.ncand, c_begin + p.dpw);
  char* const wlds = lds + wave * (NT * TILE);

  const int n = lane & 31, kh = lane >> 5;
  f32x4 qv[16];
  {
    int qlen = p.Lq;
    if (MODE == MODE_RERANK && p.q_len) qlen = min(qlen, p.q_len[qi]);
    const bool live = n < qlen;
    float qs = 1.0f;
    if (MODE == MODE_DENSE && live && p.mask_dtype != MAXSIM_MASK_NONE)
      qs = load_mask(p.q_mask, p.mask_dtype, (int64_t)qi * p.Lq + n);
    const float* qrow = (const float*)p.Q + ((int64_t)qi * p.Lq + (live ? n : 0)) * 128 + 4 * kh;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      f32x4 v = *(const f32x4*)(qrow + 32 * (i >> 2) + 8 * (i & 3));
      if (MODE == MODE_DENSE) v *= qs;
      qv[i] = live ? v : (f32x4)(0.0f);
    }
  }

  // DMA instruction i: lanes 0-31 -> row 2i, lanes 32-63 -> row 2i+1; chunk position lane & 31
  const int dhalf = lane >> 5;
  const int dchunk = lane & 31;
  const int m = lane & 31;
  const int rsw = m & 15;
  const int rdbase = m * ROWB;

  TileIt<MODE, WAVES> F, C;
  F.init(p, qi, c_begin + wave, c_end);
  C = F;
  const char* const tok = (const char*)p.index;

  auto issue_tile = [&](int buf) {
    if (ABLATE == 2) return;
    const char* base = tok + F.d.row0 * ROWB;
    char* l = wlds + buf * TILE;
    const int last = F.d.nfetch - 1;
    const int r0 = F.t * 32 + dhalf;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int mm = 2 * i + dhalf;
      const int r = min(r0 + 2 * i, last);
      const uint32_t off = (uint32_t)r * ROWB + 16u * (uint32_t)(dchunk ^ (mm & 15));
      __builtin_amdgcn_global_load_lds(GPTR(base + off), LPTR(l + i * 1024), 16, 0, 0);
    }
  };

  bool prev_issued = false;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    if (j > 0 && F.valid) F.advance(p, qi);
    if (F.valid) issue_tile(j);
    prev_issued = F.valid;
  }

  float rmax = NEG_INF;
  float myscore = 0.0f;
  int jdoc = 0;
  int buf = 0;

  while (C.valid) {
    float mv = 1.0f;
    if (MODE == MODE_DENSE && p.mask_dtype != MAXSIM_MASK_NONE) {
      int r = min(C.t * 32 + m, C.d.nfetch - 1);
      mv = load_mask(p.d_mask, p.mask_dtype, C.d.row0 + r);
    }
    if (prev_issued) wait_vmcnt<16 * (NT - 1)>(); else wait_vmcnt<0>();
    const char* tl = wlds + buf * TILE + rdbase;
    f32x4 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = *(const f32x4*)(tl + 16 * ((2 * i + kh) ^ rsw));
    wait_lgkmcnt0();
    if (F.valid) {
      F.advance(p, qi);
      if (F.valid) issue_tile(buf);
    }
    prev_issued = F.valid;
    buf = (buf + 1 == NT) ? 0 : buf + 1;
    f32x16 acc = (f32x16)(0.0f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == MODE_DENSE) a[i] *= mv;
      if (ABLATE == 1) {
        asm volatile("" ::"v"(a[i]));
        continue;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], qv[i][t], acc, 0, 0, 0);
    }
    float t0 = fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3]));
    float t1 = fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7]));
    float t2 = fmaxf(fmaxf(acc[8], acc[9]), fmaxf(acc[10], acc[11]));
    float t3 = fmaxf(fmaxf(acc[12], acc[13]), fmaxf(acc[14], acc[15]));
    rmax = fmaxf(rmax, fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)));
    if (C.t == C.ntile - 1) {
      float v = fmaxf(rmax, __shfl_xor(rmax, 32));
      if (C.d.floor0) v = fmaxf(v, 0.0f);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 2);
      v += __shfl_xor(v, 1);
      float sc = C.d.kind == 0 ? v : (C.d.kind == 1 ? 0.0f : NEG_INF);
      if (lane == jdoc) myscore = sc;
      ++jdoc;
      rmax = NEG_INF;
    }
    C.advance(p, qi);
  }
  if (lane < jdoc) {
    const int c = c_begin + wave + lane * WAVES;
    p.scores[(int64_t)qi * p.ncand + c] = myscore;
  }
}

// =============================================================================================
// Generic kernel: any h / Lq / doc length / element type.  One workgroup per (query, candidate).
// Correctness path for shapes the MFMA kernels do not cover (e.g. the reference's 2x2x3 KAT).
// =============================================================================================
template <int DT, int MODE>
__global__ void __launch_bounds__(256) k_maxsim_generic(KARGS_DECL) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  float* smax = (float*)lds;  // [Lq]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int qi = blockIdx.x / p.ncand;
  const int c = blockIdx.x - qi * p.ncand;
  const Doc d = load_doc<MODE>(p, qi, c);
  const int h = p.h;
  int qlen = p.Lq;
  if (MODE == MODE_RERANK && p.q_len) qlen = min(qlen, p.q_len[qi]);
  const bool masked = MODE == MODE_DENSE && p.mask_dtype != MAXSIM_MASK_NONE;

  for (int mq = wave; mq < p.Lq; mq += 4) {
    float best = NEG_INF;
    const bool live = mq < qlen;
    float qs = 1.0f;
    if (masked) qs = load_mask(p.q_mask, p.mask_dtype, (int64_t)qi * p.Lq + mq);
    const int64_t qbase = ((int64_t)qi * p.Lq + mq) * h;
    if (live && d.kind == 0) {
      for (int nn = lane; nn < d.nfetch; nn += 64) {
        float ds = 1.0f;
        if (masked) ds = load_mask(p.d_mask, p.mask_dtype, d.row0 + nn);
        const int64_t dbase = (d.row0 + nn) * h;
        float acc = 0.0f;
        for (int k = 0; k < h; ++k) {
          float qv = (MODE == MODE_DENSE ? load_elem<DT>(p.Q, qbase + k) : ((const float*)p.Q)[qbase + k]);
          float dv = load_elem<DT>(p.index, dbase + k);
          if (masked) { qv *= qs; dv *= ds; }
          acc = fmaf(qv, dv, acc);
        }
        best = fmaxf(best, acc);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) best = fmaxf(best, __shfl_xor(best, o));
    if (d.floor0) best = fmaxf(best, 0.0f);
    if (!live) best = 0.0f;  // dropped query token contributes nothing
    if (lane == 0) smax[mq] = best;
  }
  __syncthreads();
  if (wave == 0) {
    float s = 0.0f;
    for (int mq = lane; mq < p.Lq; mq += 64) s += smax[mq];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) p.scores[(int64_t)qi * p.ncand + c] = d.kind == 0 ? s : (d.kind == 1 ? 0.0f : NEG_INF);
  }
}

// =============================================================================================
// Per-query top-k: bitonic sort of (score, position) keys in LDS.  One workgroup per query.
// key = orderable(score) << 32 | ~position  -> descending sort = score desc, position asc.
// =============================================================================================
__device__ __forceinline__ uint32_t orderable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorderable(uint32_t k) {
  uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
  return __uint_as_float(u);
}

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

// ---------------------------------------------------------------------------------------------
template <typename K>
int allow_lds(K kernel, int bytes) {
  if (bytes <= 64 * 1024) return MAXSIM_OK;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? MAXSIM_OK : MAXSIM_ELAUNCH;
}

int check_launch() { return hipGetLastError() == hipSuccess ? MAXSIM_OK : MAXSIM_ELAUNCH; }

// docs per workgroup for the streaming kernels: enough workgroups to cover 256 CUs several times over,
// few enough that the 16 KiB query-tile load per wave is amortised.
int pick_dpw(int nq, int ncand, int waves) {
  int dpw = 8 * waves;  // 8 docs per wave
  while (dpw > waves && (int64_t)nq * ((ncand + dpw - 1) / dpw) < 4096) dpw -= waves;
  return dpw;
}

template <int MODE, int WAVES, int NSLOT, int ABLATE = 0>
int launch_f32_h128_v(Params& p, hipStream_t st) {
  p.dpw = pick_dpw(p.nq, p.ncand, WAVES);
  if (const char* e = getenv("MAXSIM_DPW")) p.dpw = atoi(e) > 0 ? atoi(e) * WAVES : p.dpw;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = WAVES * NSLOT * 4096;
  auto kern = k_maxsim_f32_h128<MODE, WAVES, NSLOT, ABLATE>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}

template <int MODE, int WAVES, int NT, int ABLATE = 0>
int launch_f32_h128_t(Params& p, hipStream_t st) {
  p.dpw = pick_dpw(p.nq, p.ncand, WAVES);
  if (const char* e = getenv("MAXSIM_DPW")) p.dpw = atoi(e) > 0 ? atoi(e) * WAVES : p.dpw;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = WAVES * NT * 16384;
  auto kern = k_maxsim_f32_h128_t<MODE, WAVES, NT, ABLATE>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}

template <int MODE>
int launch_f32_h128(Params& p, hipStream_t st) {
  int v = 0;
  if (const char* e = getenv("MAXSIM_F32_VARIANT")) v = atoi(e);  // tuning knob
  switch (v) {
    case 1: return launch_f32_h128_v<MODE, 4, 3>(p, st);   // 48 KiB/WG: 3 WG/CU = 12 waves
    case 2: return launch_f32_h128_v<MODE, 4, 5>(p, st);   // 80 KiB/WG: 2 WG/CU
    case 3: return launch_f32_h128_v<MODE, 4, 8>(p, st);   // 128 KiB/WG: 1 WG/CU
    case 4: return launch_f32_h128_v<MODE, 4, 2>(p, st);   // 32 KiB/WG: 5 WG/CU (VGPR-capped at 12 waves)
    case 5: return launch_f32_h128_v<MODE, 6, 3>(p, st);   // 72 KiB/WG: 2 WG/CU = 12 waves
    case 6: return launch_f32_h128_v<MODE, 8, 4>(p, st);   // 128 KiB/WG: 1 WG/CU = 8 waves
    case 20: return launch_f32_h128_t<MODE, 4, 1>(p, st);     // tile-granular, 64 KiB/WG: 2 WG/CU = 8 waves
    case 21: return launch_f32_h128_t<MODE, 4, 2>(p, st);     // 128 KiB/WG: 4 waves/CU, 2 tiles each
    case 22: return launch_f32_h128_t<MODE, 2, 2>(p, st);     // 64 KiB/WG
    case 23: return launch_f32_h128_t<MODE, 8, 1>(p, st);
    case 24: return launch_f32_h128_t<MODE, 4, 1, 1>(p, st);  // ablation: no MFMA
    case 25: return launch_f32_h128_t<MODE, 4, 1, 2>(p, st);  // ablation: no DMA
    case 11: return launch_f32_h128_v<MODE, 4, 4, 1>(p, st);
    case 12: return launch_f32_h128_v<MODE, 4, 4, 2>(p, st);
    case 7: return launch_f32_h128_v<MODE, 4, 4>(p, st);   // slab ring, 64 KiB/WG: 2 WG/CU = 8 waves
    default: return launch_f32_h128_t<MODE, 4, 1>(p, st);  // tile-granular, 64 KiB/WG: 2 WG/CU = 8 waves
  }
}

template <int MODE>
int launch_generic(Params& p, int dt, hipStream_t st) {
  const dim3 grid((unsigned)((int64_t)p.nq * p.ncand)), block(256);
  const int ldsb = (p.Lq > 0 ? p.Lq : 1) * (int)sizeof(float);
  switch (dt) {
    case MAXSIM_F32: hipLaunchKernelGGL((k_maxsim_generic<MAXSIM_F32, MODE>), grid, block, ldsb, st, KARGS_PASS(p)); break;
    case MAXSIM_F16: hipLaunchKernelGGL((k_maxsim_generic<MAXSIM_F16, MODE>), grid, block, ldsb, st, KARGS_PASS(p)); break;
    case MAXSIM_BF16: hipLaunchKernelGGL((k_maxsim_generic<MAXSIM_BF16, MODE>), grid, block, ldsb, st, KARGS_PASS(p)); break;
    default: return MAXSIM_EINVAL;
  }
  return check_launch();
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int maxsim_version(void) { return MAXSIM_VERSION; }

const char* maxsim_strerror(int code) {
  switch (code) {
    case MAXSIM_OK: return "ok";
    case MAXSIM_EINVAL: return "invalid argument";
    case MAXSIM_EEMPTY: return "empty candidate list or empty document axis";
    case MAXSIM_ERANGE: return "size out of supported range";
    case MAXSIM_ELAUNCH: return "HIP launch failed";
    default: return "unknown error";
  }
}

int maxsim_score_dense(const void* Q, const void* D, const void* q_mask, const void* d_mask, int nq, int nd,
                       int Lq, int Ld, int h, int dtype, int mask_dtype, float* out, void* stream) {
  if (nq < 0 || nd < 0 || Lq < 0 || Ld < 0 || h < 0) return MAXSIM_EINVAL;
  if (dtype < MAXSIM_F32 || dtype > MAXSIM_BF16) return MAXSIM_EINVAL;
  if (mask_dtype < MAXSIM_MASK_NONE || mask_dtype > MAXSIM_MASK_U8) return MAXSIM_EINVAL;
  if (nq == 0 || nd == 0) return MAXSIM_OK;
  if (Ld == 0) return MAXSIM_EEMPTY;
  if (!out) return MAXSIM_EINVAL;
  if ((int64_t)nq * nd > 0x7fffffffLL) return MAXSIM_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  if (Lq == 0) {  // sum over an empty query axis
    int64_t n = (int64_t)nq * nd;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, 0.0f);
    return check_launch();
  }
  if (!Q || !D) return MAXSIM_EINVAL;
  if (mask_dtype != MAXSIM_MASK_NONE && (!q_mask || !d_mask)) return MAXSIM_EINVAL;
  Params p{};
  p.index = D;
  p.n_tokens = (int64_t)nd * Ld;
  p.n_docs = nd;
  p.Q = Q;
  p.nq = nq; p.ncand = nd; p.Lq = Lq; p.h = h;
  p.scores = out;
  p.q_mask = q_mask; p.d_mask = d_mask; p.mask_dtype = mask_dtype;
  p.Ld = Ld;
  if (dtype == MAXSIM_F32 && h == 128 && Lq <= 32) return launch_f32_h128<MODE_DENSE>(p, st);
  return launch_generic<MODE_DENSE>(p, dtype, st);
}

int maxsim_rerank(const void* index, int index_dtype, int64_t n_tokens, const int64_t* tok_offsets,
                  const int32_t* doclens, const int32_t* pad_len, int64_t n_docs, const float* Q,
                  const int32_t* q_len, const int64_t* cand_pids, int nq, int ncand, int Lq, int h,
                  float* scores, void* stream) {
  if (nq < 0 || ncand < 0 || Lq < 0 || h < 0 || n_tokens < 0 || n_docs < 0) return MAXSIM_EINVAL;
  if (index_dtype < MAXSIM_F32 || index_dtype > MAXSIM_BF16) return MAXSIM_EINVAL;
  if (ncand == 0) return MAXSIM_EEMPTY;  // assert len(pids) > 0, colbert_ranker.py:76
  if (nq == 0) return MAXSIM_OK;
  if (!scores || !cand_pids || !tok_offsets || !doclens || !Q) return MAXSIM_EINVAL;
  if (n_tokens > 0 && !index) return MAXSIM_EINVAL;
  if ((int64_t)nq * ncand > 0x7fffffffLL) return MAXSIM_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  Params p{};
  p.index = index;
  p.n_tokens = n_tokens;
  p.tok_offsets = tok_offsets;
  p.doclens = doclens;
  p.pad_len = pad_len;
  p.n_docs = n_docs;
  p.Q = Q;
  p.q_len = q_len;
  p.cand = cand_pids;
  p.nq = nq; p.ncand = ncand; p.Lq = Lq; p.h = h;
  p.scores = scores;
  p.mask_dtype = MAXSIM_MASK_NONE;
  if (index_dtype == MAXSIM_F32 && h == 128 && Lq >= 1 && Lq <= 32 && n_tokens > 0)
    return launch_f32_h128<MODE_RERANK>(p, st);
  return launch_generic<MODE_RERANK>(p, index_dtype, st);
}

int maxsim_topk(const float* scores, const int64_t* pids, int nq, int ncand, int k, float* out_scores,
                int64_t* out_pids, void* stream) {
  if (nq < 0 || ncand < 0 || k < 1) return MAXSIM_EINVAL;
  if (ncand == 0) return MAXSIM_EEMPTY;
  if (ncand > 16384) return MAXSIM_ERANGE;
  if (nq == 0) return MAXSIM_OK;
  if (!scores || !out_scores || !out_pids) return MAXSIM_EINVAL;
  int P = 2;
  while (P < ncand) P <<= 1;
  const int ldsb = P * 8;
  int rc = allow_lds(k_topk, ldsb);
  if (rc) return rc;
  int threads = P / 2 < 64 ? 64 : (P / 2 > 1024 ? 1024 : P / 2);
  hipLaunchKernelGGL(k_topk, dim3((unsigned)nq), dim3(threads), ldsb, (hipStream_t)stream, scores, pids, ncand, k,
                     P, out_scores, out_pids);
  return check_launch();
}

}  // extern "C"
