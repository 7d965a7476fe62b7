// maxsim_stream.h -- the MFMA streaming kernel of libmaxsim: h = 128, Lq <= 32, fp32 / fp16 / bf16 token matrix.
//
// One wave64 = one independent TOKEN STREAM: the tokens of the wave's candidate docs laid end to end.  The stream
// is cut into 32-row tiles that may straddle document boundaries ("packed tiles": no MFMA work is spent on
// padding except in the wave's very last tile).  Per tile:
//   fetch    NDMA LDS-DMA instructions (global_load_lds_dwordx4, non-temporal in rerank mode -- the bytes are read
//            once and the board is power-capped, so they skip cache allocation; 1 KiB each = whole token rows; a tile that lies
//            inside one doc is one contiguous burst addressed SGPR-base + lane offset; a tile that crosses docs
//            takes each row's address from the lane that owns the slot via ds_bpermute) into the wave's private
//            LDS ring of NT tiles.  Source chunk j of row slot m is stored at chunk position j ^ (m & 15): the
//            LDS-DMA destination stays linear (it must), the ds_read_b128 operand reads are bank-conflict free.
//   compute  the whole tile goes to registers (A operands), the NEXT tile's fetch is issued into the freed
//            buffer, then the contraction runs on MFMA with fp32 accumulation:
//              fp32 index: v_mfma_f32_16x16x4_f32 (rerank: the query tokens as one or two 16-column blocks) or
//                          v_mfma_f32_32x32x2_f32 (dense) -- an exact k-ordered fp32 fmaf chain (bitwise reproducible);
//              fp16 index: v_mfma_f32_16x16x32_f16 (rerank) / 32x32x16 (other), Q = Qhi + 2^-11 Qlo (two accumulators);
//              bf16 index: v_mfma_f32_16x16x32_bf16 (rerank) / 32x32x16, Q = Q0 + Q1 + Q2.
//            A = doc tokens (rows), B = query tokens (columns): a lane's accumulators are doc tokens of ONE query
//            token, so max-over-doc-tokens is in-lane + an exchange between the lane groups holding that token.
//   reduce   per document segment of the tile: (masked) max into the doc's running max; when a doc ends:
//            halves exchange, 0-floor, DPP pairwise-tree sum over query tokens, score kept in lane (doc ordinal).
// There is no workgroup barrier: waves pace their own ring with counted s_waitcnt vmcnt.
#pragma once
#include "maxsim_common.h"
#include "maxsim_worklist.h"

namespace maxsim {

template <int DT>
struct StreamTraits;
template <>
struct StreamTraits<MAXSIM_F32> {
  static constexpr int ROWB = 512, TILE = 16384, NDMA = 16, RPD = 2, LPR = 32, NRD = 16, NP = 1;
};
// fp32 storage, "fast" contraction: both operands are split on the fly into fp16 pieces (x = hi + 2^-11 lo) and the
// products hi*hi, hi*lo, lo*hi run on the 16-bit matrix pipe (the 2^-22 lo*lo term is dropped): |error| ~ 2^-22
// relative per product (~1e-6 on a score), 3x less matrix-pipe time than the exact f32 MFMA.  Needs |x| < 65504
// (L2-normalised embeddings).  Opt-in: index_dtype MAXSIM_F32_FAST.
constexpr int F32S = MAXSIM_F32_FAST;
template <>
struct StreamTraits<F32S> {
  static constexpr int ROWB = 512, TILE = 16384, NDMA = 16, RPD = 2, LPR = 32, NRD = 16, NP = 2;
};
// fp32 storage, "3 x bf16" contraction: every fp32 value is cut EXACTLY into three bf16 pieces by truncation
// (x = t0 + t1 + t2, 8 + 8 + 8 significant bits; bf16 has the fp32 exponent range, so no magnitude restriction), and the
// six piece products of order <= 2 (t0q0, t0q1, t1q0, t0q2, t1q1, t2q0) run on the bf16 matrix pipe with fp32
// accumulation.  Piece products are exact in fp32; the dropped terms are <= 3 * 2^-24 relative -- fp32-class accuracy
// (like any fp32 GEMM it differs from a sequential fmaf chain only in rounding/summation order), 2.7x less matrix time.
constexpr int F32X = MAXSIM_F32_BF16X3;
template <>
struct StreamTraits<F32X> {
  static constexpr int ROWB = 512, TILE = 16384, NDMA = 16, RPD = 2, LPR = 32, NRD = 16, NP = 3;
};
template <>
struct StreamTraits<MAXSIM_F16> {
  static constexpr int ROWB = 256, TILE = 8192, NDMA = 8, RPD = 4, LPR = 16, NRD = 8, NP = 2;
};
template <>
struct StreamTraits<MAXSIM_BF16> {
  static constexpr int ROWB = 256, TILE = 8192, NDMA = 8, RPD = 4, LPR = 16, NRD = 8, NP = 3;
};

// The wave's candidate docs are described ONCE, up front, in "descriptor lanes": lane j holds the first token row,
// the length and the flags of the wave's j-th doc (at most 64 docs per wave), gathered with ordinary vector loads
// before the LDS-DMA stream starts.  Walking the docs afterwards is three v_readlane per doc: no memory access, hence
// no load latency (and no vmcnt/lgkmcnt traffic) inside the streaming loop.
struct DocLanes {
  uint32_t row0;  // first token row
  int len;        // tokens to score (0: empty doc or padding slot)
  int flags;      // kind (bits 0-1: 0 scored, 1 empty doc -> 0, 2 padding slot -> -inf) | floor0 << 2
};

template <int MODE>
__device__ __forceinline__ DocLanes load_doc_lanes(const Params& p, int qi, int c0, int ndoc, int lane) {
  DocLanes d;
  d.row0 = 0; d.len = 0; d.flags = 2;
  if (lane < ndoc) {
    const int c = c0 + lane;
    if constexpr (MODE == MODE_DENSE) {
      d.row0 = (uint32_t)((int64_t)c * p.Ld);
      d.len = p.Ld;
      d.flags = 0;
    } else {
      const int64_t pid = p.cand[(int64_t)qi * p.ncand + c];
      bool ok = pid >= 0 && pid < p.n_docs;
      const int64_t safe = ok ? pid : 0;
      const DocMeta dm = load_doc_meta(p, safe);
      const int64_t off = dm.off;
      const int len = dm.len, pad = dm.pad;
      ok = ok && off >= 0 && len >= 0 && off + len <= p.n_tokens;  // defensive: never stream outside the matrix
      const int kind = !ok ? 2 : (len == 0 ? 1 : 0);
      d.row0 = kind == 0 ? (uint32_t)off : 0u;
      d.len = kind == 0 ? len : 0;
      d.flags = kind | ((pad > len) ? 4 : 0);
    }
  }
  return d;
}

// Wave-uniform cursor over the wave's docs.
struct Cursor {
  int j, ndoc, pos, len, kind, floor0;
  uint32_t row0;
  bool valid;
  // (all fields are assigned unconditionally: conditional stores to different fields get merged by the optimizer
  //  into a store through a selected pointer, which pins the whole cursor in scratch memory)
  __device__ __forceinline__ void load(const DocLanes& d) {
    valid = j < ndoc;
    const int jj = valid ? j : 0;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)d.row0, jj);
    const int ln = __builtin_amdgcn_readlane(d.len, jj);
    const int fl = __builtin_amdgcn_readlane(d.flags, jj);
    row0 = valid ? r0 : 0u;
    len = valid ? ln : 0;
    kind = valid ? (fl & 3) : 2;
    floor0 = valid ? (fl >> 2) : 0;
    pos = 0;
  }
  __device__ __forceinline__ void init(const DocLanes& d, int nd) {
    j = 0;
    ndoc = nd;
    load(d);
  }
  __device__ __forceinline__ void next_doc(const DocLanes& d) {
    ++j;
    load(d);
  }
};

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {  // v as seen through the DPP lane permutation CTRL (all lanes on)
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false));
}

// Where the 32 row slots of a tile come from.
struct TileMap {
  uint32_t myrow;         // token-matrix row of slot (lane & 31)
  uint32_t base0, base1;  // first / second document segment: slot s -> row base + s
  int split;              // first slot of the second segment
  int kind;               // 0: stream exhausted; 1: 32 consecutive rows of one doc; 2: two segments; 3: anything else
};

// Lays the next 32 stream rows onto the tile's row slots.  Slots past the stream end repeat the last real row (a
// duplicate cannot change a max).
template <int ROWS = 32>
__device__ __forceinline__ TileMap fill_tile(Cursor& F, const DocLanes& dl, int r) {
  int filled = 0, nseg = 0, split = ROWS;
  uint32_t last = 0, myrow = 0, base0 = 0, base1 = 0;
  while (filled < ROWS && F.valid) {
    const int take = uni(min(ROWS - filled, max(F.len - F.pos, 0)));  // 0: empty doc / padding slot, just skipped
    const uint32_t base = F.row0 + (uint32_t)F.pos - (uint32_t)filled;  // slot r -> row base + r
    base0 = (take > 0 && nseg == 0) ? base : base0;
    base1 = (take > 0 && nseg == 1) ? base : base1;
    split = (take > 0 && nseg == 1) ? filled : split;
    nseg += take > 0 ? 1 : 0;
    const bool in = (r >= filled) & (r < filled + take);
    myrow = in ? base + (uint32_t)r : myrow;
    last = take > 0 ? base + (uint32_t)(filled + take - 1) : last;
    filled += take;
    F.pos += take;
    if (F.pos >= F.len) F.next_doc(dl);
  }
  TileMap t;
  t.myrow = (filled > 0 && r >= filled) ? last : myrow;
  t.base0 = base0;
  t.base1 = base1;
  t.split = split;
  t.kind = filled == 0 ? 0 : ((filled == ROWS && nseg == 1) ? 1 : ((filled == ROWS && nseg == 2) ? 2 : 3));
  return t;
}

// Issues the NDMA LDS-DMA instructions of one tile (or of one 128-dim block of it): instruction i moves the RPD row
// slots RPD*i + lane/LPR, 1 KiB in all, to l + 1024 i.  `rowbytes` = bytes per token row, `blk` = byte offset of the
// block inside the row.  Chunk position p of slot m receives source chunk p ^ (m & 15).
// PART: the row's last 128-dim block may be partial (h not a multiple of 128): chunks past the row end are redirected
// to the row's first chunk (valid, finite data; the query image is zero there, so they contribute exactly 0).
// CPOL: cache-policy bits of the loads (gfx950: 1 = sc0, 2 = nt, 16 = sc1).  Rerank streams every doc once per
// (query, candidate) -- nothing to keep in L2 / Infinity Cache -- and is power-capped, so it loads non-temporal
// (CPOL_STREAM: 2 % faster on C2); the all-pairs kernels re-read D from cache and keep the default policy.
constexpr int CPOL_STREAM = 2;
template <int NDMA, int RPD, int LPR, bool PART = false, int CPOL = 0>
__device__ __forceinline__ void issue_rows(const char* tok, uint32_t rowbytes, uint32_t blk, char* l,
                                           const TileMap& t, int lane) {
  // (the lane constants are made opaque so that hipcc recomputes the 2-3 VALU ops per instruction instead of keeping
  //  NDMA loop-invariant offsets alive in VGPRs across the whole tile loop)
  int ds0 = lane / LPR, dch = lane % LPR;
  asm volatile("" : "+v"(ds0), "+v"(dch));
  if (t.kind == 1) {  // one contiguous burst: uniform base in SGPRs + per-lane 32-bit offset
    const char* base = tok + (uint64_t)t.base0 * rowbytes + (PART ? 0u : blk);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int slot = RPD * i + ds0;
      uint32_t inrow = 16u * (uint32_t)(dch ^ (slot & 15));
      if (PART) inrow = (blk + inrow < rowbytes) ? blk + inrow : 0u;  // past the row end -> byte 0 of the row
      const uint32_t off = (uint32_t)slot * rowbytes + inrow;
      __builtin_amdgcn_global_load_lds(GPTR(base + off), LPTR(l + i * 1024), 16, 0, CPOL);
    }
  } else if (t.kind == 2) {  // the end of one doc and the start of the next: two uniform bases, selected per lane
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int slot = RPD * i + ds0;
      const uint32_t row = (slot < t.split ? t.base0 : t.base1) + (uint32_t)slot;
      uint32_t inrow = blk + 16u * (uint32_t)(dch ^ (slot & 15));
      if (PART && inrow >= rowbytes) inrow = 0u;
      const char* g = tok + (uint64_t)row * rowbytes + inrow;
      __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(l + i * 1024), 16, 0, CPOL);
    }
  } else {  // many short docs (or the stream's padded last tile): each slot's row comes from the lane that owns it
    uint32_t rows[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) rows[i] = (uint32_t)__shfl((int)t.myrow, RPD * i + ds0);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int slot = RPD * i + ds0;
      uint32_t inrow = blk + 16u * (uint32_t)(dch ^ (slot & 15));
      if (PART && inrow >= rowbytes) inrow = 0u;
      const char* g = tok + (uint64_t)rows[i] * rowbytes + inrow;
      __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(l + i * 1024), 16, 0, CPOL);
    }
  }
}

// Per-wave reduction state: running max of the current doc (per lane: one query token, one lane half) and the
// finished docs' scores parked one per lane.
struct Reducer {
  float rmax, myscore;
  int jdoc;
  __device__ __forceinline__ void init() {
    rmax = NEG_INF;
    myscore = 0.0f;
    jdoc = 0;
  }
  // C's current doc is complete: exchange the lane halves (v_permlane32_swap), 0-floor, then sum the 32 query-token
  // lanes with DPP adds -- a pairwise tree ((q0+q1)+(q2+q3))+... in VALU registers, no LDS round trips (a
  // ds_bpermute butterfly costs ~6 dependent LDS latencies per doc, which dominates when docs are a few tokens long).
  __device__ __forceinline__ void finish_doc(const Cursor& C, int lane) {
    float sc;
    if (C.kind == 0) {
      const uint32_t xb = __float_as_uint(rmax);
      const auto sw = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
      float v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      if (C.floor0) v = fmaxf(v, 0.0f);
      v += dpp_f32<0xB1>(v);   // quad_perm [1,0,3,2]
      v += dpp_f32<0x4E>(v);   // quad_perm [2,3,0,1]
      v += dpp_f32<0x141>(v);  // row_half_mirror: 8-lane sums
      v += dpp_f32<0x140>(v);  // row_mirror: 16-lane sums
      // rows 0 and 1 hold the two halves of the 32 query tokens (rows 2, 3 mirror them)
      sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0)) +
           __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
    } else {
      sc = C.kind == 1 ? 0.0f : NEG_INF;
    }
    myscore = (lane == jdoc) ? sc : myscore;
    ++jdoc;
    rmax = NEG_INF;
  }
  // Walks the document segments of one finished tile.  sv[v] = similarity of this lane's query token with tile row
  // (v & 3) + 8 (v >> 2) + 4 (lane >> 5).
  __device__ __forceinline__ void reduce_tile(const float (&sv)[16], Cursor& C, const DocLanes& dl, int lane) {
    const int hh = lane >> 5;
    int filled = 0;
    while (filled < 32 && C.valid) {
      const int take = uni(min(32 - filled, max(C.len - C.pos, 0)));  // 0: empty doc / padding slot
      if (take == 32) {
        float t0 = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        float t1 = fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7]));
        float t2 = fmaxf(fmaxf(sv[8], sv[9]), fmaxf(sv[10], sv[11]));
        float t3 = fmaxf(fmaxf(sv[12], sv[13]), fmaxf(sv[14], sv[15]));
        rmax = fmaxf(rmax, fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)));
      } else if (take > 0 && ((filled | take) & 7) == 0) {
        // segment made of whole 8-row groups (e.g. the 8-token multi-view docs): group g = rows 8g..8g+7 is exactly
        // accumulators 4g..4g+3 of both lane halves -> no per-row masking, wave-uniform group selection
        float m = NEG_INF;
        const int g0 = filled >> 3, g1 = (filled + take) >> 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float mg = fmaxf(fmaxf(sv[4 * g], sv[4 * g + 1]), fmaxf(sv[4 * g + 2], sv[4 * g + 3]));
          m = (g >= g0 && g < g1) ? fmaxf(m, mg) : m;
        }
        rmax = fmaxf(rmax, m);
      } else if (take > 0) {  // rows [filled, filled + take) only
        float m = NEG_INF;
        const uint32_t lo = (uint32_t)(filled - 4 * hh), n_in = (uint32_t)take;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const uint32_t rel = (uint32_t)((v & 3) + 8 * (v >> 2)) - lo;
          m = fmaxf(m, rel < n_in ? sv[v] : NEG_INF);
        }
        rmax = fmaxf(rmax, m);
      }
      filled += take;
      C.pos += take;
      if (C.pos >= C.len) {  // doc complete (empty docs / padding slots are scored on the spot)
        finish_doc(C, lane);
        C.next_doc(dl);
      }
    }
  }
  __device__ __forceinline__ void drain(Cursor& C, const DocLanes& dl, int lane) {  // trailing empty docs / padding
    while (C.valid) {
      finish_doc(C, lane);
      C.next_doc(dl);
    }
  }
};

// Reducer for the 16-column accumulator layout of v_mfma_f32_16x16x4_f32 (Lq <= 16): lane = query token (lane & 15)
// x row quarter g = lane >> 4; sv[4 b + v] = similarity with tile row 16 b + 4 g + v.
struct Reducer16 {
  float rmax, myscore;
  int jdoc;
  __device__ __forceinline__ void init() {
    rmax = NEG_INF;
    myscore = 0.0f;
    jdoc = 0;
  }
  __device__ __forceinline__ void finish_doc(const Cursor& C, int lane) {
    float sc;
    if (C.kind == 0) {
      float v = fmaxf(rmax, __shfl_xor(rmax, 16));
      const uint32_t xb = __float_as_uint(v);
      const auto sw = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
      v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      if (C.floor0) v = fmaxf(v, 0.0f);
      v += dpp_f32<0xB1>(v);
      v += dpp_f32<0x4E>(v);
      v += dpp_f32<0x141>(v);
      v += dpp_f32<0x140>(v);  // sum over the 16 query-token lanes of a row
      sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0));
    } else {
      sc = C.kind == 1 ? 0.0f : NEG_INF;
    }
    myscore = (lane == jdoc) ? sc : myscore;
    ++jdoc;
    rmax = NEG_INF;
  }
  __device__ __forceinline__ void reduce_tile(const float (&sv)[8], Cursor& C, const DocLanes& dl, int lane) {
    const int g4 = 4 * (lane >> 4);
    int filled = 0;
    while (filled < 32 && C.valid) {
      const int take = uni(min(32 - filled, max(C.len - C.pos, 0)));
      if (take == 32) {
        const float t0 = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
        const float t1 = fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7]));
        rmax = fmaxf(rmax, fmaxf(t0, t1));
      } else if (take > 0) {
        float m = NEG_INF;
        const uint32_t lo = (uint32_t)(filled - g4), n_in = (uint32_t)take;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const uint32_t rel = (uint32_t)(16 * (v >> 2) + (v & 3)) - lo;
          m = fmaxf(m, rel < n_in ? sv[v] : NEG_INF);
        }
        rmax = fmaxf(rmax, m);
      }
      filled += take;
      C.pos += take;
      if (C.pos >= C.len) {
        finish_doc(C, lane);
        C.next_doc(dl);
      }
    }
  }
  __device__ __forceinline__ void drain(Cursor& C, const DocLanes& dl, int lane) {
    while (C.valid) {
      finish_doc(C, lane);
      C.next_doc(dl);
    }
  }
};

// Reducer for 32 query tokens held as TWO 16-column blocks of v_mfma_f32_16x16x4_f32 (QT_2X16): lane = query token
// 16 cb + (lane & 15) of block cb x row quarter g = lane >> 4; sv[cb][4 b + v] = similarity with tile row 16 b + 4 g + v.
// SPLITK (small launches): a doc is streamed by several waves of the workgroup, each taking a slice of its tokens; when a
// slice ends the wave parks its 32 per-query-token maxima in LDS (`part`: [doc ordinal][32] floats of this wave) instead
// of finishing the score; the workgroup combines the slices after the stream (combine_split in the kernel).
template <bool SPLITK>
struct Reducer2x16T {
  float rmax0, rmax1, myscore;
  int jdoc;
  float* part;
  __device__ __forceinline__ void init() {
    rmax0 = rmax1 = NEG_INF;
    myscore = 0.0f;
    jdoc = 0;
    part = nullptr;
  }
  // per-query-token maximum of the finished doc (slice): lanes 0..15 = tokens 0..15, lanes 32..47 = tokens 16..31
  // (lanes 16..31 / 48..63 mirror them)
  __device__ __forceinline__ float token_max() const {
    // quarters g, g + 2 (lane halves): one swap leaves block 0 in the lower half, block 1 in the upper half
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(rmax0), __float_as_uint(rmax1), false, false);
    float v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    // quarters g, g + 1 (adjacent rows of 16 lanes)
    const uint32_t xb = __float_as_uint(v);
    const auto s16 = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
    return fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
  }
  // 0-floor, then the sum over the 32 query tokens as a fixed pairwise tree (the same tree whether the maxima come
  // straight from the accumulators or from the parked slices: split and unsplit launches are bit-identical)
  static __device__ __forceinline__ float floor_and_sum(float v, int floor0) {
    if (floor0) v = fmaxf(v, 0.0f);
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);  // 16-lane row sums: row 0 = query tokens 0..15, row 2 = tokens 16..31
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0)) +
           __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 32));
  }
  __device__ __forceinline__ void finish_doc(const Cursor& C, int lane) {
    if constexpr (SPLITK) {
      if (C.kind == 0) {
        const float v = token_max();
        if ((lane & 16) == 0) part[jdoc * 32 + (lane & 15) + ((lane >> 5) << 4)] = v;
      }
    } else {
      float sc;
      if (C.kind == 0) {
        sc = floor_and_sum(token_max(), C.floor0);
      } else {
        sc = C.kind == 1 ? 0.0f : NEG_INF;
      }
      myscore = (lane == jdoc) ? sc : myscore;
    }
    ++jdoc;
    rmax0 = rmax1 = NEG_INF;
  }
  __device__ __forceinline__ void reduce_tile(const float (&sv)[2][8], Cursor& C, const DocLanes& dl, int lane) {
    const int g4 = 4 * (lane >> 4);
    int filled = 0;
    while (filled < 32 && C.valid) {
      const int take = uni(min(32 - filled, max(C.len - C.pos, 0)));
      if (take == 32) {
        rmax0 = fmaxf(rmax0, fmaxf(fmaxf(fmaxf(sv[0][0], sv[0][1]), fmaxf(sv[0][2], sv[0][3])),
                                   fmaxf(fmaxf(sv[0][4], sv[0][5]), fmaxf(sv[0][6], sv[0][7]))));
        rmax1 = fmaxf(rmax1, fmaxf(fmaxf(fmaxf(sv[1][0], sv[1][1]), fmaxf(sv[1][2], sv[1][3])),
                                   fmaxf(fmaxf(sv[1][4], sv[1][5]), fmaxf(sv[1][6], sv[1][7]))));
      } else if (take > 0) {
        float m0 = NEG_INF, m1 = NEG_INF;
        const uint32_t lo = (uint32_t)(filled - g4), n_in = (uint32_t)take;
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const bool in = ((uint32_t)(16 * (v >> 2) + (v & 3)) - lo) < n_in;
          m0 = fmaxf(m0, in ? sv[0][v] : NEG_INF);
          m1 = fmaxf(m1, in ? sv[1][v] : NEG_INF);
        }
        rmax0 = fmaxf(rmax0, m0);
        rmax1 = fmaxf(rmax1, m1);
      }
      filled += take;
      C.pos += take;
      if (C.pos >= C.len) {
        finish_doc(C, lane);
        C.next_doc(dl);
      }
    }
  }
  __device__ __forceinline__ void drain(Cursor& C, const DocLanes& dl, int lane) {
    while (C.valid) {
      finish_doc(C, lane);
      C.next_doc(dl);
    }
  }
};
using Reducer2x16 = Reducer2x16T<false>;

// Reduction state that also tracks WHERE each query token's maximum sits (training-form forward: the backward pass
// routes gradients through the arg-max token, torch.max semantics = first maximal index).  Dense mode only.
struct ReducerArg {
  float rmax, myscore;
  int ridx, jdoc;
  __device__ __forceinline__ void init() {
    rmax = NEG_INF;
    myscore = 0.0f;
    ridx = 0;
    jdoc = 0;
  }
  // argrow: &argmax[(q * nd + d) * Lq] of the doc that just completed
  __device__ __forceinline__ void finish_doc(int lane, int32_t* argrow, int Lq) {
    const uint32_t xb = __float_as_uint(rmax);
    const auto sv = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
    const auto si = __builtin_amdgcn_permlane32_swap((uint32_t)ridx, (uint32_t)ridx, false, false);
    const float a = __uint_as_float(sv[0]), b = __uint_as_float(sv[1]);  // lower / upper lane half's running max
    const int ia = (int)si[0], ib = (int)si[1];
    const bool take_b = (b > a) || (b == a && ib < ia);
    float v = take_b ? b : a;
    const int idx = take_b ? ib : ia;
    if (lane < Lq) argrow[lane] = idx;
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    const float sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0)) +
                     __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
    myscore = (lane == jdoc) ? sc : myscore;
    ++jdoc;
    rmax = NEG_INF;
    ridx = 0;
  }
  // argbase: &argmax[(q * nd + first doc of this wave) * Lq]; docs of a wave are consecutive
  __device__ __forceinline__ void reduce_tile(const float (&sv)[16], Cursor& C, const DocLanes& dl, int lane,
                                              int32_t* argbase, int Lq) {
    const int hh = lane >> 5;
    int filled = 0;
    while (filled < 32 && C.valid) {
      const int take = uni(min(32 - filled, max(C.len - C.pos, 0)));
      const int nbase = C.pos - filled;  // slot s of this tile is token nbase + s of the current doc
      const uint32_t lo = (uint32_t)(filled - 4 * hh), n_in = (uint32_t)take;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int slot = (v & 3) + 8 * (v >> 2) + 4 * hh;
        const uint32_t rel = (uint32_t)((v & 3) + 8 * (v >> 2)) - lo;
        const bool better = (rel < n_in) && (sv[v] > rmax);  // strict: the first maximal token wins
        rmax = better ? sv[v] : rmax;
        ridx = better ? nbase + slot : ridx;
      }
      filled += take;
      C.pos += take;
      if (C.pos >= C.len) {
        finish_doc(lane, argbase + (int64_t)C.j * Lq, Lq);
        C.next_doc(dl);
      }
    }
  }
};

// Eight consecutive query dims -> the index type's NP 16-bit pieces, packed in pairs (what a lane holds per MFMA k-group):
//   fp16 / fp32-fast  x = hi + 2^-11 lo (two fp16 pieces)
//   bf16              x = b0 + b1 + b2 (round-to-nearest pieces: 24 significant bits in all)
//   fp32 as 3 x bf16  x = t0 + t1 + t2 exactly (truncation pieces)
template <int DT>
__device__ __forceinline__ void q_split_pack(const float (&q)[8], u32x4 (&w)[DT == MAXSIM_F32 ? 1 : StreamTraits<DT>::NP]) {
  constexpr int NP = StreamTraits<DT>::NP;
  constexpr bool F16Q = (DT == MAXSIM_F16 || DT == F32S);
  uint16_t pc[DT == MAXSIM_F32 ? 1 : NP][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = q[j];
    if constexpr (DT == MAXSIM_F32) {
      pc[0][j] = 0;
    } else if constexpr (DT == F32X) {  // exact truncation split: x = t0 + t1 + t2
      const uint32_t u0 = __float_as_uint(x) & 0xffff0000u;
      const float r1 = x - __uint_as_float(u0);
      const uint32_t u1 = __float_as_uint(r1) & 0xffff0000u;
      const float r2 = r1 - __uint_as_float(u1);
      pc[0][j] = (uint16_t)(u0 >> 16);
      pc[1][j] = (uint16_t)(u1 >> 16);
      pc[NP - 1][j] = (uint16_t)(__float_as_uint(r2) >> 16);
    } else if constexpr (F16Q) {
      _Float16 hi = (_Float16)x;
      _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
      __builtin_memcpy(&pc[0][j], &hi, 2);
      __builtin_memcpy(&pc[1][j], &lo, 2);
    } else {
      uint16_t b0 = f32_to_bf16_rn(x);
      float r1 = x - bf16_to_f32(b0);
      uint16_t b1 = f32_to_bf16_rn(r1);
      float r2 = r1 - bf16_to_f32(b1);
      pc[0][j] = b0;
      pc[1][j] = b1;
      pc[NP - 1][j] = f32_to_bf16_rn(r2);
    }
  }
#pragma unroll
  for (int k = 0; k < (DT == MAXSIM_F32 ? 1 : NP); ++k)
#pragma unroll
    for (int y = 0; y < 4; ++y) w[k][y] = (uint32_t)pc[k][2 * y] | ((uint32_t)pc[k][2 * y + 1] << 16);
}

constexpr int QT_2X16 = 48;  // 32 query tokens as two 16-column blocks of v_mfma_f32_16x16x4_f32
// QT = 16: at most 16 query tokens (e.g. the multi-view configs, dense.yaml q_view): fp32 index on
// v_mfma_f32_16x16x4_f32 -- half the matrix-pipe time of the 32-column form, half the query registers.
// Small launches (the reference's online call is ONE query x ~1000 candidates, faiss_indexers.py:234):
//   SPLITK  a doc is streamed by p.split (2 or 4) waves of the workgroup, each a slice of its tokens -- with one wave per
//           doc a 1000-candidate launch leaves half the wave slots of the chip empty and every wave walks 6 tiles serially.
//           Measured (tools/bench_small.py, 1 query x 1000 docs x 180 tokens): fp16 index 31 -> 17 us; the fp32 index
//           is already bandwidth-bound with one wave per doc (16 KiB tiles: 16 MB in flight) and does not gain.
constexpr int SPLIT_MAX_DOCS = 8;  // docs per team of waves in a SPLITK launch (their parked maxima live in LDS)
// LIST (counted candidate rows, maxsim_worklist.h): the grid is fixed and every WAVE walks the device-built list of wave
//   items (query, first slot, docs): item = global wave id, + waves in the grid, ...; the waves of a workgroup share
//   nothing (each holds its own query registers), so they may be on different queries.
// BAL (static-grid rerank of a RAGGED index): the workgroup's docs are dealt to its waves by TOKENS, not by count.  A wave's
//   stream is ~12 docs of 120 +- 40 tokens: with equal doc counts the waves of a workgroup differ by +-10 % in length and the
//   workgroup keeps its LDS until its longest wave is done.  Every wave loads the descriptors of ALL the workgroup's docs (one
//   per lane: dpw <= 64), scans the lengths, and takes the contiguous run of docs whose midpoints fall into its quarter of the
//   workgroup's tokens.  Per-doc arithmetic is untouched: scores are bit-identical; uniform indexes keep the plain cut, and so
//   does the fp32 index (power-limited: no gain measured, tu_stream.hip).
template <int MODE, int DT, int WAVES, int NT, int ABLATE = 0, int QT = 32, bool SPLITK = false, bool LIST = false, bool BAL = false>  // ABLATE (diagnostic): 1 = no MFMA, 2 = no DMA
__global__ void __launch_bounds__(WAVES * 64) k_maxsim_stream(KARGS_DECL) {
  static_assert(!BAL || (MODE == MODE_RERANK && !SPLITK && !LIST), "token-balanced cut: static-grid rerank only");
  static_assert(MODE == MODE_RERANK || DT == MAXSIM_F32, "dense (masked) mode is exact fp32 only");
  static_assert(!LIST || (MODE == MODE_RERANK && !SPLITK), "work-list form: rerank, unsplit");
  static_assert(!SPLITK || (MODE == MODE_RERANK && QT == QT_2X16 && WAVES == 4), "split form: two 16-column blocks only");
  static_assert(QT == 32 || (MODE == MODE_RERANK && ((QT == 16 && DT == MAXSIM_F32) ||
                                                     (QT == QT_2X16 && DT != F32S))),
                "16-column forms: rerank only (the fp16-split fast mode keeps the 32x32x16 form: 1 % faster there)");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  using T = StreamTraits<DT>;
  constexpr int ROWB = T::ROWB, TILE = T::TILE, NDMA = T::NDMA, RPD = T::RPD, LPR = T::LPR, NRD = T::NRD, NP = T::NP;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  // a finished score (passes after the first of a query longer than 32 tokens add to the cell)
  auto put_score = [&](float* cell, float v) __attribute__((always_inline)) { *cell = (p.accum ? *cell : 0.0f) + v; };
  // unsplit: every wave has its own docs.  SPLITK: `split` consecutive waves form a team that shares dpwv docs; wave
  // `part` of the team streams the part-th slice of each doc (slices are whole 32-row tiles, cut as evenly as possible)
  const int split = SPLITK ? p.split : 1;
  const int team = SPLITK ? wave / split : wave, part = SPLITK ? wave - team * split : 0;
  // one wave item: candidates [c_begin, c_begin + ndoc) of query qi
  auto wave_item = [&](const int qi, const int c_begin, const int ndoc, const DocLanes* pre = nullptr) __attribute__((always_inline)) {
#ifdef MAXSIM_STAMP  // timing builds only (tools/probe_timeline.py): 100 MHz stamps of this wave's phases -> p.d_mask
  uint64_t stamp[6];
  stamp[0] = __builtin_amdgcn_s_memrealtime();
#define MAXSIM_STAMP_AT(i) stamp[i] = __builtin_amdgcn_s_memrealtime()
#else
#define MAXSIM_STAMP_AT(i)
#endif
  DocLanes dl = pre ? *pre : load_doc_lanes<MODE>(p, qi, c_begin, ndoc, threadIdx.x & 63);
  if constexpr (SPLITK) {
    const int per = (((dl.len + 31) >> 5) + split - 1) / split * 32;  // rows per slice
    const int start = min(dl.len, part * per);
    dl.row0 += (uint32_t)start;
    dl.len = min(dl.len - start, per);  // may be 0: the slice then parks -inf maxima (neutral)
  }
  char* const wlds = lds + wave * (NT * TILE);
  const int r = lane & 31, hh = lane >> 5;

  // ---- per-lane constants ------------------------------------------------------------------------------------
  const int rsw = r & 15;
  const int rdbase = r * ROWB;
  const char* const tok = (const char*)p.index;

  Cursor F, C;
  F.init(dl, ndoc);
  C = F;
  MAXSIM_STAMP_AT(1);  // descriptors are here

  auto issue_tile = [&](int buf, const TileMap& t) __attribute__((always_inline)) {
    if (ABLATE == 2) return;
    issue_rows<NDMA, RPD, LPR, false, MODE == MODE_RERANK ? CPOL_STREAM : 0>(tok, (uint32_t)ROWB, 0u, wlds + buf * TILE, t, lane);
  };

  // ---- prologue: up to NT tiles in flight (issued BEFORE the query tile is loaded: its latency overlaps the
  //      first fetch) -----------------------------------------------------------------------
  int nissued = 0, nconsumed = 0;
  bool prev_issued = false;
  auto prologue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const TileMap t = fill_tile(F, dl, r);
      if (t.kind != 0) {
        issue_tile(j, t);
        ++nissued;
      }
      prev_issued = t.kind != 0;
    }
  };
  // QFIRST (the split form: small launches of a 16-bit index, two tiles per wave in the ring): the query goes to registers
  // BEFORE the first fetches.  Behind them its loads sit at the end of the in-order vmcnt queue, so the wave cannot start
  // on tile 0 before BOTH prologue tiles have landed -- in a one-query launch every wave asks for its first two tiles at
  // once (2000 waves x 16 KiB), that burst takes ~8 us to deliver, and nothing is in flight when it ends (stamped:
  // tools/probe_timeline.py).  In front of them the query costs one exposed L2 / HBM latency (~1 us) and the stream never stops.
  constexpr bool QFIRST = SPLITK;
  if constexpr (!QFIRST) prologue();
  if (!SPLITK && nissued == 0) {  // (the split form meets at a workgroup barrier below: no early exit)
    // nothing to stream: every slot of this wave is a padding slot (-inf) or an empty doc (0) -- e.g. the tail of a
    // doc-sharded candidate row (maxsim_shard_candidates).  Retire before the 16 KiB query tile is fetched.
    float* const srow0 = p.scores + (int64_t)qi * p.ncand + c_begin;
    if (lane < ndoc) put_score(srow0 + lane, (dl.flags & 3) == 1 ? 0.0f : NEG_INF);
    return;
  }

  // ---- query tile -> registers in MFMA B layout --------------------------------------------------------------
  // fp32: lane (n, hh) holds Q[n][32 s + 8 u + 4 hh + t]   in qv[4 s + u][t]
  // 16b : lane (n, hh) holds Q[n][16 i + 8 hh + j], j=0..7 in qp[piece][i] (packed pairs)
  f32x4 qv[DT == MAXSIM_F32 ? 16 : 1];
  u32x4 qp[DT == MAXSIM_F32 ? 1 : NP][DT == MAXSIM_F32 ? 1 : 8];
  // QSTAGE (small launches: every workgroup of the launch wants the SAME one or few query tiles at the same moment): the
  // workgroup fetches its query's 32 x 128 fp32 image ONCE, in whole cache lines, into the still empty rings, and each
  // wave takes its B-operand layout from LDS.  With every wave gathering its own 16 KiB from global memory a one-query
  // launch sends 2000 waves x 256 line requests at the same 128 lines: the L2 channels that hold them serialise the
  // requests (stamped: 4-7 us from "descriptors there" to "query in registers"; 16x fewer requests this way).
  constexpr bool QSTAGE = QFIRST;
  constexpr int QS_ROW = 132;  // floats per staged row: 512 B + 16 B of padding (bank spread of the 16 token rows)
  const float* const lds_q = (const float*)lds;
  constexpr bool F16Q = (DT == MAXSIM_F16 || DT == F32S);  // query split into fp16 hi + 2^-11 lo
  // eight consecutive query dims -> this index type's NP 16-bit pieces, packed in pairs (what a lane holds per k-group)
  auto split_pack = [&](const float (&q)[8], u32x4 (&w)[DT == MAXSIM_F32 ? 1 : NP]) __attribute__((always_inline)) {
    q_split_pack<DT>(q, w);
  };
  if constexpr (QSTAGE) {
    int qlen0 = p.Lq;
    if (MODE == MODE_RERANK && p.q_len) qlen0 = min(qlen0, p.q_len[qi]);
    const bool qf32s = p.q_dtype == MAXSIM_F32;
    if constexpr (DT == MAXSIM_F32) {  // the fp32 image, row-major (each wave gathers its layout from it)
      for (int c = threadIdx.x; c < 32 * 32; c += WAVES * 64) {
        const int row = c >> 5, chunk = c & 31;
        const int tokq = p.q_tok0 + row;
        const bool lv = q_token_live<MODE>(p, qi, tokq, qlen0);
        const int64_t src = ((int64_t)qi * p.Lq + (lv ? tokq : 0)) * 128 + 4 * chunk;
        f32x4 v;
        if (qf32s) {
          v = *(const f32x4*)((const float*)p.Q + src);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = load_q(p.Q, p.q_dtype, src + t);
        }
        *(f32x4*)((float*)lds + row * QS_ROW + 4 * chunk) = lv ? v : (f32x4)(0.0f);
      }
    } else {
      // 16-bit index: the pieces are split ONCE per workgroup and staged in their final per-lane order -- entry (piece k,
      // k-group i, lane) at [(8 k + i) * 64 + lane] -- so a wave's query registers are NP * 8 ds_read_b128, no VALU
      // (every wave splitting the same 32 x 128 values itself: ~400 VALU per wave at the head of a ~13 us kernel)
      for (int e = threadIdx.x; e < 8 * 64; e += WAVES * 64) {
        const int i = e >> 6, ln = e & 63;
        const int tok16 = p.q_tok0 + 16 * (i >> 2) + (ln & 15);
        const bool lv = q_token_live<MODE>(p, qi, tok16, qlen0);
        const int64_t e0 = ((int64_t)qi * p.Lq + (lv ? tok16 : 0)) * 128 + 8 * (4 * (i & 3) + (ln >> 4));
        float q[8];
        if (qf32s) {
          const f32x4 v0 = *(const f32x4*)((const float*)p.Q + e0), v1 = *(const f32x4*)((const float*)p.Q + e0 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { q[j] = v0[j]; q[4 + j] = v1[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) q[j] = load_q(p.Q, p.q_dtype, e0 + j);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = lv ? q[j] : 0.0f;
        u32x4 w[DT == MAXSIM_F32 ? 1 : NP];
        split_pack(q, w);
#pragma unroll
        for (int k = 0; k < NP; ++k) ((u32x4*)lds)[(8 * k + i) * 64 + ln] = w[k];
      }
    }
    __syncthreads();
  }
  {
    int qlen = p.Lq;
    if (MODE == MODE_RERANK && p.q_len) qlen = min(qlen, p.q_len[qi]);
    const int qtok = p.q_tok0 + r;  // this lane's query token (queries longer than 32 tokens: one launch per 32)
    const bool live = q_token_live<MODE>(p, qi, qtok, qlen);
    const int64_t qoff = ((int64_t)qi * p.Lq + (live ? qtok : 0)) * 128;
    const float* qrow = (const float*)p.Q + qoff;
    const bool qf32 = p.q_dtype == MAXSIM_F32;  // a 16-bit query is widened element by element (start-up only)
    if constexpr (QT != 32 && DT == MAXSIM_F32) {
      // lane (n = lane & 15, kq = lane >> 4) holds Q[16 cb + n][16 j + 4 kq + t] in qv[8 cb + j][t], j = 0..7
      const int n16 = lane & 15, kq = lane >> 4;
#pragma unroll
      for (int cb = 0; cb < (QT == 16 ? 1 : 2); ++cb) {
        const int qt16 = p.q_tok0 + 16 * cb + n16;
        const bool live16 = q_token_live<MODE>(p, qi, qt16, qlen);
        const int64_t qo = ((int64_t)qi * p.Lq + (live16 ? qt16 : 0)) * 128;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f32x4 v;
          if constexpr (QSTAGE) {  // staged rows are already zero where the token is not live
            qv[8 * cb + j] = *(const f32x4*)(lds_q + (16 * cb + n16) * QS_ROW + 16 * j + 4 * kq);
            continue;
          }
          if (qf32) {
            v = *(const f32x4*)((const float*)p.Q + qo + 16 * j + 4 * kq);
          } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = load_q(p.Q, p.q_dtype, qo + 16 * j + 4 * kq + t);
          }
          qv[8 * cb + j] = live16 ? v : (f32x4)(0.0f);
        }
      }
    } else if constexpr (DT == MAXSIM_F32) {
      float qs = 1.0f;
      if (MODE == MODE_DENSE && live && p.mask_dtype != MAXSIM_MASK_NONE)
        qs = load_mask(p.q_mask, p.mask_dtype, (int64_t)qi * p.Lq + qtok);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e0 = 4 * hh + 32 * (i >> 2) + 8 * (i & 3);
        f32x4 v;
        if (qf32) {
          v = *(const f32x4*)(qrow + e0);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = load_q(p.Q, p.q_dtype, qoff + e0 + t);
        }
        if (MODE == MODE_DENSE) v *= qs;  // Q * q_mask[..., None], BaseModel.py:42
        qv[i] = live ? v : (f32x4)(0.0f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // 32-column form: lane (n, hh) takes dims 16 i + 8 hh .. + 7 of token n into qp[.][i];
        // two 16-column blocks (QT_2X16): lane (n16, kq), i = 4 cb + j: dims 8 (4 j + kq) .. + 7 of token 16 cb + n16
        bool lv = live;
        int64_t e0 = qoff + 8 * hh + 16 * i;
        if constexpr (QT == QT_2X16) {
          const int tok16 = p.q_tok0 + 16 * (i >> 2) + (lane & 15);
          lv = q_token_live<MODE>(p, qi, tok16, qlen);
          e0 = ((int64_t)qi * p.Lq + (lv ? tok16 : 0)) * 128 + 8 * (4 * (i & 3) + (lane >> 4));
        }
        if constexpr (QSTAGE && QT == QT_2X16) {  // staged in final order by the workgroup (above)
#pragma unroll
          for (int k = 0; k < NP; ++k) qp[k][i] = ((const u32x4*)lds)[(8 * k + i) * 64 + lane];
          continue;
        }
        float q[8];
        if (qf32) {
          const f32x4 v0 = *(const f32x4*)((const float*)p.Q + e0);
          const f32x4 v1 = *(const f32x4*)((const float*)p.Q + e0 + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { q[j] = v0[j]; q[4 + j] = v1[j]; }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) q[j] = load_q(p.Q, p.q_dtype, e0 + j);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = lv ? q[j] : 0.0f;
        u32x4 w[DT == MAXSIM_F32 ? 1 : NP];
        split_pack(q, w);
#pragma unroll
        for (int k = 0; k < NP; ++k) qp[k][i] = w[k];
      }
    }
  }
  if constexpr (QFIRST) {
    wait_vmcnt<0>();  // the query is in registers: from here on only tile fetches count in vmcnt
    if constexpr (QSTAGE) {
      wait_lgkmcnt0();
      __syncthreads();  // every wave has taken its operands: the rings may be overwritten
    }
    prologue();
  }

  Reducer red;
  Reducer16 red16;
  Reducer2x16T<SPLITK> red2;
  red.init();
  red16.init();
  red2.init();
  // SPLITK: this wave's parked maxima, [doc ordinal][32] floats, behind the waves' rings
  float* const parked = (float*)(lds + WAVES * (NT * TILE));
  if constexpr (SPLITK) red2.part = parked + wave * (SPLIT_MAX_DOCS * 32);
  int buf = 0;

  MAXSIM_STAMP_AT(2);  // first fetches issued, query in registers
#ifdef MAXSIM_STAMP2  // (with MAXSIM_STAMP) per-wave sums over its tiles, 10 ns units: arrival wait | operand reads | next fetch issued | contraction + reduce
  uint64_t ph_wait = 0, ph_read = 0, ph_issue = 0, ph_mm = 0;
  // a time stamp the scheduler may not move anything across (the matrix instructions have no memory dependence to hold them)
  auto ph_now = [&]() __attribute__((always_inline)) {
    uint64_t t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };
#endif
  while (nconsumed < nissued) {
    __builtin_amdgcn_s_setprio(0);
#ifdef MAXSIM_STAMP2
    const uint64_t ph_a = ph_now();
#endif
    // tiles c+1 .. c+NT-1 were issued after this one iff the previous step issued
    if (prev_issued) wait_vmcnt<NDMA * (NT - 1)>(); else wait_vmcnt<0>();
#ifdef MAXSIM_STAMP
    if (nconsumed == 0) stamp[3] = __builtin_amdgcn_s_memrealtime();  // the first tile has arrived
#endif
#ifdef MAXSIM_STAMP2
    const uint64_t ph_b = ph_now();
    ph_wait += ph_b - ph_a;
#endif
    const char* tl = wlds + buf * TILE + rdbase;
    u32x4 a[NRD];
#pragma unroll
    for (int i = 0; i < NRD; ++i) {
      if constexpr (QT == QT_2X16 && DT == F32X) {
        // fp32 storage contracted in 16-bit pieces: a lane's k-step operand is 8 consecutive dims = two 16-byte chunks:
        // operand pair (2 u, 2 u + 1), u = 4 b + j: row 16 b + (lane & 15), chunks 2 (4 j + (lane >> 4)) + {0, 1}
        const int n16 = lane & 15;
        const int c = 2 * (4 * ((i >> 1) & 3) + (lane >> 4)) + (i & 1);
        a[i] = *(const u32x4*)(wlds + buf * TILE + (16 * (i >> 3) + n16) * ROWB + 16 * (c ^ n16));
      } else if constexpr (QT != 32) {
        // operand i = (row block b = i / CPB, k group j = i % CPB): row 16 b + (lane & 15), chunk 4 j + (lane >> 4)
        constexpr int CPB = NRD / 2;
        const int n16 = lane & 15;
        a[i] = *(const u32x4*)(wlds + buf * TILE + (16 * (i / CPB) + n16) * ROWB + 16 * ((4 * (i % CPB) + (lane >> 4)) ^ n16));
      } else {
        // chunk of the row this lane needs for operand i: 16-bit MFMA k-step = 8 consecutive dims per lane half
        const int c = (DT == F32S || DT == F32X) ? (4 * (i >> 1) + 2 * hh + (i & 1)) : (2 * i + hh);
        a[i] = *(const u32x4*)(tl + 16 * (c ^ rsw));
      }
    }
    wait_lgkmcnt0();  // operands are in registers: the buffer may be overwritten
#ifdef MAXSIM_STAMP2
    const uint64_t ph_c = ph_now();
    ph_read += ph_c - ph_b;
#endif
    {
      const TileMap t = fill_tile(F, dl, r);
      if (t.kind != 0) {
        issue_tile(buf, t);
        ++nissued;
      }
      prev_issued = t.kind != 0;
    }
    buf = (buf + 1 == NT) ? 0 : buf + 1;
#ifdef MAXSIM_STAMP2
    const uint64_t ph_d = ph_now();
    ph_issue += ph_d - ph_c;
#endif
    // The contraction + reduce phase runs at raised priority: when both waves of a SIMD hold a tile, the matrix pipe
    // finishes one of them first (instead of interleaving both), so that wave's next fetch wait starts earlier.
    __builtin_amdgcn_s_setprio(3);

    float mv = 1.0f;
    if constexpr (MODE == MODE_DENSE) {  // D * d_mask[..., None], BaseModel.py:41
      if (p.mask_dtype != MAXSIM_MASK_NONE) {
        Cursor Cp = C;
        const TileMap ct = fill_tile(Cp, dl, r);
        mv = load_mask(p.d_mask, p.mask_dtype, (int64_t)ct.myrow);
      }
    }

    if constexpr (QT == QT_2X16 && DT != MAXSIM_F32) {
      // v_mfma_f32_16x16x32_{f16,bf16}: the chip holds a higher clock on this shape than on 32x32x16 at the same
      // flop rate (MI355X_MICROARCH.md, DVFS give-back item 7), and these kernels run on the board power cap
      f32x4 acc0[2][2], acc1[2][2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc0[b][cb] = acc1[b][cb] = (f32x4)(0.0f);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if constexpr (DT == F32X) {
            const u32x4 x0 = a[8 * b + 2 * j], x1 = a[8 * b + 2 * j + 1];
            if (ABLATE == 1) {
              asm volatile("" ::"v"(x0), "v"(x1));
              continue;
            }
            {
              u32x4 t0, t1, t2;  // packed bf16 pairs: element w = dims 2w, 2w+1 of this lane's 8
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                const uint32_t xa = w < 2 ? x0[2 * w] : x1[2 * w - 4], xb = w < 2 ? x0[2 * w + 1] : x1[2 * w - 3];
                const float ra = __uint_as_float(xa) - __uint_as_float(xa & 0xffff0000u);
                const float rb = __uint_as_float(xb) - __uint_as_float(xb & 0xffff0000u);
                const uint32_t ua = __float_as_uint(ra), ub = __float_as_uint(rb);
                const float sa = ra - __uint_as_float(ua & 0xffff0000u);
                const float sb = rb - __uint_as_float(ub & 0xffff0000u);
                t0[w] = __builtin_amdgcn_perm(xb, xa, 0x07060302u);  // (hi16(xb) << 16) | hi16(xa)
                t1[w] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
                t2[w] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
              }
              const bf16x8 d0 = __builtin_bit_cast(bf16x8, t0), d1 = __builtin_bit_cast(bf16x8, t1), d2 = __builtin_bit_cast(bf16x8, t2);
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) {  // one accumulator per block: small piece products first within a k-step
                const bf16x8 q0 = __builtin_bit_cast(bf16x8, qp[0][4 * cb + j]), q1 = __builtin_bit_cast(bf16x8, qp[1][4 * cb + j]),
                             q2 = __builtin_bit_cast(bf16x8, qp[NP - 1][4 * cb + j]);
                f32x4 c = acc0[b][cb];
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d2, q0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1, q1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, q2, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1, q0, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, q1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0, q0, c, 0, 0, 0);
                acc0[b][cb] = c;
              }
            }
          } else {
            if (ABLATE == 1) {
              asm volatile("" ::"v"(a[4 * b + j]));
              continue;
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              if constexpr (DT == MAXSIM_F16) {
                const f16x8 av = __builtin_bit_cast(f16x8, a[4 * b + j]);
                acc0[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, qp[0][4 * cb + j]), acc0[b][cb], 0, 0, 0);
                acc1[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, qp[1][4 * cb + j]), acc1[b][cb], 0, 0, 0);
              } else {
                const bf16x8 av = __builtin_bit_cast(bf16x8, a[4 * b + j]);
                acc0[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, qp[0][4 * cb + j]), acc0[b][cb], 0, 0, 0);
                acc1[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, qp[1][4 * cb + j]), acc1[b][cb], 0, 0, 0);
                acc1[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, qp[NP - 1][4 * cb + j]), acc1[b][cb], 0, 0, 0);
              }
            }
          }
        }
      }
      float sv2[2][8];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const float s0 = acc0[v >> 2][cb][v & 3], s1 = acc1[v >> 2][cb][v & 3];
          sv2[cb][v] = (DT == MAXSIM_F16) ? fmaf(s1, 1.0f / 2048.0f, s0) : (s0 + s1);
        }
      red2.reduce_tile(sv2, C, dl, lane);
      ++nconsumed;
      continue;
    }
    if constexpr (QT == QT_2X16 && DT == MAXSIM_F32) {
      f32x4 acc[2][2] = {{(f32x4)(0.0f), (f32x4)(0.0f)}, {(f32x4)(0.0f), (f32x4)(0.0f)}};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (ABLATE == 1) {
          asm volatile("" ::"v"(a[j]), "v"(a[8 + j]));
          continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              acc[b][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, a[8 * b + j])[t], qv[8 * cb + j][t], acc[b][cb], 0, 0, 0);
      }
      float sv2[2][8];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int v = 0; v < 8; ++v) sv2[cb][v] = acc[v >> 2][cb][v & 3];
      red2.reduce_tile(sv2, C, dl, lane);
      ++nconsumed;
      continue;
    }
    if constexpr (QT == 16) {
      f32x4 acc[2] = {(f32x4)(0.0f), (f32x4)(0.0f)};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (ABLATE == 1) {
          asm volatile("" ::"v"(a[j]), "v"(a[8 + j]));
          continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, a[8 * b + j])[t], qv[j][t], acc[b], 0, 0, 0);
      }
      float sv8[8];
#pragma unroll
      for (int v = 0; v < 8; ++v) sv8[v] = acc[v >> 2][v & 3];
      red16.reduce_tile(sv8, C, dl, lane);
      ++nconsumed;
      continue;
    }
    float sv[16];
    if constexpr (DT == MAXSIM_F32) {
      f32x16 acc = (f32x16)(0.0f);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        f32x4 av = __builtin_bit_cast(f32x4, a[i]);
        if (MODE == MODE_DENSE) av *= mv;
        if (ABLATE == 1) {
          asm volatile("" ::"v"(av));
#ifdef MAXSIM_SLEEP_ABLATE
          if (i == 0) __builtin_amdgcn_s_sleep(MAXSIM_SLEEP_ABLATE);
#endif
          continue;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], qv[i][t], acc, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) sv[v] = acc[v];
    } else if constexpr (DT == F32X) {
      f32x16 acc0 = (f32x16)(0.0f);  // one accumulator (register budget): small piece products first within a k-step
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const u32x4 x0 = a[2 * i], x1 = a[2 * i + 1];
        if (ABLATE == 1) {
          asm volatile("" ::"v"(x0), "v"(x1));
          continue;
        }
        u32x4 t0, t1, t2;  // packed bf16 pairs: element j = dims 2j, 2j+1 of this lane's 8
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          const uint32_t xa = w < 2 ? x0[2 * w] : x1[2 * w - 4], xb = w < 2 ? x0[2 * w + 1] : x1[2 * w - 3];
          const float ra = __uint_as_float(xa) - __uint_as_float(xa & 0xffff0000u);
          const float rb = __uint_as_float(xb) - __uint_as_float(xb & 0xffff0000u);
          const uint32_t ua = __float_as_uint(ra), ub = __float_as_uint(rb);
          const float sa = ra - __uint_as_float(ua & 0xffff0000u);
          const float sb = rb - __uint_as_float(ub & 0xffff0000u);
          t0[w] = __builtin_amdgcn_perm(xb, xa, 0x07060302u);  // (hi16(xb) << 16) | hi16(xa)
          t1[w] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
          t2[w] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
        }
        const bf16x8 d0 = __builtin_bit_cast(bf16x8, t0), d1 = __builtin_bit_cast(bf16x8, t1), d2 = __builtin_bit_cast(bf16x8, t2);
        const bf16x8 q0 = __builtin_bit_cast(bf16x8, qp[0][i]), q1 = __builtin_bit_cast(bf16x8, qp[1][i]),
                     q2 = __builtin_bit_cast(bf16x8, qp[NP - 1][i]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d2, q0, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, q1, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, q2, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, q0, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, q1, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, q0, acc0, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) sv[v] = acc0[v];
    } else if constexpr (DT == F32S) {
      f32x16 acc0 = (f32x16)(0.0f), acc1 = (f32x16)(0.0f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f32x4 x0 = __builtin_bit_cast(f32x4, a[2 * i]), x1 = __builtin_bit_cast(f32x4, a[2 * i + 1]);
        if (ABLATE == 1) {
          asm volatile("" ::"v"(x0), "v"(x1));
          continue;
        }
        f16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = j < 4 ? x0[j & 3] : x1[j & 3];
          const _Float16 h = (_Float16)x;
          hi[j] = h;
          lo[j] = (_Float16)((x - (float)h) * 2048.0f);
        }
        const f16x8 qh = __builtin_bit_cast(f16x8, qp[0][i]), ql = __builtin_bit_cast(f16x8, qp[1][i]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, qh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, ql, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, qh, acc1, 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 16; ++v) sv[v] = fmaf(acc1[v], 1.0f / 2048.0f, acc0[v]);
    } else {
      f32x16 acc0 = (f32x16)(0.0f), acc1 = (f32x16)(0.0f);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (ABLATE == 1) {
          asm volatile("" ::"v"(a[i]));
          continue;
        }
        if constexpr (DT == MAXSIM_F16) {
          const f16x8 av = __builtin_bit_cast(f16x8, a[i]);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, qp[0][i]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, qp[1][i]), acc1, 0, 0, 0);
        } else {
          const bf16x8 av = __builtin_bit_cast(bf16x8, a[i]);
          acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, qp[0][i]), acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, qp[1][i]), acc1, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, qp[NP - 1][i]), acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int v = 0; v < 16; ++v)
        sv[v] = (DT == MAXSIM_F16) ? fmaf(acc1[v], 1.0f / 2048.0f, acc0[v]) : (acc0[v] + acc1[v]);
    }

    red.reduce_tile(sv, C, dl, lane);
    ++nconsumed;
#ifdef MAXSIM_STAMP2
    ph_mm += ph_now() - ph_d;
#endif
  }
  MAXSIM_STAMP_AT(4);  // last tile contracted and reduced
  float* const srow = p.scores + (int64_t)qi * p.ncand + c_begin;
#ifdef MAXSIM_STAMP
  if (p.d_mask && lane == 0) {
    uint64_t* const sb = (uint64_t*)p.d_mask + ((int64_t)blockIdx.x * WAVES + wave) * 8;
    sb[0] = stamp[0]; sb[1] = stamp[1]; sb[2] = stamp[2]; sb[3] = stamp[3]; sb[4] = stamp[4];
    sb[5] = (uint64_t)nissued;
    sb[6] = __builtin_amdgcn_s_memrealtime();
#ifdef MAXSIM_STAMP2
    auto c16 = [](uint64_t v) { return v > 65535 ? (uint64_t)65535 : v; };
    sb[7] = c16(ph_wait) | (c16(ph_read) << 16) | (c16(ph_issue) << 32) | (c16(ph_mm) << 48);
#endif
  }
#endif
  if constexpr (SPLITK) {
    red2.drain(C, dl, lane);
    __syncthreads();  // every slice of the workgroup's docs is parked
    if (part == 0) {
      // combine the team's slices doc by doc: max over the slices, then exactly the unsplit wave's floor + sum tree
      const int tok = (lane & 15) + ((lane >> 5) << 4);
      float mine = 0.0f;
      for (int j = 0; j < ndoc; ++j) {
        const int fl = __builtin_amdgcn_readlane(dl.flags, j);
        float sc;
        if ((fl & 3) == 0) {
          float v = NEG_INF;
          for (int sp = 0; sp < split; ++sp) v = fmaxf(v, parked[(wave + sp) * (SPLIT_MAX_DOCS * 32) + j * 32 + tok]);
          sc = Reducer2x16T<true>::floor_and_sum(v, fl >> 2);
        } else {
          sc = (fl & 3) == 1 ? 0.0f : NEG_INF;
        }
        mine = (lane == j) ? sc : mine;
      }
      if (lane < ndoc) put_score(srow + lane, mine);
    }
  } else if constexpr (QT == QT_2X16) {
    red2.drain(C, dl, lane);
    if (lane < red2.jdoc) put_score(srow + lane, red2.myscore);
  } else if constexpr (QT == 16) {
    red16.drain(C, dl, lane);
    if (lane < red16.jdoc) put_score(srow + lane, red16.myscore);
  } else {
    red.drain(C, dl, lane);
    if (lane < red.jdoc) put_score(srow + lane, red.myscore);
  }
  };  // wave_item
  if constexpr (LIST) {
    // The list is cut into workgroup items of WAVES consecutive wave items; workgroup slot s = blockIdx.x, + gridDim.x, ...
    // (the grid is a multiple of 8 workgroups, so a workgroup's slots all have its own s % 8).  Slot -> item is
    // XCD-aware: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs, so the slots with s % 8 = x
    // walk the x-th EIGHTH of the list in order -- the ~16 wave items of one query (4 workgroup items) are taken by
    // one XCD at about the same time and its 16 KiB query tile is fetched into that XCD's L2 once instead of by every
    // workgroup from the fabric (2048 queries x 16 wave items x 16 KiB = 0.5 GB per launch otherwise: +4 % time).
    const int32_t* const wl = (const int32_t*)p.worklist;
    const int wl_total = uni(wl[0]);
    const int2* const wl_items = (const int2*)(wl + worklist_items_word(p.nq));
    const int J = (wl_total + WAVES - 1) / WAVES, Jx = (J + 7) >> 3;
    for (int s = (int)blockIdx.x; s < 8 * Jx; s += (int)gridDim.x) {
      const int item = ((s & 7) * Jx + (s >> 3)) * WAVES + wave;
      if ((s >> 3) >= Jx || item >= wl_total) continue;
      const int2 e = wl_items[item];
      __builtin_amdgcn_s_setprio(0);
      wave_item(uni(e.x), uni(e.y) & ((1 << WL_SLOT_BITS) - 1), uni(e.y) >> WL_SLOT_BITS);
    }
  } else {
    int qi, chunk;
    wg_to_work((int)blockIdx.x, p.nq, p.nchunk, qi, chunk);
    if constexpr (BAL) {
      const int c0 = chunk * p.dpw, nwg = max(0, min(p.dpw, p.ncand - c0));     // the workgroup's docs: one per lane (dpw <= 64)
      const DocLanes all = load_doc_lanes<MODE>(p, qi, c0, nwg, lane);
      int incl = all.len;                                                         // inclusive scan of the lengths over the lanes
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        incl += lane >= d ? t : 0;
      }
      const int64_t T = __builtin_amdgcn_readlane(incl, 63);
      const int64_t mid2 = 2 * (int64_t)(incl - all.len) + all.len;               // twice the doc's midpoint in the workgroup's stream
      // docs in front of wave w's share: those whose midpoint lies below w T / WAVES (midpoints are non-decreasing over the
      // lanes, so the ballot is a prefix mask); padding slots and empty docs (length 0) go with their neighbours
      auto cut = [&](int w) {
        if (w <= 0) return 0;
        if (w >= WAVES) return nwg;
        return (int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < nwg && mid2 * WAVES < 2 * (int64_t)w * T));
      };
      const int s0 = cut(wave), s1 = cut(wave + 1);
      DocLanes mine;
      mine.row0 = (uint32_t)__shfl((int)all.row0, (lane + s0) & 63);
      mine.len = __shfl(all.len, (lane + s0) & 63);
      mine.flags = __shfl(all.flags, (lane + s0) & 63);
      if (lane >= s1 - s0) { mine.row0 = 0; mine.len = 0; mine.flags = 2; }
      wave_item(qi, c0 + s0, s1 - s0, &mine);
    } else {
    const int dpwv = SPLITK ? p.dpw / (WAVES / split) : p.dpw / WAVES;  // docs per wave / per team (<= 64; SPLITK: <= SPLIT_MAX_DOCS)
    const int c_begin = chunk * p.dpw + team * dpwv;
    wave_item(qi, c_begin, max(0, min(dpwv, p.ncand - c_begin)));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 rerank on HALF tiles: the stream is cut into 16-row tiles (8 KiB, one row block of v_mfma_f32_16x16x4_f32), the
// wave's ring holds two of them.  While one half tile is being contracted the other is in flight, and the gap between
// "tile arrived" and "next fetch issued" (operand reads, row map, address math) never leaves the wave with nothing in
// flight -- with one 16 KiB tile per wave that gap costs ~10 % of the fetch rate.
template <int NCB>
struct ReducerH {
  float rmax[NCB], myscore;
  int jdoc;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) rmax[cb] = NEG_INF;
    myscore = 0.0f;
    jdoc = 0;
  }
  __device__ __forceinline__ void finish_doc(const Cursor& C, int lane) {
    float sc;
    if (C.kind == 0) {
      // row quarters g, g + 2 (lane halves); with two column blocks the swap also parks block 0 in the lower half
      // and block 1 in the upper half
      const uint32_t x0 = __float_as_uint(rmax[0]), x1 = __float_as_uint(rmax[NCB - 1]);
      const auto sw = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
      float v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      const uint32_t xb = __float_as_uint(v);
      const auto s16 = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);  // quarters g, g + 1
      v = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
      if (C.floor0) v = fmaxf(v, 0.0f);
      v += dpp_f32<0xB1>(v);
      v += dpp_f32<0x4E>(v);
      v += dpp_f32<0x141>(v);
      v += dpp_f32<0x140>(v);  // 16-lane row sums: row 0 = query tokens 0..15, row 2 = tokens 16..31
      sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0));
      if (NCB == 2) sc += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 32));
    } else {
      sc = C.kind == 1 ? 0.0f : NEG_INF;
    }
    myscore = (lane == jdoc) ? sc : myscore;
    ++jdoc;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) rmax[cb] = NEG_INF;
  }
  // sv[cb][v] = similarity of query token 16 cb + (lane & 15) with tile row 4 (lane >> 4) + v
  __device__ __forceinline__ void reduce_tile(const float (&sv)[NCB][4], Cursor& C, const DocLanes& dl, int lane) {
    const int g4 = 4 * (lane >> 4);
    int filled = 0;
    while (filled < 16 && C.valid) {
      const int take = uni(min(16 - filled, max(C.len - C.pos, 0)));
      if (take == 16) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          rmax[cb] = fmaxf(rmax[cb], fmaxf(fmaxf(sv[cb][0], sv[cb][1]), fmaxf(sv[cb][2], sv[cb][3])));
      } else if (take > 0) {
        const uint32_t lo = (uint32_t)(filled - g4), n_in = (uint32_t)take;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const bool in = ((uint32_t)v - lo) < n_in;
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) rmax[cb] = fmaxf(rmax[cb], in ? sv[cb][v] : NEG_INF);
        }
      }
      filled += take;
      C.pos += take;
      if (C.pos >= C.len) {
        finish_doc(C, lane);
        C.next_doc(dl);
      }
    }
  }
  __device__ __forceinline__ void drain(Cursor& C, const DocLanes& dl, int lane) {
    while (C.valid) {
      finish_doc(C, lane);
      C.next_doc(dl);
    }
  }
};

template <int WAVES, int NCB, int NT = 2, int ABLATE = 0>  // NCB 16-column query blocks: 1 (Lq <= 16) or 2 (Lq <= 32)
__global__ void __launch_bounds__(WAVES * 64) k_maxsim_stream_f32h(KARGS_DECL) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  constexpr int ROWB = 512, HT = 16 * ROWB, NDMA = 8;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  int qi, chunk;
  wg_to_work((int)blockIdx.x, p.nq, p.nchunk, qi, chunk);
  const int dpwv = p.dpw / WAVES;
  const int c_begin = chunk * p.dpw + wave * dpwv;
  const int ndoc = max(0, min(dpwv, p.ncand - c_begin));
  const DocLanes dl = load_doc_lanes<MODE_RERANK>(p, qi, c_begin, ndoc, lane);
  char* const wlds = lds + wave * (NT * HT);
  const int n16 = lane & 15, kq = lane >> 4;
  const char* const tok = (const char*)p.index;

  Cursor F, C;
  F.init(dl, ndoc);
  C = F;
  int nissued = 0, nconsumed = 0;
  bool prev_issued = false;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const TileMap t = fill_tile<16>(F, dl, n16);
    if (t.kind != 0) {
      if (ABLATE != 2) issue_rows<NDMA, 2, 32, false, CPOL_STREAM>(tok, (uint32_t)ROWB, 0u, wlds + j * HT, t, lane);
      ++nissued;
    }
    prev_issued = t.kind != 0;
  }
  if (nissued == 0) {  // all padding slots / empty docs: retire before the query tile is fetched
    float* const srow0 = p.scores + (int64_t)qi * p.ncand + c_begin;
    if (lane < ndoc) srow0[lane] = (p.accum ? srow0[lane] : 0.0f) + ((dl.flags & 3) == 1 ? 0.0f : NEG_INF);
    return;
  }

  // lane (n, kq) holds Q[16 cb + n][16 j + 4 kq + t] in qv[8 cb + j][t]
  f32x4 qv[8 * NCB];
  {
    int qlen = p.Lq;
    if (p.q_len) qlen = min(qlen, p.q_len[qi]);
    const bool qf32 = p.q_dtype == MAXSIM_F32;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int qt = p.q_tok0 + 16 * cb + n16;
      const bool live = q_token_live<MODE_RERANK>(p, qi, qt, qlen);
      const int64_t qo = ((int64_t)qi * p.Lq + (live ? qt : 0)) * 128;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 v;
        if (qf32) {
          v = *(const f32x4*)((const float*)p.Q + qo + 16 * j + 4 * kq);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = load_q(p.Q, p.q_dtype, qo + 16 * j + 4 * kq + t);
        }
        qv[8 * cb + j] = live ? v : (f32x4)(0.0f);
      }
    }
  }

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
        if (ABLATE != 2) issue_rows<NDMA, 2, 32, false, CPOL_STREAM>(tok, (uint32_t)ROWB, 0u, wlds + buf * HT, t, lane);
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
    for (int j = 0; j < 8; ++j) {
      if (ABLATE == 1) {
        asm volatile("" ::"v"(a[j]));
        continue;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, a[j])[t], qv[8 * cb + j][t], acc[cb], 0, 0, 0);
    }
    float sv[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int v = 0; v < 4; ++v) sv[cb][v] = acc[cb][v];
    red.reduce_tile(sv, C, dl, lane);
    ++nconsumed;
  }
  red.drain(C, dl, lane);
  float* const srow = p.scores + (int64_t)qi * p.ncand + c_begin;
  if (lane < red.jdoc) srow[lane] = (p.accum ? srow[lane] : 0.0f) + red.myscore;
}

// ---------------------------------------------------------------------------------------------------------------------
// Uniform short docs: EVERY doc of the index has exactly L tokens, L = 4, 8 or 16 (maxsim_index_view.uniform_len).  That is
// the reference's multi-view configuration by construction -- `enable_multiview` keeps d_view viewer tokens per doc
// (proj_conf/dense.yaml:31-32: d_view = 8) -- and BASELINE configs[3].
// A launch over such docs is not bound by bytes alone: with the general half-tile kernel the wave spends ~230 scalar / vector
// instructions per 8 KiB tile on walking documents of unknown length (cursor, row map, two-segment addresses, masked
// maxima, one finish per doc) next to its 32 MFMAs, four waves per SIMD: with the fetch switched off the launch still takes
// 0.111 ms of its 0.183 (MAXSIM_VARIANT=2 in a diagnostic build).  With the length a compile-time constant all of that is
// straight-line code: a tile is DPT = 16 / L whole docs, its rows come from DPT readlanes, the source address of DMA
// instruction i is a scalar base plus ONE xor on a per-lane constant, and the DPT docs of a tile finish together (quarter /
// half exchanges instead of a per-doc reduction).  Same k-order, same sum tree: scores are bit-identical to the general
// kernels'.  Padding slots (pid out of range) fetch row 0 and are overwritten with -inf at the end; no 0-floor exists in a
// uniform index (its only length bucket is L itself, colbert_ranker.py:36-40,90).
// (one 16-column query block needs <= 96 registers: five waves per SIMD -- with 4-wave workgroups of 32 KiB that is five
//  workgroups = 20 streams per CU instead of 16; the second launch bound asks the compiler to stay there)
// LIST: counted rows (a doc shard's share of every list): the waves walk the device-built list of wave items, as in k_maxsim_stream.
template <int WAVES, int NCB, int L, int ABLATE = 0, bool LIST = false>
__global__ void __launch_bounds__(WAVES * 64, NCB == 1 ? (LIST ? 4 : 5) : 3) k_maxsim_stream_uni(KARGS_DECL) {
  static_assert(L == 4 || L == 8 || L == 16, "uniform doc length: 4, 8 or 16 tokens");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  constexpr int ROWB = 512, HT = 16 * ROWB, NDMA = 8, DPT = 16 / L;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  auto wave_item = [&](const int qi, const int c_begin, const int ndoc) __attribute__((always_inline)) {
  if (ndoc == 0) return;
  float* const srow = p.scores + (int64_t)qi * p.ncand + c_begin;
  // descriptor lanes: lane j = first token row of the wave's j-th doc
  uint32_t row0 = 0;
  bool bad = true;
  if (lane < ndoc) {
    // (no descriptor lookup: in a uniform index doc pid starts at token row pid * L -- tok_offsets is the prefix sum of the
    //  doclens -- which saves the random 128-byte table line per 4 KiB doc, 3.8 % of C4's traffic, and one dependent load
    //  of the start-up chain)
    const int64_t pid = p.cand[(int64_t)qi * p.ncand + c_begin + lane];
    const bool ok = pid >= 0 && pid < p.n_docs && (pid + 1) * L <= p.n_tokens;
    row0 = ok ? (uint32_t)(pid * L) : 0u;
    bad = !ok;
  }
  char* const wlds = lds + wave * HT;
  const int n16 = lane & 15, kq = lane >> 4;
  const char* const tok = (const char*)p.index;
  const int ntile = (ndoc + DPT - 1) / DPT;
  // DMA instruction i moves tile rows 2 i (lanes 0..31) and 2 i + 1 (lanes 32..63); chunk position c of row s receives
  // source chunk c ^ (s & 15): per lane  off_i = ds0 * 512 + 16 * ((dch ^ ds0) ^ (2 i & 15)) = lane_off ^ (32 i)
  const uint32_t lane_off = (uint32_t)((lane >> 5) * ROWB + 16 * ((lane & 31) ^ (lane >> 5)));
  auto issue = [&](int t) __attribute__((always_inline)) {
    if (ABLATE == 2) return;
    uint32_t base[DPT];
#pragma unroll
    for (int d = 0; d < DPT; ++d) base[d] = (uint32_t)__builtin_amdgcn_readlane((int)row0, min(t * DPT + d, ndoc - 1));
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const char* const g = tok + (uint64_t)(base[(2 * i) / L] + (uint32_t)((2 * i) % L)) * ROWB;
      __builtin_amdgcn_global_load_lds(GPTR(g + (lane_off ^ (uint32_t)(32 * i))), LPTR(wlds + i * 1024), 16, 0, CPOL_STREAM);
    }
  };
  issue(0);
  // lane (n, kq) holds Q[16 cb + n][16 j + 4 kq + t] in qv[8 cb + j][t]
  f32x4 qv[8 * NCB];
  {
    int qlen = p.Lq;
    if (p.q_len) qlen = min(qlen, p.q_len[qi]);
    const bool qf32 = p.q_dtype == MAXSIM_F32;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int qt = p.q_tok0 + 16 * cb + n16;
      const bool live = q_token_live<MODE_RERANK>(p, qi, qt, qlen);
      const int64_t qo = ((int64_t)qi * p.Lq + (live ? qt : 0)) * 128;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 v;
        if (qf32) {
          v = *(const f32x4*)((const float*)p.Q + qo + 16 * j + 4 * kq);
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = load_q(p.Q, p.q_dtype, qo + 16 * j + 4 * kq + t);
        }
        qv[8 * cb + j] = live ? v : (f32x4)(0.0f);
      }
    }
  }
  float myscore = 0.0f;
  for (int t = 0; t < ntile; ++t) {
    __builtin_amdgcn_s_setprio(0);
    wait_vmcnt<0>();
    u32x4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = *(const u32x4*)(wlds + n16 * ROWB + 16 * ((4 * j + kq) ^ n16));
    wait_lgkmcnt0();
    if (t + 1 < ntile) issue(t + 1);
    __builtin_amdgcn_s_setprio(3);
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4)(0.0f);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (ABLATE == 1) {
        asm volatile("" ::"v"(a[j]));
        continue;
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(f32x4, a[j])[tt], qv[8 * cb + j][tt], acc[cb], 0, 0, 0);
    }
    // acc[cb][v] = similarity of query token 16 cb + n16 with tile row 4 kq + v; doc d of the tile = rows [d L, (d + 1) L)
    float tree[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      float m = fmaxf(fmaxf(acc[cb][0], acc[cb][1]), fmaxf(acc[cb][2], acc[cb][3]));  // the quarter's 4 rows
      if constexpr (L >= 8) {  // quarters g, g + 1
        const uint32_t xb = __float_as_uint(m);
        const auto s16 = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
        m = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
      }
      if constexpr (L == 16) {  // quarters g, g + 2
        const uint32_t xb = __float_as_uint(m);
        const auto s32 = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
        m = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
      }
      // sum over the block's 16 query-token lanes: the general kernels' pairwise tree
      m += dpp_f32<0xB1>(m);
      m += dpp_f32<0x4E>(m);
      m += dpp_f32<0x141>(m);
      m += dpp_f32<0x140>(m);
      tree[cb] = m;
    }
#pragma unroll
    for (int d = 0; d < DPT; ++d) {  // doc d's sums sit in the 16-lane row that holds its rows: lane d * (64 / DPT)
      float sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tree[0]), d * (64 / DPT)));
      if constexpr (NCB == 2) sc += __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tree[1]), d * (64 / DPT)));
      myscore = (lane == t * DPT + d) ? sc : myscore;
    }
  }
  if (lane < ndoc) srow[lane] = (p.accum ? srow[lane] : 0.0f) + (bad ? NEG_INF : myscore);
  };  // wave_item
  if constexpr (LIST) {  // (slot -> item as in k_maxsim_stream's list form: XCD x walks the x-th eighth of the list)
    const int32_t* const wl = (const int32_t*)p.worklist;
    const int wl_total = uni(wl[0]);
    const int2* const wl_items = (const int2*)(wl + worklist_items_word(p.nq));
    const int J = (wl_total + WAVES - 1) / WAVES, Jx = (J + 7) >> 3;
    for (int s = (int)blockIdx.x; s < 8 * Jx; s += (int)gridDim.x) {
      const int item = ((s & 7) * Jx + (s >> 3)) * WAVES + wave;
      if ((s >> 3) >= Jx || item >= wl_total) continue;
      const int2 e = wl_items[item];
      wave_item(uni(e.x), uni(e.y) & ((1 << WL_SLOT_BITS) - 1), uni(e.y) >> WL_SLOT_BITS);
    }
  } else {
    int qi, chunk;
    wg_to_work((int)blockIdx.x, p.nq, p.nchunk, qi, chunk);
    const int dpwv = p.dpw / WAVES;
    const int c_begin = chunk * p.dpw + wave * dpwv;
    wave_item(qi, c_begin, max(0, min(dpwv, p.ncand - c_begin)));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Uniform short docs on a 16-BIT index: the reference stores fp16 (colbert_ranker.py:62, encoder.py:175), so its multi-view
// deployments (proj_conf/dense.yaml:29-32: every doc keeps d_view viewer tokens) are exactly this -- every doc L = 4, 8 or
// 16 rows of 256 bytes.  Through the general kernel such a launch is instruction-bound (a packed 32-row tile holds 2-8
// docs: cursor walk, per-slot row gathers, masked maxima and one finish per doc next to 16 short MFMAs: 0.51 of the HBM
// peak at 256 x 1000 eight-token docs).  With L compiled in, as in k_maxsim_stream_uni: a tile is DPT = 32 / L whole docs,
// the source address of DMA instruction i is a scalar base + ONE xor on a per-lane constant (its 4 rows lie in one doc
// since 4 | L), and the DPB = 16 / L docs of a 16-row block finish together.  Contraction, piece split, k-order, 2^-11
// scaling and sum tree are the general 16-bit kernel's (v_mfma_f32_16x16x32, Q = Qhi + 2^-11 Qlo / three bf16 pieces):
// scores are bit-identical to k_maxsim_stream<RERANK, DT, .., QT_2X16>.  NT tiles of 8 KiB per wave in the ring.
template <int DT, int WAVES, int NCB, int L, int NT, bool LIST = false>
__global__ void __launch_bounds__(WAVES * 64) k_maxsim_stream_uni16(KARGS_DECL) {
  static_assert(DT == MAXSIM_F16 || DT == MAXSIM_BF16, "16-bit index types");
  static_assert(L == 4 || L == 8 || L == 16, "uniform doc length: 4, 8 or 16 tokens");
  static_assert(NT == 1 || NT == 2, "one or two tiles per wave in flight");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  constexpr int ROWB = 256, TILE = 32 * ROWB, NDMA = 8, DPT = 32 / L, DPB = 16 / L, NP = StreamTraits<DT>::NP;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  auto wave_item = [&](const int qi, const int c_begin, const int ndoc) __attribute__((always_inline)) {
  if (ndoc == 0) return;
  float* const srow = p.scores + (int64_t)qi * p.ncand + c_begin;
  const int n16 = lane & 15, kq = lane >> 4;
  // ---- query -> registers first (its loads and the pid load below are in flight together; the piece split runs under the
  //      first tile's fetch): lane (n16, kq) holds dims 8 (4 j + kq) .. + 7 of token 16 cb + n16 in qp[piece][4 cb + j]
  u32x4 qp[NP][4 * NCB];
  {
    int qlen = p.Lq;
    if (p.q_len) qlen = min(qlen, p.q_len[qi]);
    const bool qf32 = p.q_dtype == MAXSIM_F32;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int qt = p.q_tok0 + 16 * cb + n16;
      const bool live = q_token_live<MODE_RERANK>(p, qi, qt, qlen);
      const int64_t qo = ((int64_t)qi * p.Lq + (live ? qt : 0)) * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t e0 = qo + 8 * (4 * j + kq);
        float q[8];
        if (qf32) {
          const f32x4 v0 = *(const f32x4*)((const float*)p.Q + e0), v1 = *(const f32x4*)((const float*)p.Q + e0 + 4);
#pragma unroll
          for (int x = 0; x < 4; ++x) { q[x] = v0[x]; q[4 + x] = v1[x]; }
        } else {
#pragma unroll
          for (int x = 0; x < 8; ++x) q[x] = load_q(p.Q, p.q_dtype, e0 + x);
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) q[x] = live ? q[x] : 0.0f;
        u32x4 w[NP];
        q_split_pack<DT>(q, w);
#pragma unroll
        for (int k = 0; k < NP; ++k) qp[k][4 * cb + j] = w[k];
      }
    }
  }
  // ---- descriptor lanes: lane j = first token row of the wave's j-th doc (row = pid * L: no descriptor lookup) ---------
  uint32_t row0 = 0;
  bool bad = true;
  if (lane < ndoc) {
    const int64_t pid = p.cand[(int64_t)qi * p.ncand + c_begin + lane];
    const bool ok = pid >= 0 && pid < p.n_docs && (pid + 1) * L <= p.n_tokens;
    row0 = ok ? (uint32_t)(pid * L) : 0u;
    bad = !ok;
  }
  char* const wlds = lds + wave * (NT * TILE);
  const char* const tok = (const char*)p.index;
  const int ntile = (ndoc + DPT - 1) / DPT;
  // DMA instruction i moves tile rows 4 i + (lane >> 4); chunk position c of row s receives source chunk c ^ (s & 15):
  // per lane  off_i = ds0 * 256 + 16 * ((dch ^ ds0) ^ (4 i & 15)) = lane_off ^ ((i & 3) << 6)     (ds0 < 4: no carry into 4 i)
  const uint32_t lane_off = (uint32_t)((lane >> 4) * ROWB + 16 * ((lane & 15) ^ (lane >> 4)));
  auto issue = [&](int t, int buf) __attribute__((always_inline)) {
    uint32_t base[DPT];
#pragma unroll
    for (int d = 0; d < DPT; ++d) base[d] = (uint32_t)__builtin_amdgcn_readlane((int)row0, min(t * DPT + d, ndoc - 1));
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const char* const g = tok + (uint64_t)(base[(4 * i) / L] + (uint32_t)((4 * i) % L)) * ROWB;
      __builtin_amdgcn_global_load_lds(GPTR(g + (lane_off ^ (uint32_t)((i & 3) << 6))), LPTR(wlds + buf * TILE + i * 1024), 16, 0, CPOL_STREAM);
    }
  };
#pragma unroll
  for (int j = 0; j < NT; ++j)
    if (j < ntile) issue(j, j);
  float myscore = 0.0f;
  int buf = 0;
  for (int t = 0; t < ntile; ++t) {
    __builtin_amdgcn_s_setprio(0);
    // tiles t + 1 .. t + NT - 1 were issued after this one (when they exist)
    if (NT == 2 && t + 1 < ntile) wait_vmcnt<NDMA>(); else wait_vmcnt<0>();
    // operand (row block b, k group j): row 16 b + n16, chunk 4 j + kq
    u32x4 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *(const u32x4*)(wlds + buf * TILE + (16 * (i >> 2) + n16) * ROWB + 16 * ((4 * (i & 3) + kq) ^ n16));
    wait_lgkmcnt0();
    if (t + NT < ntile) issue(t + NT, buf);
    buf = (NT == 1 || buf == 1) ? 0 : 1;
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x4 acc0[NCB], acc1[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc0[cb] = acc1[cb] = (f32x4)(0.0f);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          if constexpr (DT == MAXSIM_F16) {
            const f16x8 av = __builtin_bit_cast(f16x8, a[4 * b + j]);
            acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, qp[0][4 * cb + j]), acc0[cb], 0, 0, 0);
            acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, __builtin_bit_cast(f16x8, qp[1][4 * cb + j]), acc1[cb], 0, 0, 0);
          } else {
            const bf16x8 av = __builtin_bit_cast(bf16x8, a[4 * b + j]);
            acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, qp[0][4 * cb + j]), acc0[cb], 0, 0, 0);
            acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, qp[1][4 * cb + j]), acc1[cb], 0, 0, 0);
            acc1[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf16x8, qp[NP - 1][4 * cb + j]), acc1[cb], 0, 0, 0);
          }
        }
      // similarity of query token 16 cb + n16 with tile row 16 b + 4 kq + v; doc d of the block = rows [d L, (d + 1) L)
      float tree[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        float sv[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) sv[v] = (DT == MAXSIM_F16) ? fmaf(acc1[cb][v], 1.0f / 2048.0f, acc0[cb][v]) : (acc0[cb][v] + acc1[cb][v]);
        float m = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));  // the quarter's 4 rows
        if constexpr (L >= 8) {  // quarters g, g + 1
          const uint32_t xb = __float_as_uint(m);
          const auto s16 = __builtin_amdgcn_permlane16_swap(xb, xb, false, false);
          m = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
        }
        if constexpr (L == 16) {  // quarters g, g + 2
          const uint32_t xb = __float_as_uint(m);
          const auto s32 = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
          m = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
        }
        m += dpp_f32<0xB1>(m);  // the general kernels' pairwise tree over the block's 16 query-token lanes
        m += dpp_f32<0x4E>(m);
        m += dpp_f32<0x141>(m);
        m += dpp_f32<0x140>(m);
        tree[cb] = m;
      }
#pragma unroll
      for (int d = 0; d < DPB; ++d) {  // doc d's sums sit in the 16-lane row that holds its rows: lane d * (64 / DPB)
        float sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tree[0]), d * (64 / DPB)));
        // (one block: the general kernel adds the all-zero second block's +0.0 -- kept, so that even a -0.0 sum agrees)
        sc += (NCB == 2) ? __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(tree[NCB - 1]), d * (64 / DPB))) : 0.0f;
        myscore = (lane == t * DPT + b * DPB + d) ? sc : myscore;
      }
    }
  }
  if (lane < ndoc) srow[lane] = (p.accum ? srow[lane] : 0.0f) + (bad ? NEG_INF : myscore);
  };  // wave_item
  if constexpr (LIST) {  // (slot -> item as in k_maxsim_stream's list form: XCD x walks the x-th eighth of the list)
    const int32_t* const wl = (const int32_t*)p.worklist;
    const int wl_total = uni(wl[0]);
    const int2* const wl_items = (const int2*)(wl + worklist_items_word(p.nq));
    const int J = (wl_total + WAVES - 1) / WAVES, Jx = (J + 7) >> 3;
    for (int s = (int)blockIdx.x; s < 8 * Jx; s += (int)gridDim.x) {
      const int item = ((s & 7) * Jx + (s >> 3)) * WAVES + wave;
      if ((s >> 3) >= Jx || item >= wl_total) continue;
      const int2 e = wl_items[item];
      wave_item(uni(e.x), uni(e.y) & ((1 << WL_SLOT_BITS) - 1), uni(e.y) >> WL_SLOT_BITS);
    }
  } else {
    int qi, chunk;
    wg_to_work((int)blockIdx.x, p.nq, p.nchunk, qi, chunk);
    const int dpwv = p.dpw / WAVES;
    const int c_begin = chunk * p.dpw + wave * dpwv;
    wave_item(qi, c_begin, max(0, min(dpwv, p.ncand - c_begin)));
  }
}

}  // namespace maxsim
