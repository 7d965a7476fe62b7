// maxsim_stream_bigh.h -- the streaming MFMA kernel with the query tile in LDS: h = 128 * KB (KB = 1..8, e.g. the
// reference's default dim 768, proj_conf/dense.yaml:8), Lq <= 32, fp32 / fp16 / bf16 token matrix.
//
// Same token-stream structure as maxsim_stream.h (packed 32-row tiles, descriptor lanes, per-wave LDS-DMA ring, DPP
// reduce), with the contraction split into KB blocks of 128 dims: a "sub-tile" is 32 rows x one 128-dim block (each
// row contributes >= 256 contiguous bytes = whole cache lines), the accumulators live across the KB sub-tiles of a
// tile, and the per-document reduction runs after the last block.  The query does not fit in registers at this
// width, so the workgroup stages it ONCE in LDS in MFMA B-operand order -- per 128-dim block the same swizzled
// [32 rows][128 dims] image as a doc sub-tile, so A and B operand reads use the same lane offsets -- as NPQ
// pieces: 1 when the query arrives in the index's own 16-bit type (exact), else hi + lo (fp16: lo pre-scaled by
// 2^11, exact to 2^-22; bf16: 16 significant bits, |error| ~1e-5 per token, inside the 1e-3 tolerance stated for
// 16-bit inputs); fp32 index: the fp32 query itself (exact f32 MFMA chain).
//
// MODE_DENSE (all-pairs BaseModel.score with masks, any of the three dtypes) runs here too: Q * q_mask is folded
// into the staged query image; D * d_mask multiplies the A operands (fp32, exactly as BaseModel.py:41) or the
// finished similarities (16-bit inputs; identical for 0/1 masks).  In the all-pairs form every query meets every
// doc, so a workgroup takes QB queries at once (QB query images in LDS, QB accumulator sets): each doc sub-tile is
// fetched and read into registers once for all of them.  AM = true additionally records each query token's arg-max
// doc token (training-form forward, see maxsim_backward.h).
#pragma once
#include "maxsim_stream.h"

namespace maxsim {

// Reduction state for QB queries that walk the same documents (dense mode), optionally with arg-max tracking.
// SPLIT (small launches, QB = 1): a doc is streamed by several waves, each a slice of its tokens; a finished slice PARKS its
// per-query-token maxima in a register (parked[doc ordinal], lanes = query tokens) instead of finishing the score -- the
// workgroup combines the slices after the stream (k_maxsim_stream_bigh).
// HALFQ (at most 16 query tokens, image of 16 rows): the 32 columns of the accumulators are the 16 tokens twice -- the upper
// copy (lanes 16..31 of each half) is dropped before the sum over query tokens: it adds the +0.0 a zero query row would.
template <int QB, bool AM, bool SPLIT = false, bool HALFQ = false>
struct MultiReducer {
  float rmax[QB], myscore[QB];
  float parked[SPLIT ? SPLIT_MAX_DOCS : 1];
  int ridx[QB];
  int jdoc;
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int x = 0; x < QB; ++x) {
      rmax[x] = NEG_INF;
      myscore[x] = 0.0f;
      ridx[x] = 0;
    }
#pragma unroll
    for (int k = 0; k < (SPLIT ? SPLIT_MAX_DOCS : 1); ++k) parked[k] = NEG_INF;
    jdoc = 0;
  }
  // argdoc[x]: &argmax[(query x, this doc) * Lq] (AM only)
  __device__ __forceinline__ void finish_doc(const Cursor& C, int lane, int32_t* const (&argdoc)[QB], int Lq) {
    if constexpr (SPLIT) {  // both lane halves -> the token's maximum over this slice, parked under the doc's ordinal
      const uint32_t xb = __float_as_uint(rmax[0]);
      const auto sw = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
      const float v = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
#pragma unroll
      for (int k = 0; k < SPLIT_MAX_DOCS; ++k) parked[k] = (jdoc == k && C.kind == 0) ? v : parked[k];
      rmax[0] = NEG_INF;
      ++jdoc;
      return;
    }
#pragma unroll
    for (int x = 0; x < QB; ++x) {
      float sc;
      if (C.kind == 0) {
        const uint32_t xb = __float_as_uint(rmax[x]);
        const auto sw = __builtin_amdgcn_permlane32_swap(xb, xb, false, false);
        const float a = __uint_as_float(sw[0]), b = __uint_as_float(sw[1]);
        float v;
        if constexpr (AM) {
          const auto si = __builtin_amdgcn_permlane32_swap((uint32_t)ridx[x], (uint32_t)ridx[x], false, false);
          const int ia = (int)si[0], ib = (int)si[1];
          const bool take_b = (b > a) || (b == a && ib < ia);
          v = take_b ? b : a;
          if (lane < Lq) argdoc[x][lane] = take_b ? ib : ia;
        } else {
          v = fmaxf(a, b);
        }
        if (C.floor0) v = fmaxf(v, 0.0f);
        if constexpr (HALFQ) v = (lane & 16) ? 0.0f : v;
        v += dpp_f32<0xB1>(v);
        v += dpp_f32<0x4E>(v);
        v += dpp_f32<0x141>(v);
        v += dpp_f32<0x140>(v);
        sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0)) +
             __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
      } else {
        sc = C.kind == 1 ? 0.0f : NEG_INF;
      }
      myscore[x] = (lane == jdoc) ? sc : myscore[x];
      rmax[x] = NEG_INF;
      ridx[x] = 0;
    }
    ++jdoc;
  }
  // sv[x][v] = similarity of query x's token (this lane) with tile row (v & 3) + 8 (v >> 2) + 4 (lane >> 5);
  // argq[x]: &argmax[(query x, first doc of this wave) * Lq]
  __device__ __forceinline__ void reduce_tile(const float (&sv)[QB][16], Cursor& C, const DocLanes& dl, int lane,
                                              int32_t* const (&argq)[QB], int Lq) {
    const int hh = lane >> 5;
    int filled = 0;
    while (filled < 32 && C.valid) {
      const int take = uni(min(32 - filled, max(C.len - C.pos, 0)));  // 0: empty doc / padding slot
      if (!AM && take == 32) {
#pragma unroll
        for (int x = 0; x < QB; ++x) {
          const float t0 = fmaxf(fmaxf(sv[x][0], sv[x][1]), fmaxf(sv[x][2], sv[x][3]));
          const float t1 = fmaxf(fmaxf(sv[x][4], sv[x][5]), fmaxf(sv[x][6], sv[x][7]));
          const float t2 = fmaxf(fmaxf(sv[x][8], sv[x][9]), fmaxf(sv[x][10], sv[x][11]));
          const float t3 = fmaxf(fmaxf(sv[x][12], sv[x][13]), fmaxf(sv[x][14], sv[x][15]));
          rmax[x] = fmaxf(rmax[x], fmaxf(fmaxf(t0, t1), fmaxf(t2, t3)));
        }
      } else if (take > 0) {  // rows [filled, filled + take) only (and every row when arg-max is tracked)
        const int nbase = C.pos - filled;  // slot s of this tile is token nbase + s of the current doc
        const uint32_t lo = (uint32_t)(filled - 4 * hh), n_in = (uint32_t)take;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const uint32_t rel = (uint32_t)((v & 3) + 8 * (v >> 2)) - lo;
          const bool in = rel < n_in;
#pragma unroll
          for (int x = 0; x < QB; ++x) {
            if constexpr (AM) {
              const bool better = in && (sv[x][v] > rmax[x]);  // strict: the first maximal token wins
              rmax[x] = better ? sv[x][v] : rmax[x];
              ridx[x] = better ? nbase + (v & 3) + 8 * (v >> 2) + 4 * hh : ridx[x];
            } else {
              rmax[x] = fmaxf(rmax[x], in ? sv[x][v] : NEG_INF);
            }
          }
        }
      }
      filled += take;
      C.pos += take;
      if (C.pos >= C.len) {  // doc complete (empty docs / padding slots are scored on the spot)
        int32_t* argdoc[QB];
#pragma unroll
        for (int x = 0; x < QB; ++x) argdoc[x] = AM ? argq[x] + (int64_t)C.j * Lq : nullptr;
        finish_doc(C, lane, argdoc, Lq);
        C.next_doc(dl);
      }
    }
  }
  __device__ __forceinline__ void drain(Cursor& C, const DocLanes& dl, int lane, int32_t* const (&argq)[QB], int Lq) {
    while (C.valid) {  // trailing empty docs / padding slots
      int32_t* argdoc[QB];
#pragma unroll
      for (int x = 0; x < QB; ++x) argdoc[x] = AM ? argq[x] + (int64_t)C.j * Lq : nullptr;
      finish_doc(C, lane, argdoc, Lq);
      C.next_doc(dl);
    }
  }
};

// PART: h is not a multiple of 128 (but of 16 bytes): the last block is partial -- the query image is zero past h and
// the doc fetch never leaves the row (issue_rows<PART>).
// LIST (counted candidate rows, maxsim_worklist.h): a fixed grid whose WORKGROUPS walk the device-built list of workgroup
//   items (query, first slot, docs); an item's docs are dealt evenly over the waves, the query image is staged per item.
// SPLITK (small launches -- the reference's online call is ONE query x ~1000 docs): a doc is streamed by p.split (2 or 4)
//   waves of the workgroup, each a slice of whole 32-row tiles.  With one wave per doc the launch lasts as long as its
//   LONGEST doc takes one wave (dim 768: 384 tokens = 72 sub-tiles of ~1.5 us), whatever the average.
// BAL (static-grid rerank of a ragged index): the workgroup's docs are dealt to its waves by TOKENS instead of by count
//   (k_maxsim_stream's BAL: descriptors of all the workgroup's docs one per lane, a scan of the lengths, contiguous runs by
//   midpoint); bit-identical scores.
// HALFQ (at most 16 query tokens -- the multi-view configuration's q_view = 16, dense.yaml:31 -- on a 16-bit index): the query
//   image holds 16 rows instead of 32 (48 KiB instead of 96 at dim 768 with a two-piece image), the B-operand reads of query
//   columns 16..31 alias columns 0..15 and the reducer drops them: the same arithmetic for the 16 real tokens (bit-identical
//   scores), and the LDS it frees goes to the waves' rings -- at 16 tokens per doc the launch is bound by how many waves
//   stream, not by bytes per wave (mv768: 8 waves x 1 sub-tile 0.755, 12 x 1 with the small image 0.79+).
template <int MODE, int DT, int NPQ, int WAVES, int NT, bool AM, int QB, bool PART = false, bool LIST = false, bool SPLITK = false, bool BAL = false,
          bool HALFQ = false>
__global__ void __launch_bounds__(WAVES * 64) k_maxsim_stream_bigh(KARGS_DECL) {
  static_assert(!BAL || (MODE == MODE_RERANK && QB == 1 && !LIST && !SPLITK), "token-balanced cut: static-grid rerank only");
  static_assert(!HALFQ || (MODE == MODE_RERANK && QB == 1 && !AM && !PART && !SPLITK && DT != MAXSIM_F32), "16-row query image: 16-bit rerank");
  static_assert(!LIST || (MODE == MODE_RERANK && QB == 1 && !AM), "work-list form: rerank, one query per workgroup");
  static_assert(!SPLITK || (MODE == MODE_RERANK && QB == 1 && !AM && !LIST && !PART), "split form: static-grid rerank");
  static_assert(DT != MAXSIM_F32 || NPQ == 1, "fp32 index: the fp32 query is used as is");
  static_assert(!AM || MODE == MODE_DENSE, "arg-max tracking is a dense (training-form) feature");
  static_assert(QB == 1 || MODE == MODE_DENSE, "several queries share documents only in the all-pairs form");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  using T = StreamTraits<DT>;
  constexpr int BLKB = T::ROWB;  // bytes of one 128-dim block of a row
  constexpr int SUB = T::TILE;   // bytes of a sub-tile (32 rows x one block) == one query piece block
  constexpr int NDMA = T::NDMA, RPD = T::RPD, LPR = T::LPR, NRD = T::NRD;
  constexpr int ESZ = BLKB / 128;
  const int KB = (p.h + 127) >> 7;
  const uint32_t rowbytes = (uint32_t)p.h * ESZ;
  const int lane = threadIdx.x & 63;
  const int wave = uni(threadIdx.x >> 6);
  // one workgroup item: the wave's candidates [c_begin, c_begin + ndoc) of the queries q0 .. q0 + QB - 1
  const int split = SPLITK ? p.split : 1;
  const int team = SPLITK ? wave / split : wave, part = SPLITK ? wave - team * split : 0;
  auto wg_item = [&](const int q0, const int c_begin, const int ndoc, const DocLanes* pre = nullptr) __attribute__((always_inline)) {
  DocLanes dl = pre ? *pre : load_doc_lanes<MODE>(p, q0, c_begin, ndoc, lane);
  if constexpr (SPLITK) {  // this wave's slice of every doc: whole 32-row tiles, cut as evenly as possible
    const int per = (((dl.len + 31) >> 5) + split - 1) / split * 32;
    const int start = min(dl.len, part * per);
    dl.row0 += (uint32_t)start;
    dl.len = min(dl.len - start, per);  // may be 0: the slice then parks -inf maxima (neutral)
  }
  constexpr int QROWS = HALFQ ? 16 : 32;                    // query tokens held in the image
  constexpr int SUBQ = QROWS * BLKB;                        // bytes of one 128-dim block of it
  const int qimg = NPQ * KB * SUBQ;                         // one query's image: [NPQ][KB][QROWS rows][BLKB]
  char* const qlds = lds;
  char* const wlds = lds + QB * qimg + wave * (NT * SUB);
  const int r = lane & 31, hh = lane >> 5;
  const bool masked = MODE == MODE_DENSE && p.mask_dtype != MAXSIM_MASK_NONE;

  const int rsw = r & 15;
  const int rdbase = r * BLKB;
  const char* const tok = (const char*)p.index;

  Cursor F, C;
  F.init(dl, ndoc);
  C = F;

  // fetch side: the tile being fetched block by block
  TileMap ft;
  ft.myrow = 0; ft.base0 = 0; ft.base1 = 0; ft.split = 32; ft.kind = 0;
  int fkb = 0;
  auto fetch_next = [&](int buf) __attribute__((always_inline)) -> bool {
    if (fkb == 0) ft = fill_tile(F, dl, r);
    if (ft.kind == 0) return false;
    issue_rows<NDMA, RPD, LPR, PART, MODE == MODE_RERANK ? CPOL_STREAM : 0>(tok, rowbytes, (uint32_t)fkb * BLKB, wlds + buf * SUB, ft, lane);
    fkb = (fkb + 1 == KB) ? 0 : fkb + 1;
    return true;
  };

  // prologue fetches are issued BEFORE the query images are staged: the staging latency overlaps the first fetch
  int nissued = 0, nconsumed = 0;
  bool prev_issued = false;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const bool ok = fetch_next(j);
    nissued += ok ? 1 : 0;
    prev_issued = ok;
  }

  // ---- stage the query tiles in LDS (all waves), B-operand order, swizzled like a doc sub-tile -------------------
  {
    constexpr int CPB = BLKB / 16;       // 16-byte chunks per row block
    constexpr int EPC = 16 / ESZ;        // elements per chunk
    const int nchunks = KB * QROWS * CPB;
    for (int idx = threadIdx.x; idx < QB * nchunks; idx += WAVES * 64) {
      const int x = idx / nchunks;
      const int rem = idx - x * nchunks;
      const int c = rem % CPB;
      const int n = (rem / CPB) % QROWS;
      const int kb = rem / (CPB * QROWS);
      const int qi = min(q0 + x, p.nq - 1);
      int qlen = p.Lq;
      if (MODE == MODE_RERANK && p.q_len) qlen = min(qlen, p.q_len[qi]);
      const int qtok = p.q_tok0 + n;  // queries longer than 32 tokens: one launch per 32
      const bool live = q_token_live<MODE>(p, qi, qtok, qlen);
      const int64_t src = ((int64_t)qi * p.Lq + (live ? qtok : 0)) * p.h + kb * 128 + c * EPC;
      const float qs = (masked && live) ? load_mask(p.q_mask, p.mask_dtype, (int64_t)qi * p.Lq + qtok) : 1.0f;
      float q[EPC];
#pragma unroll
      for (int j = 0; j < EPC; ++j) {
        const bool indim = !PART || (kb * 128 + c * EPC + j < p.h);
        q[j] = (live && indim) ? load_q(p.Q, p.q_dtype, src + (indim ? j : 0)) : 0.0f;
        if (MODE == MODE_DENSE) q[j] *= qs;  // Q * q_mask[..., None], BaseModel.py:42
      }
      char* dst = qlds + x * qimg + kb * SUBQ + n * BLKB + 16 * (c ^ (n & 15));
      if constexpr (DT == MAXSIM_F32) {
        *(f32x4*)dst = f32x4{q[0], q[1], q[2], q[3]};
      } else {
        uint16_t pc[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr (DT == MAXSIM_F16) {
            _Float16 hi = (_Float16)q[j];
            _Float16 lo = (_Float16)((q[j] - (float)hi) * 2048.0f);
            __builtin_memcpy(&pc[0][j], &hi, 2);
            __builtin_memcpy(&pc[1][j], &lo, 2);
          } else {
            pc[0][j] = f32_to_bf16_rn(q[j]);
            pc[1][j] = f32_to_bf16_rn(q[j] - bf16_to_f32(pc[0][j]));
          }
        }
#pragma unroll
        for (int k = 0; k < NPQ; ++k) {
          u32x4 w;
#pragma unroll
          for (int y = 0; y < 4; ++y) w[y] = (uint32_t)pc[k][2 * y] | ((uint32_t)pc[k][2 * y + 1] << 16);
          *(u32x4*)(dst + k * KB * SUBQ) = w;
        }
      }
    }
  }
  __syncthreads();  // the only workgroup barrier: the query images are read-only from here on

  MultiReducer<QB, AM, SPLITK, HALFQ> red;
  red.init();
  int32_t* argq[QB];
#pragma unroll
  for (int x = 0; x < QB; ++x)
    argq[x] = AM ? p.argmax + ((int64_t)min(q0 + x, p.nq - 1) * p.ncand + c_begin) * p.Lq : nullptr;
  int buf = 0, ckb = 0;
  float mv = 1.0f;  // dense: d_mask value of this lane's row slot in the tile being consumed
  f32x16 acc0[QB], acc1[NPQ == 2 ? QB : 1];
#pragma unroll
  for (int x = 0; x < QB; ++x) acc0[x] = (f32x16)(0.0f);
#pragma unroll
  for (int x = 0; x < (NPQ == 2 ? QB : 1); ++x) acc1[x] = (f32x16)(0.0f);

  while (nconsumed < nissued) {
    __builtin_amdgcn_s_setprio(0);
    if (prev_issued) wait_vmcnt<NDMA * (NT - 1)>(); else wait_vmcnt<0>();
    const char* tl = wlds + buf * SUB + rdbase;
    u32x4 a[NRD];
#pragma unroll
    for (int i = 0; i < NRD; ++i) a[i] = *(const u32x4*)(tl + 16 * ((2 * i + hh) ^ rsw));
    wait_lgkmcnt0();
    {
      const bool ok = fetch_next(buf);
      nissued += ok ? 1 : 0;
      prev_issued = ok;
    }
    buf = (buf + 1 == NT) ? 0 : buf + 1;
    __builtin_amdgcn_s_setprio(3);  // contraction phase at raised priority (see maxsim_stream.h)

    if constexpr (MODE == MODE_DENSE) {
      if (masked && ckb == 0) {  // first block of a tile: look up the tile's 32 mask values (one per lane pair)
        Cursor Cp = C;
        const TileMap ct = fill_tile(Cp, dl, r);
        mv = load_mask(p.d_mask, p.mask_dtype, (int64_t)ct.myrow);
      }
      if constexpr (DT == MAXSIM_F32) {
        if (masked) {  // D * d_mask[..., None], BaseModel.py:41
#pragma unroll
          for (int i = 0; i < NRD; ++i) a[i] = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, a[i]) * mv);
        }
      }
    }

#pragma unroll
    for (int x = 0; x < QB; ++x) {
      // query x, this block: same lane offsets as the doc image (HALFQ: query columns 16..31 read columns 0..15 again)
      const char* qb = qlds + x * qimg + ckb * SUBQ + (HALFQ ? rsw * BLKB : rdbase);
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        const int qoff = 16 * ((2 * i + hh) ^ rsw);
        if constexpr (DT == MAXSIM_F32) {
          const f32x4 av = __builtin_bit_cast(f32x4, a[i]);
          const f32x4 bv = *(const f32x4*)(qb + qoff);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc0[x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc0[x], 0, 0, 0);
        } else if constexpr (DT == MAXSIM_F16) {
          const f16x8 av = __builtin_bit_cast(f16x8, a[i]);
          acc0[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, *(const f16x8*)(qb + qoff), acc0[x], 0, 0, 0);
          if constexpr (NPQ == 2)
            acc1[x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, *(const f16x8*)(qb + KB * SUBQ + qoff), acc1[x], 0, 0, 0);
        } else {
          const bf16x8 av = __builtin_bit_cast(bf16x8, a[i]);
          acc0[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, *(const bf16x8*)(qb + qoff), acc0[x], 0, 0, 0);
          if constexpr (NPQ == 2)
            acc1[x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, *(const bf16x8*)(qb + KB * SUBQ + qoff), acc1[x], 0, 0, 0);
        }
      }
    }
    ckb = (ckb + 1 == KB) ? 0 : ckb + 1;
    if (ckb == 0) {  // last block of the tile: similarities are complete
      float sv[QB][16];
#pragma unroll
      for (int x = 0; x < QB; ++x)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          if constexpr (NPQ == 2)
            sv[x][v] = (DT == MAXSIM_F16) ? fmaf(acc1[x][v], 1.0f / 2048.0f, acc0[x][v]) : (acc0[x][v] + acc1[x][v]);
          else
            sv[x][v] = acc0[x][v];
        }
      if constexpr (MODE == MODE_DENSE && DT != MAXSIM_F32) {
        if (masked) {  // 16-bit inputs: the mask multiplies the finished similarity of its row
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int s0 = (v & 3) + 8 * (v >> 2);
            const float m0 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mv), s0));
            const float m1 = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(mv), s0 + 4));
#pragma unroll
            for (int x = 0; x < QB; ++x) sv[x][v] *= hh ? m1 : m0;
          }
        }
      }
      red.reduce_tile(sv, C, dl, lane, argq, p.Lq);
#pragma unroll
      for (int x = 0; x < QB; ++x) acc0[x] = (f32x16)(0.0f);
#pragma unroll
      for (int x = 0; x < (NPQ == 2 ? QB : 1); ++x) acc1[x] = (f32x16)(0.0f);
    }
    ++nconsumed;
  }
  red.drain(C, dl, lane, argq, p.Lq);
  if constexpr (SPLITK) {
    // the slices meet: every wave writes its parked maxima ([doc ordinal][32 query tokens]) to its own, now idle, ring;
    // wave `part == 0` of a team combines them doc by doc: max over the slices, then exactly the unsplit wave's floor
    // + sum tree (bit-identical scores)
    float* const mine = (float*)wlds;
#pragma unroll
    for (int k = 0; k < SPLIT_MAX_DOCS; ++k)
      if (lane < 32) mine[k * 32 + lane] = red.parked[k];
    __syncthreads();
    if (part == 0) {
      float my = 0.0f;
      for (int j = 0; j < ndoc; ++j) {
        const int fl = __builtin_amdgcn_readlane(dl.flags, j);
        float sc;
        if ((fl & 3) == 0) {
          float v = NEG_INF;
          for (int sp = 0; sp < split; ++sp)
            v = fmaxf(v, ((const float*)(lds + QB * qimg + (wave + sp) * (NT * SUB)))[j * 32 + (lane & 31)]);
          if (fl >> 2) v = fmaxf(v, 0.0f);
          v += dpp_f32<0xB1>(v);
          v += dpp_f32<0x4E>(v);
          v += dpp_f32<0x141>(v);
          v += dpp_f32<0x140>(v);
          sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0)) +
               __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
        } else {
          sc = (fl & 3) == 1 ? 0.0f : NEG_INF;
        }
        my = (lane == j) ? sc : my;
      }
      if (lane < ndoc) {
        float* const cell = p.scores + (int64_t)q0 * p.ncand + c_begin + lane;
        *cell = (p.accum ? *cell : 0.0f) + my;
      }
    }
    return;
  }
#pragma unroll
  for (int x = 0; x < QB; ++x)
    if (lane < red.jdoc && q0 + x < p.nq) {
      float* const cell = p.scores + (int64_t)(q0 + x) * p.ncand + c_begin + lane;
      *cell = (p.accum ? *cell : 0.0f) + red.myscore[x];
    }
  };  // wg_item
  if constexpr (LIST) {
    // workgroup slot s = blockIdx.x, + gridDim.x, ... (the grid is a multiple of 8); slot -> item is XCD-aware as in the
    // h = 128 list kernel: the slots with s % 8 = x walk the x-th eighth of the list in order
    const int32_t* const wl = (const int32_t*)p.worklist;
    const int J = uni(wl[0]), Jx = (J + 7) >> 3;
    const int2* const wl_items = (const int2*)(wl + worklist_items_word(p.nq));
    for (int s = (int)blockIdx.x; s < 8 * Jx; s += (int)gridDim.x) {
      const int item = (s & 7) * Jx + (s >> 3);
      if ((s >> 3) >= Jx || item >= J) continue;
      const int2 e = wl_items[item];
      const int first = uni(e.y) & ((1 << WL_SLOT_BITS) - 1), n = uni(e.y) >> WL_SLOT_BITS;
      const int b0 = (int)((int64_t)wave * n / WAVES), b1 = (int)((int64_t)(wave + 1) * n / WAVES);
      __builtin_amdgcn_s_setprio(0);
      wg_item(uni(e.x), first + b0, b1 - b0);
      __syncthreads();  // every wave is done with this item's query image: the next item may stage its own
    }
  } else {
    int qblk, chunk;
    if constexpr (MODE == MODE_RERANK) {
      wg_to_work((int)blockIdx.x, (p.nq + QB - 1) / QB, p.nchunk, qblk, chunk);
    } else {  // all-pairs: query-block-major (a doc chunk's query blocks close in time re-read it from cache)
      qblk = blockIdx.x / p.nchunk;
      chunk = blockIdx.x - qblk * p.nchunk;
    }
    if constexpr (BAL) {
      const int c0 = chunk * p.dpw, nwg = max(0, min(p.dpw, p.ncand - c0));     // the workgroup's docs: one per lane (dpw <= 64)
      const DocLanes all = load_doc_lanes<MODE>(p, qblk, c0, nwg, lane);
      int incl = all.len;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        incl += lane >= d ? t : 0;
      }
      const int64_t T = __builtin_amdgcn_readlane(incl, 63);
      const int64_t mid2 = 2 * (int64_t)(incl - all.len) + all.len;
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
      wg_item(qblk, c0 + s0, s1 - s0, &mine);
    } else {
    const int dpwv = SPLITK ? p.dpw / (WAVES / split) : p.dpw / WAVES;  // docs per wave / per team (SPLITK: <= SPLIT_MAX_DOCS)
    const int c_begin = chunk * p.dpw + team * dpwv;
    // (queries past nq - 1 are clamped and not written)
    wg_item(qblk * QB, c_begin, max(0, min(dpwv, p.ncand - c_begin)));
    }
  }
}

}  // namespace maxsim
