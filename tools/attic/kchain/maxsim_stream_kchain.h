// maxsim_stream_kchain.h -- rerank on wide 16-bit token rows (h = 128 * KB: the reference's default dim 768,
// proj_conf/dense.yaml:8, on its fp16 index, encoder.py:175) with the QUERY IN REGISTERS: the contraction of a 32-row
// tile is handed from wave to wave, one 128-dim block per wave.
//
// The LDS-query kernel (maxsim_stream_bigh.h) stages the query image in LDS -- 96 KiB for an fp32 query at dim 768 (two
// 16-bit pieces) -- which leaves 64 KiB of the CU's 160 KiB for the rings: 8 KiB in flight per wave, 64 KiB per CU, and a
// one-query launch lasts as long as its longest doc takes ONE wave at one memory latency per sub-tile.
//
// Here a workgroup is KB compute waves + one control wave, and walks the packed 32-row tiles of its docs in steps:
//   * compute wave kb owns block kb of the contraction: its 8 k-groups of the query (32 tokens x 128 dims -> 8 x NPQ
//     B-operand registers of 4 VGPRs) stay in registers for the whole launch -- no query image in LDS, no B-operand reads
//     (the LDS-query kernel reads 16 KiB of them per sub-tile);
//   * it streams sub-tile (t, kb) of tile t through its own ring, NT sub-tiles deep (KB x NT x 8 KiB in flight per CU);
//   * the accumulators of tile t travel wave 0 -> 1 -> ... -> KB-1 through one LDS slot per hop, one hop per step: wave kb
//     works on tile (s - kb) in step s.  Every accumulator sees block 0's k-groups, then block 1's, ... in order -- the
//     SAME fp32 chain as the LDS-query kernel's single wave: scores are bit-identical to it;
//   * the last compute wave owns the per-document reduction (packed tiles, descriptor lanes, DPP sum tree: maxsim_stream.h);
//   * the control wave walks the docs' descriptor lanes once for everybody: it lays the stream's rows onto tiles
//     (fill_tile) and publishes each tile's map in a small LDS ring a few steps before the compute waves request the tile.
//     A compute wave's step is then ~110 instructions (8 A-operand reads, 8 x NPQ MFMAs, 8 LDS-DMA requests with a scalar
//     base and four loop-invariant lane offsets, the hand-over): with one or two waves per SIMD and the workgroup in step,
//     the instruction stream of a step IS the step's duration.
// Two bare s_barrier per step order the hand-over slots and the map ring (read | write | read ...); the rings are private
// to their wave.  Reference semantics as everywhere: colbert_ranker.py:88-112 (buckets, pad, mask) + BaseModel.py:41-45.
#pragma once
#include "maxsim_stream_bigh.h"

#ifndef KCHAIN_ABLATE  // timing experiments (wrong scores): 1 = no fetch in the loop, 2 = no MFMAs, 3 = no hand-over
#define KCHAIN_ABLATE 0
#endif

namespace maxsim {

__device__ __forceinline__ void kchain_barrier() {  // this wave's LDS accesses are done, then the bare barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int KCHAIN_MAPD = 16;    // tile maps in the ring (> KB + NT + 1)
constexpr int KCHAIN_MAPB = 144;   // one map: {kind, base0, base1, split} + the 32 slots' token rows

template <int DT, int NPQ, int KB, int NT>
__global__ void __launch_bounds__((KB + 1) * 64) k_maxsim_stream_kchain(KARGS_DECL) {
  static_assert(DT == MAXSIM_F16 || DT == MAXSIM_BF16, "16-bit token rows");
  static_assert(NPQ == 1 || NPQ == 2, "query: the index's own 16-bit type, or hi + lo pieces of an fp32 query");
  static_assert(KB + NT + 1 < KCHAIN_MAPD, "map ring depth");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  KARGS_TO_PARAMS;
  using T = StreamTraits<DT>;
  constexpr int BLKB = T::ROWB, SUB = T::TILE, NDMA = T::NDMA, RPD = T::RPD, LPR = T::LPR, NRD = T::NRD;
  static_assert(NRD == 8 && NDMA == 8 && RPD == 4 && LPR == 16, "eight k-groups / eight 4-row requests per 128-dim block");
  constexpr int HOP = NPQ * 4096;  // one hand-over slot: NPQ x 16 accumulators x 64 lanes
  const int lane = threadIdx.x & 63;
  const int wv = uni(threadIdx.x >> 6);
  const uint32_t rowbytes = (uint32_t)p.h * 2u;
  int qi, chunk;
  wg_to_work((int)blockIdx.x, p.nq, p.nchunk, qi, chunk);
  const int c_begin = chunk * p.dpw;
  const int ndoc = max(0, min(p.dpw, p.ncand - c_begin));
  char* const hops = lds + KB * NT * SUB;
  char* const maps = hops + (KB - 1) * HOP;
  const int r = lane & 31, hh = lane >> 5;

  if (wv == KB) {
    // ================= control wave: the docs' rows -> tiles, published NT + 1 tiles ahead of the first request
    const DocLanes dl = load_doc_lanes<MODE_RERANK>(p, qi, c_begin, ndoc, lane);
    Cursor F;
    F.init(dl, ndoc);
    bool ended = false;
    int ntiles = 0;
    auto publish = [&](int m) __attribute__((always_inline)) {
      TileMap t;
      t.myrow = 0; t.base0 = 0; t.base1 = 0; t.split = 32; t.kind = 0;
      if (!ended) {
        t = fill_tile(F, dl, r);
        if (t.kind == 0) { ended = true; ntiles = m; }
      }
      char* const e = maps + (m & (KCHAIN_MAPD - 1)) * KCHAIN_MAPB;
      if (lane == 0) *(u32x4*)e = u32x4{(uint32_t)t.kind, t.base0, t.base1, (uint32_t)t.split};
      if (lane < 32) ((uint32_t*)(e + 16))[lane] = t.myrow;
    };
    for (int m = 0; m <= NT; ++m) publish(m);
    kchain_barrier();  // #0: the first NT + 1 maps are there
    int s = 0;
    while (!ended || s < ntiles + KB - 1) {
      kchain_barrier();
      publish(s + NT + 1);
      kchain_barrier();
      ++s;
    }
    return;
  }

  // ================= compute wave kb
  const int kb = wv;
  const bool last = kb == KB - 1;
  DocLanes dl;
  dl.row0 = 0; dl.len = 0; dl.flags = 2;
  if (last) dl = load_doc_lanes<MODE_RERANK>(p, qi, c_begin, ndoc, lane);  // (the reduction walks the docs, too)
  char* const wlds = lds + kb * (NT * SUB);
  char* const hop_in = hops + (kb - 1) * HOP;  // written by wave kb - 1 (kb > 0)
  char* const hop_out = hops + kb * HOP;       // read by wave kb + 1 (kb < KB - 1)
  const int rsw = r & 15;
  const int rdbase = r * BLKB;
  const char* const tok = (const char*)p.index + (uint32_t)kb * BLKB;  // this wave's block of every row

  // ---- this wave's block of the query -> registers, MFMA B layout: lane (n, hh) holds Q[n][128 kb + 16 i + 8 hh + j],
  //      j = 0..7, of k-group i in qp[piece][i] -- the operand the LDS-query kernel reads from its staged image
  u32x4 qp[NPQ][8];
  {
    int qlen = p.Lq;
    if (p.q_len) qlen = min(qlen, p.q_len[qi]);
    const bool live = q_token_live<MODE_RERANK>(p, qi, r, qlen);
    const int64_t e0 = ((int64_t)qi * p.Lq + (live ? r : 0)) * p.h + 128 * kb + 8 * hh;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float q[8];
      if (p.q_dtype == MAXSIM_F32) {
        const f32x4 v0 = *(const f32x4*)((const float*)p.Q + e0 + 16 * i), v1 = *(const f32x4*)((const float*)p.Q + e0 + 16 * i + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { q[j] = v0[j]; q[4 + j] = v1[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = load_q(p.Q, p.q_dtype, e0 + 16 * i + j);
      }
      uint16_t pc[2][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = live ? q[j] : 0.0f;
        if constexpr (DT == MAXSIM_F16) {
          const _Float16 hi = (_Float16)x;
          const _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
          __builtin_memcpy(&pc[0][j], &hi, 2);
          __builtin_memcpy(&pc[1][j], &lo, 2);
        } else {
          pc[0][j] = f32_to_bf16_rn(x);
          pc[1][j] = f32_to_bf16_rn(x - bf16_to_f32(pc[0][j]));
        }
      }
#pragma unroll
      for (int k = 0; k < NPQ; ++k)
#pragma unroll
        for (int y = 0; y < 4; ++y) qp[k][i][y] = (uint32_t)pc[k][2 * y] | ((uint32_t)pc[k][2 * y + 1] << 16);
    }
  }

  // ---- fetch side.  Request i of a sub-tile moves the row slots 4 i + lane / 16, 1 KiB, to ring slot + 1024 i; chunk
  //      position c of row slot m receives source chunk c ^ (m & 15) (issue_rows' layout).  A tile that lies inside one
  //      doc (kind 1) is 32 consecutive rows: scalar base + (i / 4) * 16 rows, and one of FOUR loop-invariant lane offsets
  const int ds0 = lane >> 4, dch = lane & 15;
  uint32_t lane_off[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) lane_off[j] = (uint32_t)(4 * j + ds0) * rowbytes + 16u * (uint32_t)(dch ^ ((4 * j + ds0) & 15));
  struct MapHead { int kind; uint32_t base0, base1; int split; };
  auto read_head = [&](int m) __attribute__((always_inline)) -> MapHead {
    const u32x4 hd = *(const u32x4*)(maps + (m & (KCHAIN_MAPD - 1)) * KCHAIN_MAPB);  // (one address for all lanes)
    MapHead mh;
    mh.kind = uni((int)hd[0]);
    mh.base0 = (uint32_t)uni((int)hd[1]);
    mh.base1 = (uint32_t)uni((int)hd[2]);
    mh.split = uni((int)hd[3]);
    return mh;
  };
  auto issue_one = [&](const MapHead& mh, int m, int i, char* l) __attribute__((always_inline)) {
    if (KCHAIN_ABLATE == 1 || KCHAIN_ABLATE == 5) return;
    if (mh.kind == 1) {
      const char* const base = tok + (uint64_t)(mh.base0 + 16u * (uint32_t)(i >> 2)) * rowbytes;
      __builtin_amdgcn_global_load_lds(GPTR(base + lane_off[i & 3]), LPTR(l + i * 1024), 16, 0, CPOL_STREAM);
    } else {
      const int slot = RPD * i + ds0;
      uint32_t row;
      if (mh.kind == 2) row = (slot < mh.split ? mh.base0 : mh.base1) + (uint32_t)slot;
      else row = ((const uint32_t*)(maps + (m & (KCHAIN_MAPD - 1)) * KCHAIN_MAPB + 16))[slot];
      const char* g = tok + (uint64_t)row * rowbytes + 16u * (uint32_t)(dch ^ (slot & 15));
      __builtin_amdgcn_global_load_lds(GPTR(g), LPTR(l + i * 1024), 16, 0, CPOL_STREAM);
    }
  };

  kchain_barrier();  // #0: the first NT + 1 maps are there
  int nissued = 0, nconsumed = 0;
  bool prev_issued = false;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const MapHead mh = read_head(j);
    const bool ok = mh.kind != 0;
    if (ok) {
#pragma unroll
      for (int i = 0; i < NDMA; ++i) issue_one(mh, j, i, wlds + j * SUB);
    }
    nissued += ok ? 1 : 0;
    prev_issued = ok;
  }

  Cursor C;
  C.init(dl, last ? ndoc : 0);
  MultiReducer<1, false, false> red;
  red.init();
  int32_t* argq[1] = {nullptr};
  // steps before this wave's first tile arrives from its predecessor: barriers only
  for (int s = 0; s < kb; ++s) {
    kchain_barrier();
    kchain_barrier();
  }
  int buf = 0;
#ifdef MAXSIM_STAMP  // timing builds (tools/probe_kchain_phases.py): shader-clock ticks per phase of a step, summed per wave
  uint64_t ph[7] = {0, 0, 0, 0, 0, 0, 0}, tk = __builtin_amdgcn_s_memtime();
#define KCHAIN_PHASE(i) do { const uint64_t n_ = __builtin_amdgcn_s_memtime(); ph[i] += n_ - tk; tk = n_; } while (0)
#else
#define KCHAIN_PHASE(i)
#endif
  while (nconsumed < nissued) {
    KCHAIN_PHASE(6);
    // ---- (1) the tile's accumulators so far (blocks 0 .. kb - 1), handed over in the previous step
    f32x16 acc0 = (f32x16)(0.0f), acc1 = (f32x16)(0.0f);
    if (kb > 0 && KCHAIN_ABLATE != 3) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(hop_in + g * 1024 + lane * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc0[4 * g + t] = v[t];
        if constexpr (NPQ == 2) {
          const f32x4 w = *(const f32x4*)(hop_in + 4096 + g * 1024 + lane * 16);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc1[4 * g + t] = w[t];
        }
      }
    }
    kchain_barrier();  // every wave has taken its input: the slots may be rewritten
    KCHAIN_PHASE(0);
    // ---- (2) this wave's block of the tile; the sub-tile NT tiles ahead is requested between its MFMAs
    const int mnext = nconsumed + NT;
    const MapHead mh = read_head(mnext);
    const bool more = mh.kind != 0;
    if (prev_issued) wait_vmcnt<NDMA * (NT - 1)>(); else wait_vmcnt<0>();
    KCHAIN_PHASE(1);
    char* const slot = wlds + buf * SUB;
    const char* tl = slot + rdbase;
    u32x4 a[NRD];
#pragma unroll
    for (int i = 0; i < NRD; ++i) a[i] = *(const u32x4*)(tl + 16 * ((2 * i + hh) ^ rsw));
    wait_lgkmcnt0();  // operands are in registers: the ring slot may be refilled
    KCHAIN_PHASE(2);
#pragma unroll
    for (int i = 0; i < NRD; ++i) {
      if (KCHAIN_ABLATE == 2) {
        acc0[i] += __uint_as_float(a[i][0] ^ qp[0][i][0]);
      } else if constexpr (DT == MAXSIM_F16) {
        const f16x8 av = __builtin_bit_cast(f16x8, a[i]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, qp[0][i]), acc0, 0, 0, 0);
        if constexpr (NPQ == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(f16x8, qp[1][i]), acc1, 0, 0, 0);
      } else {
        const bf16x8 av = __builtin_bit_cast(bf16x8, a[i]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, qp[0][i]), acc0, 0, 0, 0);
        if constexpr (NPQ == 2) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, qp[1][i]), acc1, 0, 0, 0);
      }
      if (more) issue_one(mh, mnext, i, slot);
    }
    nissued += more ? 1 : 0;
    prev_issued = more;
    buf = (buf + 1 == NT) ? 0 : buf + 1;
    KCHAIN_PHASE(3);
    // ---- (3) on to the next wave, or -- last block -- the similarities are complete
    if (!last) {
      if (KCHAIN_ABLATE != 3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *(f32x4*)(hop_out + g * 1024 + lane * 16) = f32x4{acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]};
          if constexpr (NPQ == 2)
            *(f32x4*)(hop_out + 4096 + g * 1024 + lane * 16) = f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
        }
      }
    } else {
      float sv[1][16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        if constexpr (NPQ == 2)
          sv[0][v] = (DT == MAXSIM_F16) ? fmaf(acc1[v], 1.0f / 2048.0f, acc0[v]) : (acc0[v] + acc1[v]);
        else
          sv[0][v] = acc0[v];
      }
      if (KCHAIN_ABLATE != 4 && KCHAIN_ABLATE != 5) red.reduce_tile(sv, C, dl, lane, argq, p.Lq);
      else red.rmax[0] = fmaxf(red.rmax[0], sv[0][nconsumed & 15]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    KCHAIN_PHASE(4);
    kchain_barrier();  // the hand-over is written: the next step may read it
    KCHAIN_PHASE(5);
    ++nconsumed;
  }
#ifdef MAXSIM_STAMP
  if (p.d_mask && lane == 0) {
    uint64_t* const sb = (uint64_t*)p.d_mask + ((int64_t)blockIdx.x * 8 + kb) * 8;
    for (int i = 0; i < 7; ++i) sb[i] = ph[i];
    sb[7] = (uint64_t)nconsumed;
  }
#endif
  // steps in which only the waves behind this one still work
  for (int s = kb; s < KB - 1; ++s) {
    kchain_barrier();
    kchain_barrier();
  }
  if (last) {
    red.drain(C, dl, lane, argq, p.Lq);
    if (lane < red.jdoc) p.scores[(int64_t)qi * p.ncand + c_begin + lane] = red.myscore[0];
  }
}

}  // namespace maxsim
