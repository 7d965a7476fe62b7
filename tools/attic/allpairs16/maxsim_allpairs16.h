// maxsim_allpairs16.h -- the all-pairs (training-form) MaxSim kernel of maxsim_allpairs.h on v_mfma_f32_16x16x32.
//
// Same blocking, same K-sliced LDS image, same LDS-DMA fetch side, same tile walk and the same results as
// k_maxsim_allpairs (read that header first); what differs is the matrix instruction and everything that hangs on its
// operand / accumulator layout.  Why: the forward is matrix-bound and the chip runs it on its power limit -- the
// 16x16x32 shape holds a higher clock than 32x32x16 at the same flop rate (MI355X_MICROARCH.md, DVFS give-back item 7;
// here: the timing experiment -DAP_MFMA16=1 of maxsim_allpairs.h, 2.48 -> 2.31 ms at the reference's training step).
//   wave       2R row blocks x 2QB column blocks of 16x16: acc[ia][ib] (4 registers each, 16 R QB in all, as before);
//              lane (n16 = lane & 15, kq = lane >> 4) holds rows 16 ia + 4 kq + v (v = 0..3) of column (ib, n16)
//   K          a slice (64 dims) is TWO k-steps of 32 dims; a fragment is ONE ds_read_b128: row 16 blk + n16, chunk
//              4 kt + kq at position chunk ^ ((row >> 1) & 7) -- the image of maxsim_allpairs.h is conflict-free for
//              these reads too (checked lane group by lane group)
//   registers  the B fragments of a k-step stay resident (2QB), double-buffered by k-step; the A fragments stream through
//              three register sets, two 16-row blocks ahead
//   loop       per k-step: for ia: [MFMA (ia, 0); read A two blocks ahead; MFMA (ia, 1); reads of the NEXT k-step's B
//              fragments, as early in the k-step as they fit; ...]; counted s_waitcnt lgkmcnt before a block's first MFMA
//              (LDS reads return in order; a scalar load in flight only makes the wait longer; ap16_wait).  The first
//              k-step of a slice starts behind the slice's barrier with its 2QB + 2 reads exposed, as in the 32x32x16 kernel.
//   epilogue   a lane scans TWO query tokens per query (n16 and 16 + n16) over its 8 R rows each, in increasing row
//              order; the four row quarters meet through v_permlane16_swap + v_permlane32_swap (first maximal row wins)
#pragma once
#include "maxsim_allpairs.h"

namespace maxsim {

// MFMA slots of k-step 0 behind which a wave issues its LDS-DMA instructions of the next slice: the first-half waves (doc
// rows, 4 R instructions) one every AP16_DMA_ASTEP MFMAs from slot AP16_DMA_A0; the second-half waves (query rows, 2 QB
// instructions) behind them, from slot A0 + 4 R ASTEP on (the two waves of a SIMD -- w, w + 4 -- fetch at different times:
// an LDS-DMA instruction holds its wave for 60-180 cycles, and matrix beside memory is what two waves of a SIMD overlap)
#ifndef AP16_DMA_A0
#define AP16_DMA_A0 3
#endif
#ifndef AP16_DMA_ASTEP
#define AP16_DMA_ASTEP 2
#endif
#ifndef AP16_PRIO
#define AP16_PRIO 0
#endif
// (timing experiments, WRONG results: -DAP16_NOWAIT=1 drops the counted waits in front of the blocks, -DAP16_NOREAD=1 the
//  fragment reads inside the k-steps)
#ifndef AP16_NOWAIT
#define AP16_NOWAIT 0
#endif
#ifndef AP16_NOREAD
#define AP16_NOREAD 0
#endif
#ifndef AP16_SHAREB
#define AP16_SHAREB 0
#endif

// The read schedule of a slice as compile-time numbers.  Slot S = kt * NM + ia * NB + ib counts the slice's MFMAs; block
// g = kt * NA + ia.  Reads are SPREAD: with a 16-cycle MFMA every gap that carries a ds_read_b128 from each of the CU's
// waves saturates the LDS array, and reads in consecutive gaps (bursts) slow the MFMA stream itself -- measured: the loop
// without its in-loop reads 2.11 ms, with them in bursts of three 2.30 ms, the 32x32x16 kernel's one-per-32-cycles spacing
// 2.16 ms (timing builds, no arg-max).
//   A(g + 2)      behind MFMA 0 of block g (two blocks ahead, three register sets)
//   next B set    k-step 0 only: read j behind MFMA NB / 2 of block j (j < NA - 1), the rest behind the LAST MFMA of blocks
//                 0, 1, ... -- all of them before the last block, so that the k-step boundary does not wait for a fresh read
//   wait          in front of block g: s_waitcnt lgkmcnt(n), n = the reads issued AFTER the youngest one block g needs
//                 (A(g); at the k-step boundary also the youngest B read) -- LDS reads return in order
constexpr int ap16_pos_a(int g, int NB) { return g < 2 ? -1 : (g - 2) * NB; }            // slot behind which A(g) is read (-1: slice start)
constexpr int ap16_pos_b(int j, int NA, int NB) {                                         // ... the next k-step's B(j)
  return j < NA - 1 ? j * NB + NB / 2 : (j - (NA - 1)) * NB + NB - 1;
}
constexpr int ap16_wait(int g, int NA, int NB) {
  if (g == 0) return 1;                                   // behind the slice's barrier: B set, A(0), A(1) -- A(1) may be in flight
  int need = ap16_pos_a(g, NB);                           // the youngest read block g needs
  if (g == NA)
    for (int j = 0; j < NB; ++j) need = ap16_pos_b(j, NA, NB) > need ? ap16_pos_b(j, NA, NB) : need;
  const int start = g * NB;                               // block g's first slot: reads behind slots < start are issued
  int younger = 0;
  for (int h = 2; h < 2 * NA; ++h) {                      // A reads (A(1) of the slice start counts as position -1, after A(0))
    const int p = ap16_pos_a(h, NB);
    younger += (p > need && p < start) ? 1 : 0;
  }
  if (g == 1) younger += 0;                               // (A(1) itself is the youngest of the start-up reads)
  for (int j = 0; j < NB; ++j) {
    const int p = ap16_pos_b(j, NA, NB);
    younger += (p > need && p < start) ? 1 : 0;
  }
  return younger;
}

template <int DT, int R, int QB, bool AM>
__global__ void __launch_bounds__(512) k_maxsim_allpairs16(const AllPairsArgs a) {
  static_assert(DT == MAXSIM_F16 || DT == MAXSIM_BF16, "16-bit operands");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NQ = 2 * QB;                     // queries of a tile
  constexpr int TM = 128 * R, TN = 32 * NQ;      // tile rows (doc tokens) / columns (query tokens)
  constexpr int ROWS = TM + TN;                  // rows of a K slice image
  constexpr int STAGE = ROWS * 128;              // bytes: 64 dims x 2 B per row
  constexpr int KT = 2;                          // k-steps (32 dims) of a slice
  constexpr int NAI = TM / 8;                    // LDS-DMA instructions (8 rows each) < NAI move doc rows, the others query rows
  constexpr int NDA = NAI / 4;                   // ... per slice of a first-half wave (the doc rows: 4 R)
  constexpr int NDB = (TN / 8) / 4;              // ... of a second-half wave (the query rows: 2 QB)
  constexpr int NA = 2 * R, NB = 2 * QB;         // 16-row A blocks / 16-column B blocks of a wave
  constexpr int NM = NA * NB;                    // MFMAs of a k-step
  static_assert(NB <= 2 * (NA - 1) && NB >= 2, "the next B set is read in k-step 0's blocks 0 .. NA - 2, at most two reads per block");
  constexpr int DMA_B0 = AP16_DMA_A0 + NDA * AP16_DMA_ASTEP;
  constexpr int DMA_BSTEP = (NM - DMA_B0) / NDB >= 1 ? (NM - DMA_B0) / NDB : 1;
  static_assert(AP16_DMA_A0 + (NDA - 1) * AP16_DMA_ASTEP < NM && DMA_B0 + (NDB - 1) * DMA_BSTEP < NM,
                "the LDS-DMA instructions of a slice must fit into k-step 0's MFMA slots");
  float* const ex_v = (float*)(lds + 2 * STAGE);        // [NQ][4 row-waves][32]: per-wave (max) ...
  int* const ex_i = (int*)(ex_v + NQ * 4 * 32);         // ... and (first index)
  float* const dm_lds = (float*)(ex_i + NQ * 4 * 32);   // [TM]: d_mask row of the tile's doc
  float* const qm_lds = dm_lds + TM;                    // [NQ * 32]: q_mask rows of the tile's queries
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const lds_ptr_t lds3 = (lds_ptr_t)LPTR(lds), dm_dst = (lds_ptr_t)LPTR(dm_lds), qm_dst = (lds_ptr_t)LPTR(qm_lds);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6), wm = wave & 3, wn = wave >> 2;
  const bool first_half = wn == 0;  // waves 0-3 fetch the doc rows, waves 4-7 (their SIMD partners) the query rows
  const int r = lane & 31;
  const int n16 = lane & 15, kq = lane >> 4;
  const int nslices = a.h >> 6;
  const uint32_t rowb = (uint32_t)a.h * 2;
  const int nqb = (a.nq + NQ - 1) / NQ;

  // ---- this workgroup's tiles (as k_maxsim_allpairs): XCD x = id % 8 owns docs x, x + 8, ...
  const int x = blockIdx.x & 7, l = blockIdx.x >> 3, nl = max(1, (int)gridDim.x >> 3);
  const int ndx = (a.nd - x + 7) >> 3;
  const int ntx = ndx * nqb;
  auto tile_doc = [&](int u) { return x + 8 * (u / nqb); };
  auto tile_q0 = [&](int u) { return NQ * (u % nqb); };
  const int my_tiles = l < ntx ? (ntx - l + nl - 1) / nl : 0;
  const int total = my_tiles * nslices;

  // ---- fetch side: identical to k_maxsim_allpairs (same image)
  const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(first_half ? a.D : a.Q), 0,
      (int)(uint32_t)((uint64_t)(first_half ? (int64_t)a.nd * a.Ld : (int64_t)a.nq * a.Lq) * rowb), 0x00020000);
  const uint32_t lane_off = (uint32_t)((first_half ? lane >> 3 : 8 * wm + (lane >> 3)) * rowb) + (first_half ? 8u * wm * rowb : 0u) +
                            (uint32_t)((((lane & 7) ^ ((4 * (wm & 1) + (lane >> 4)) & 7))) << 4);
  const uint32_t j_stride = first_half ? 32u * rowb : (uint32_t)a.Lq * rowb;
  const uint32_t dst_off = (uint32_t)(((first_half ? 0 : NAI) + wm) * 1024);
  const bool masked = a.mask_dtype != MAXSIM_MASK_NONE;
  for (int i = tid; i < TM + NQ * 32; i += 512) {
    const bool pad = i >= a.Ld && i < TM;
    if (pad || !masked) dm_lds[i] = pad ? __builtin_nanf("") : 1.0f;
  }
  auto issue_masks = [&](int ti) __attribute__((always_inline)) {  // waves 6 and 7: the mask rows of tile ti
    const int u = l + ti * nl;
    const int d = tile_doc(u), q0 = tile_q0(u);
    if (wave == 6) {
#pragma unroll
      for (int j = 0; j < TM / 64; ++j) {
        const int row = j * 64 + lane;
        if (row < a.Ld)
          __builtin_amdgcn_global_load_lds(GPTR((const float*)a.d_mask + (int64_t)d * a.Ld + row), (__attribute__((address_space(3))) void*)(dm_dst + j * 256), 4, 0, 0);
      }
    } else if (wave == 7) {
#pragma unroll
      for (int j = 0; j < NQ / 2; ++j) {
        const int slot = 2 * j + (lane >> 5);
        const int qq = min(q0 + slot, a.nq - 1), t = min(lane & 31, a.Lq - 1);
        __builtin_amdgcn_global_load_lds(GPTR((const float*)a.q_mask + (int64_t)qq * a.Lq + t), (__attribute__((address_space(3))) void*)(qm_dst + j * 256), 4, 0, 0);
      }
    }
  };
  int is_ti = 0, is_s = 0;
  uint32_t is_base = 0;
  auto issue_setup = [&]() __attribute__((always_inline)) {
    if (is_s == 0) {
      const int u = l + is_ti * nl;
      is_base = (uint32_t)(first_half ? tile_doc(u) * a.Ld : tile_q0(u) * a.Lq) * rowb;
    }
  };
  auto issue_done = [&]() __attribute__((always_inline)) {
    if (++is_s == nslices) { is_s = 0; ++is_ti; }
  };
#define AP_DMA(st, j)                                                                                                 \
  do {                                                                                                                \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (__attribute__((address_space(3))) void*)(lds3 + (st) * STAGE + dst_off + (j) * 4096), 16, \
                                             (int)(lane_off + (is_base + (uint32_t)is_s * 128u + (uint32_t)(j) * j_stride)), 0, 0, 0); \
  } while (0)

  // ---- compute side.  Fragment of 16-row block blk, k-step kt: row base + 16 blk + n16, chunk 4 kt + kq at position
  //      chunk ^ ((row >> 1) & 7) = ((kq ^ swz) ^ 4 kt): the k-step is an XOR of bit 6 of the byte address
  const uint32_t swz = (uint32_t)(n16 >> 1);
  const uint32_t fa0 = (uint32_t)(((wm * R) * 32 + n16) * 128) + ((kq ^ swz) << 4);
  const uint32_t fb0 = (uint32_t)((TM + (wn * QB) * 32 + n16) * 128) + ((kq ^ swz) << 4);
  f32x4 acc[NA][NB];
#pragma unroll
  for (int ia = 0; ia < NA; ++ia)
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) acc[ia][ib] = (f32x4)(0.0f);
  u32x4 fa[3], fb[2][NB];  // A: block g in set g % 3, two blocks ahead in flight; B: the k-step's set, double-buffered by k-step
#define AP_READ_A(slot, blk, addr) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[slot]) : "v"(addr), "n"((blk) * 2048))
#define AP_READ_B(set, blk, addr) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][blk]) : "v"(addr), "n"((blk) * 2048))
#define AP_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
  // (the MFMAs behind a wait must not be hoisted above it: tie the operand registers to the wait's position)
#define AP_PIN_A(slot) asm volatile("" : "+v"(fa[slot]))
#define AP_PIN_B(set) _Pragma("unroll") for (int pb_ = 0; pb_ < NB; ++pb_) asm volatile("" : "+v"(fb[set][pb_]))
#define AP_MM16(slot, set, ia, ib)                                                                                  \
  do {                                                                                                                \
    if constexpr (DT == MAXSIM_F16)                                                                                   \
      acc[ia][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[slot]), __builtin_bit_cast(f16x8, fb[set][ib]), acc[ia][ib], 0, 0, 0); \
    else                                                                                                              \
      acc[ia][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[slot]), __builtin_bit_cast(bf16x8, fb[set][ib]), acc[ia][ib], 0, 0, 0); \
  } while (0)

  // ---- prologue: tile 0's mask rows and slice 0 on their way
  if (total > 0) {
    if (masked) issue_masks(0);
    issue_setup();
    if (first_half) {
#pragma unroll
      for (int j = 0; j < NDA; ++j) AP_DMA(0, j);
    } else {
#pragma unroll
      for (int j = 0; j < NDB; ++j) AP_DMA(0, j);
    }
    issue_done();
  }

#if AP16_PRIO
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);  // (experiment: static priority for the later-dispatched half, MI355X_MICROARCH.md)
#endif
  int cs = 0, cti = 0;  // (slice in tile, tile) being computed
  for (int g = 0; g < total; ++g) {
    const int st = g & 1;
    wait_vmcnt<0>();
    wg_barrier();
    const bool first = cs == 0;
    const bool do_issue = g + 1 < total;
    if (do_issue) issue_setup();
    if (masked && first && g > 0) issue_masks(cti);  // (tile 0's mask rows are issued in the prologue)
    const uint32_t aa = fa0 + (uint32_t)(st * STAGE), ab = fb0 + (uint32_t)(st * STAGE);
    // k-step 0's operands: the B set and the first two A blocks (exposed: the slice is known to be there only now)
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) AP_READ_B(0, ib, ab);
    AP_READ_A(0, 0, aa);
    AP_READ_A(1, 1, aa);
    // One k-step: MFMA (ia, ib) for ia, ib in order, reads as scheduled above; this wave's LDS-DMA instructions of the next
    // slice in k-step 0 (slots: see AP16_DMA_A0).
#define AP16_KSTEP(kt)                                                                                                \
  {                                                                                                                   \
    const uint32_t bnx_ = ab ^ ((uint32_t)((kt) + 1) << 6);                                                           \
    _Pragma("unroll") for (int ia = 0; ia < NA; ++ia) {                                                               \
      const int g_ = (kt) * NA + ia;                                                                                  \
      if (!AP16_NOWAIT || g_ == 0) AP_WAIT_LGKM(ap16_wait(g_, NA, NB));                                               \
      if (ia == 0) { AP_PIN_B((kt) & 1); }                                                                            \
      AP_PIN_A(g_ % 3);                                                                                               \
      _Pragma("unroll") for (int ib = 0; ib < NB; ++ib) {                                                             \
        if (AP16_SHAREB) { /* timing experiment, WRONG results: pairs of MFMAs with the same A AND B registers */     \
          if constexpr (DT == MAXSIM_F16) acc[ia][ib] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[g_ % 3]), __builtin_bit_cast(f16x8, fb[(kt) & 1][ib & ~1]), acc[ia][ib], 0, 0, 0); \
          else acc[ia][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[g_ % 3]), __builtin_bit_cast(bf16x8, fb[(kt) & 1][ib & ~1]), acc[ia][ib], 0, 0, 0); \
        } else                                                                                                        \
        AP_MM16(g_ % 3, (kt) & 1, ia, ib);                                                                            \
        if (!AP16_NOREAD && ib == 0 && g_ + 2 < KT * NA) {                                                            \
          const int g2_ = g_ + 2; /* A two blocks ahead: k-step g2 / NA, block g2 % NA */                             \
          const uint32_t a2_ = aa ^ ((uint32_t)(g2_ / NA) << 6);                                                      \
          AP_READ_A(g2_ % 3, g2_ % NA, a2_);                                                                          \
        }                                                                                                             \
        if (!AP16_NOREAD && (kt) + 1 < KT) {                                                                          \
          _Pragma("unroll") for (int j = 0; j < NB; ++j)                                                              \
            if (ap16_pos_b(j, NA, NB) == ia * NB + ib) AP_READ_B(((kt) + 1) & 1, j, bnx_);                            \
        }                                                                                                             \
        if ((kt) == 0 && do_issue) {                                                                                  \
          const int m_ = ia * NB + ib;                                                                                \
          if (first_half && m_ >= AP16_DMA_A0 && (m_ - AP16_DMA_A0) % AP16_DMA_ASTEP == 0 && (m_ - AP16_DMA_A0) / AP16_DMA_ASTEP < NDA) \
            AP_DMA(st ^ 1, (m_ - AP16_DMA_A0) / AP16_DMA_ASTEP < NDA ? (m_ - AP16_DMA_A0) / AP16_DMA_ASTEP : 0);      \
          if (!first_half && m_ >= DMA_B0 && (m_ - DMA_B0) % DMA_BSTEP == 0 && (m_ - DMA_B0) / DMA_BSTEP < NDB)       \
            AP_DMA(st ^ 1, (m_ - DMA_B0) / DMA_BSTEP < NDB ? (m_ - DMA_B0) / DMA_BSTEP : 0);                          \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
  }
    AP16_KSTEP(0)
    AP16_KSTEP(1)
#undef AP16_KSTEP
    if (do_issue) issue_done();
    if (++cs < nslices) continue;

    // ---- epilogue of tile cti: similarities complete ---------------------------------------------------------------
    const int u = l + cti * nl;
    const int d = tile_doc(u), q0 = tile_q0(u);
    const float* const dm = dm_lds;
    const float* const qmp = qm_lds;
    // the d_mask words of this wave's 16-row blocks, block by block (wave-uniform): all 1, all 0, all NaN (tile padding
    // past Ld: no candidate), or mixed
    enum { BLK_MIXED = 0, BLK_ONES = 1, BLK_ZEROS = 2, BLK_NONE = 3 };
    int kind[NA];
#pragma unroll
    for (int ia = 0; ia < NA; ++ia) {
      const float w = dm[(wm * R) * 32 + 16 * ia + n16];
      const bool ones = __builtin_amdgcn_ballot_w64(w == 1.0f) == ~0ull, zeros = __builtin_amdgcn_ballot_w64(w == 0.0f) == ~0ull;
      const bool nans = __builtin_amdgcn_ballot_w64(w != w) == ~0ull;
      kind[ia] = ones ? BLK_ONES : zeros ? BLK_ZEROS : nans ? BLK_NONE : BLK_MIXED;
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {  // one query at a time; this lane's two tokens of it: n16 (ib = 2 q) and 16 + n16 (ib = 2 q + 1)
      const int qq = q0 + wn * QB + q;
      float res[2];
      int resi[2];
#pragma unroll
      for (int tk = 0; tk < 2; ++tk) {
        const int tok = 16 * tk + n16;
        // tokens past Lq and query slots past nq: weight 0 (similarity 0, nothing written)
        const float qm = (qq < a.nq && tok < a.Lq) ? qmp[(wn * QB + q) * 32 + tok] : 0.0f;
        // a non-negative q_mask commutes with the max (see k_maxsim_allpairs); a negative weight anywhere in the wave takes
        // the multiplication back into the mask words of every block
        const bool premul = __builtin_amdgcn_ballot_w64(qm < 0.0f) != 0;
        float best = NEG_INF;
        int bidx = 0;  // position ia * 4 + v of the winner among this lane's values (compile-time numbers)
#define AP_PICK(sim_, pos_)                                                                                           \
  do {                                                                                                                \
    if constexpr (AM) {                                                                                               \
      const bool better = (sim_) > best; /* strict >: the first maximal token wins (torch.max); a NaN never does */    \
      best = better ? (sim_) : best;                                                                                  \
      bidx = better ? (pos_) : bidx;                                                                                  \
    } else {                                                                                                          \
      best = __builtin_fmaxf(best, (sim_));                                                                           \
    }                                                                                                                 \
  } while (0)
#define AP_SCAN(PREMUL)                                                                                               \
  _Pragma("unroll") for (int ia = 0; ia < NA; ++ia) {                                                                 \
    if (!PREMUL && kind[ia] == BLK_ONES) {                                                                            \
      _Pragma("unroll") for (int v = 0; v < 4; ++v) AP_PICK(acc[ia][2 * q + tk][v], ia * 4 + v);                      \
    } else if (!PREMUL && kind[ia] == BLK_ZEROS) {                                                                    \
      AP_PICK(0.0f, ia * 4);                                                                                          \
    } else if (kind[ia] != BLK_NONE) {                                                                                \
      int rowbase = (wm * R) * 32 + 16 * ia + 4 * kq;                                                                 \
      asm volatile("" : "+v"(rowbase));                                                                               \
      f32x4 w = *(const f32x4*)(dm + rowbase); /* d_mask of this lane's 4 rows of the block; NaN past Ld */           \
      if (PREMUL) w *= qm;                                                                                            \
      _Pragma("unroll") for (int v = 0; v < 4; ++v) AP_PICK(acc[ia][2 * q + tk][v] * w[v], ia * 4 + v);               \
    }                                                                                                                 \
  }
        if (premul) { AP_SCAN(true) } else { AP_SCAN(false) }
#undef AP_SCAN
#undef AP_PICK
        // (Q q_mask) . (D d_mask), BaseModel.py:41-43.  q_mask = 0 makes every similarity of the token 0: the first row wins
        const bool qzero = !premul && qm == 0.0f;
        if (!premul) best = qzero ? 0.0f : best * qm;
        if (AM) bidx = qzero ? (wm == 0 && kq == 0 ? 0 : 0x7fffffff) : (wm * R) * 32 + 16 * (bidx >> 2) + 4 * kq + (bidx & 3);
        // the four row quarters (lanes n16 + 16 kq) hold interleaved rows of the same query token
        float v2 = best;
        int i2 = bidx;
#pragma unroll
        for (int step = 0; step < 2; ++step) {
          float va, vb;
          int ia2 = 0, ib2 = 0;
          if (step == 0) {
            const auto sv = __builtin_amdgcn_permlane16_swap(__float_as_uint(v2), __float_as_uint(v2), false, false);
            va = __uint_as_float(sv[0]); vb = __uint_as_float(sv[1]);
            if constexpr (AM) {
              const auto si = __builtin_amdgcn_permlane16_swap((uint32_t)i2, (uint32_t)i2, false, false);
              ia2 = (int)si[0]; ib2 = (int)si[1];
            }
          } else {
            const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(v2), __float_as_uint(v2), false, false);
            va = __uint_as_float(sv[0]); vb = __uint_as_float(sv[1]);
            if constexpr (AM) {
              const auto si = __builtin_amdgcn_permlane32_swap((uint32_t)i2, (uint32_t)i2, false, false);
              ia2 = (int)si[0]; ib2 = (int)si[1];
            }
          }
          if constexpr (AM) {
            const bool take_b = (vb > va) || (vb == va && ib2 < ia2);
            v2 = take_b ? vb : va;
            i2 = take_b ? ib2 : ia2;
          } else {
            v2 = fmaxf(va, vb);
          }
        }
        res[tk] = v2;
        resi[tk] = i2;
      }
      if (lane < 32) {  // lane t writes token t: lanes 0..15 their first token, lanes 16..31 their second (16 + n16 = lane)
        ex_v[((wn * QB + q) * 4 + wm) * 32 + lane] = lane < 16 ? res[0] : res[1];
        if (AM) ex_i[((wn * QB + q) * 4 + wm) * 32 + lane] = lane < 16 ? resi[0] : resi[1];
      }
    }
    lds_barrier();
    // the row-waves' results meet: wave w finishes query slot w of the tile (NQ <= 8 slots)
    if (wave < NQ) {
      const int qq = q0 + wave;
      float best = NEG_INF;
      int bidx = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {  // increasing row ranges: strict > keeps the first maximal token
        const float v = ex_v[(wave * 4 + w) * 32 + r];
        const bool better = v > best;
        best = better ? v : best;
        if (AM) bidx = better ? ex_i[(wave * 4 + w) * 32 + r] : bidx;
      }
      if (qq < a.nq) {
        if (AM && lane < a.Lq) a.argmax[((int64_t)qq * a.nd + d) * a.Lq + lane] = bidx;
        float v = best;  // both lane halves hold the 32 tokens: sum one half with a fixed DPP tree
        v += ap_dpp<0xB1>(v);
        v += ap_dpp<0x4E>(v);
        v += ap_dpp<0x141>(v);
        v += ap_dpp<0x140>(v);
        const float sc = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 0)) +
                         __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(v), 16));
        if (lane == 0) a.scores[(int64_t)qq * a.nd + d] = sc;
      }
    }
#pragma unroll
    for (int ia = 0; ia < NA; ++ia)
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) acc[ia][ib] = (f32x4)(0.0f);
    cs = 0;
    ++cti;
  }
#undef AP_DMA
#undef AP_READ_A
#undef AP_READ_B
#undef AP_WAIT_LGKM
#undef AP_PIN_A
#undef AP_PIN_B
#undef AP_MM16
}

}  // namespace maxsim
