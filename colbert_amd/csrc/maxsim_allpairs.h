// maxsim_allpairs.h -- the all-pairs (training-form) MaxSim as a register-blocked, K-sliced GEMM with a fused
// max / arg-max / sum epilogue: 16-bit Q and D of any width h that is a multiple of 64 (the reference trains at dim 768,
// proj_conf/dense.yaml:8), Lq <= 32 query tokens, Ld <= 384 doc tokens (doc_maxlen, dense.yaml:7).
//
// Reference: BaseModel.score (colbert/modeling/BaseModel.py:39-46) as ColbertModel.forward calls it on the gathered batch
// (colbert/modeling/colbert_model.py:87-90): simmat[q,d,m,n] = <Q[q,m] q_mask[q,m], D[d,n] d_mask[d,n]>, max over n
// (torch.max: the FIRST maximal index is what autograd routes the gradient through), sum over m.  At the reference's
// step (Q 272 x 32 x 768, D 544 x 384 x 768) that is a 208 896 x 8 704 x 768 GEMM = 2.79 PFLOP whose result never has
// to exist: only 148 k scores and 4.7 M arg-max indices leave the kernel.
//
// The streaming kernel (maxsim_stream_bigh.h) computes this with one wave per 32-row doc tile and whole query images
// in LDS: 2 KB of LDS traffic per MFMA, measured 22 % of the bf16 peak.  Here the blocking is a GEMM's:
//   workgroup  8 waves = 4 (doc rows) x 2 (queries); tile = ONE doc (up to 128 R rows, R = 1..3) x 2 QB queries
//   wave       R row blocks x QB queries of 32x32x16 MFMAs: R + QB fragment reads (1 KB each) per R QB MFMAs, 16 R QB
//              accumulator registers.  (R, QB) = (1, 4), (2, 4), (3, 3): with 8 waves a wave has 256 registers, and
//              3 x 4 blocks (192 accumulators) spilled
//   K          sliced by 64 dims: a slice is (128 R + 64 QB) rows x 128 B -- whole cache lines: an LDS-DMA instruction
//              (buffer_load_dwordx4 ... lds, 1 KB, no register staging) that takes 8 rows x 128 B costs the CU's memory
//              pipeline 12 cycles, one that takes 16 rows x 64 B (a 32-dim slice) 18 (tools/micro/lds_dma_rate.hip) --
//              into a 2-stage ring: one slice being read, the next in flight for the four k-steps of the current one.
//              Addresses are a buffer descriptor over the whole tensor + ONE 32-bit lane offset + a wave-uniform term:
//              what lies past the end of the tensor reads as 0 (rows past a doc's Ld and query slots past nq need no
//              clamping: they are tile padding, weighted NaN / 0 in the epilogue)
//   loop       ONE bare s_barrier per slice (__syncthreads would be a fence: see wg_barrier); fragment reads are
//              ds_read_b128 by hand, double-buffered by k-step and placed between the MFMAs of the previous k-step; the
//              two waves of a SIMD (w, w + 4) issue their LDS-DMA instructions in DIFFERENT k-steps (an LDS-DMA
//              instruction holds its wave for 60-180 cycles; matrix beside memory is what two waves of a SIMD overlap)
//   tiles      workgroups are persistent (one per CU: ~152 KB of LDS) and walk (doc, query block) tiles; the ring runs
//              across tile boundaries.  XCD x takes the docs x, x + 8, ... and walks their query blocks in order: the ~32
//              workgroups of an XCD work on one or two docs at a time (the doc comes from that XCD's L2; the 13 MB of
//              queries from the Infinity Cache)
//   epilogue   per lane and query: running (max, first position) over its rows in increasing row order, lane halves
//              combined with v_permlane32_swap, the four row-waves through LDS; d_mask multiplies the finished
//              similarities block by block (skipped for blocks of all ones / all zeros / tile padding), a non-negative
//              q_mask the maximum (exact for 0/1 masks, the only ones training uses: tokenizers.py:36,57; a negative
//              weight takes the multiplication back into every similarity)
#pragma once
#include "maxsim_common.h"

namespace maxsim {

struct AllPairsArgs {
  const void* Q;       // [nq, Lq, h]
  const void* D;       // [nd, Ld, h]
  const void* q_mask;  // [nq, Lq] or NULL
  const void* d_mask;  // [nd, Ld] or NULL
  float* scores;       // [nq, nd]
  int32_t* argmax;     // [nq, nd, Lq] (AM)
  int mask_dtype, nq, nd, Lq, Ld, h;
};

template <int CTRL>
__device__ __forceinline__ float ap_dpp(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false));
}

// The bare hardware barrier.  __syncthreads() is a workgroup fence + barrier, and the fence waits for EVERY outstanding
// memory operation of the wave (s_waitcnt vmcnt(0)) at a point the compiler chooses; the loop orders its data by hand:
// this wave's part of the next slice has landed (vmcnt) / its LDS accesses are done (lgkmcnt), then the barrier.
__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ void lds_barrier() {  // this wave's LDS stores are done, then the barrier (no vmcnt wait)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// (timing experiments: -DAP_ABLATE_DMA=1 drops the loop's LDS-DMA instructions, -DAP_ABLATE_MFMA=1 its MFMAs; results
//  are then wrong, only the time is of interest)
#ifndef AP_ABLATE_DMA
#define AP_ABLATE_DMA 0
#endif
#ifndef AP_ABLATE_MFMA
#define AP_ABLATE_MFMA 0
#endif
// k-steps (0..3) in which the first-half waves (doc rows) and the second-half waves (query rows) issue their LDS-DMA
// instructions of the next slice: the first half in A0 and A0 + 1, the second half in B0 (R = 2: in 0 and 1, see the loop)
#ifndef AP_DMA_A0
#define AP_DMA_A0 0
#endif
#ifndef AP_DMA_B0
#define AP_DMA_B0 1
#endif
#ifndef AP_DMA_BSPLIT
#define AP_DMA_BSPLIT 0
#endif
// (timing experiment: -DAP_MFMA16=1 issues every 32x32x16 MFMA as two 16x16x32 on quarters of its accumulator: same flops,
//  operand reads and pipe cycles, WRONG results -- what the matrix shape alone does to the clock the chip holds)
#ifndef AP_MFMA16
#define AP_MFMA16 0
#endif

template <int DT, int R, int QB, bool AM>
__global__ void __launch_bounds__(512) k_maxsim_allpairs(const AllPairsArgs a) {
  static_assert(DT == MAXSIM_F16 || DT == MAXSIM_BF16, "16-bit operands");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NQ = 2 * QB;                     // queries of a tile
  constexpr int TM = 128 * R, TN = 32 * NQ;      // tile rows (doc tokens) / columns (query tokens)
  constexpr int ROWS = TM + TN;                  // rows of a K slice image
  constexpr int STAGE = ROWS * 128;              // bytes: 64 dims x 2 B per row
  constexpr int KS = 4;                          // k-steps (16 dims) of a slice
  constexpr int NAI = TM / 8;                    // LDS-DMA instructions (8 rows each) < NAI move doc rows, the others query rows
  constexpr int NDA = NAI / 4;                   // ... per slice of a first-half wave (the doc rows: 4 R)
  constexpr int NDB = (TN / 8) / 4;              // ... of a second-half wave (the query rows: 2 QB)
  constexpr int NM = R * QB, NF = R + QB;        // MFMAs / fragment reads of a k-step
  float* const ex_v = (float*)(lds + 2 * STAGE);        // [NQ][4 row-waves][32]: per-wave (max) ...
  int* const ex_i = (int*)(ex_v + NQ * 4 * 32);         // ... and (first index)
  float* const dm_lds = (float*)(ex_i + NQ * 4 * 32);   // [TM]: d_mask row of the tile's doc
  float* const qm_lds = dm_lds + TM;                    // [NQ * 32]: q_mask rows of the tile's queries
  // (LDS-space views for the DMA destinations, cast here in uniform control flow: the generic -> LDS cast inside a
  //  divergent branch trips a code-generation bug of this compiler)
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const lds_ptr_t lds3 = (lds_ptr_t)LPTR(lds), dm_dst = (lds_ptr_t)LPTR(dm_lds), qm_dst = (lds_ptr_t)LPTR(qm_lds);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6), wm = wave & 3, wn = wave >> 2;
  const bool first_half = wn == 0;  // waves 0-3 fetch the doc rows, waves 4-7 (their SIMD partners) the query rows
  const int r = lane & 31, hh = lane >> 5;
  const int nslices = a.h >> 6;
  const uint32_t rowb = (uint32_t)a.h * 2;
  const int nqb = (a.nq + NQ - 1) / NQ;

  // ---- this workgroup's tiles: XCD x = id % 8 owns docs x, x + 8, ...; its workgroups walk (doc, query block) in order
  const int x = blockIdx.x & 7, l = blockIdx.x >> 3, nl = max(1, (int)gridDim.x >> 3);
  const int ndx = (a.nd - x + 7) >> 3;
  const int ntx = ndx * nqb;  // tiles of this XCD
  auto tile_doc = [&](int u) { return x + 8 * (u / nqb); };
  auto tile_q0 = [&](int u) { return NQ * (u % nqb); };
  const int my_tiles = l < ntx ? (ntx - l + nl - 1) / nl : 0;
  const int total = my_tiles * nslices;  // slices this workgroup streams

  // ---- fetch side.  An LDS-DMA instruction moves 8 rows x 128 B of the slice image: lane i -> row + i / 8, 16-byte
  //      position i % 8, which receives source chunk (i % 8) ^ ((row >> 1) & 7): the four 16-lane groups of a fragment
  //      read (ds_read_b128: one chunk of 32 consecutive rows) then cover all 64 banks exactly once.  First-half wave wm
  //      moves the image rows 8 (wm + 4 j) .. + 7 (j < NDA), second-half wave wm the rows TM + 8 (wm + 4 j) .. (j < NDB):
  //      (row >> 1) & 7 = (4 (wm & 1) + lane / 16) & 7 for every j, so the lane offset is ONE register; the rest of
  //      the address -- tensor offset of the doc / query block, 32 j rows / query slot j, 128 B per slice -- is scalar.
  const __amdgpu_buffer_rsrc_t src = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(first_half ? a.D : a.Q), 0,
      (int)(uint32_t)((uint64_t)(first_half ? (int64_t)a.nd * a.Ld : (int64_t)a.nq * a.Lq) * rowb), 0x00020000);
  const uint32_t lane_off = (uint32_t)((first_half ? lane >> 3 : 8 * wm + (lane >> 3)) * rowb) + (first_half ? 8u * wm * rowb : 0u) +
                            (uint32_t)((((lane & 7) ^ ((4 * (wm & 1) + (lane >> 4)) & 7))) << 4);
  const uint32_t j_stride = first_half ? 32u * rowb : (uint32_t)a.Lq * rowb;  // doc rows 32 j / query slot j
  const uint32_t dst_off = (uint32_t)(((first_half ? 0 : NAI) + wm) * 1024);   // instruction j lands at + 4096 j
  // masks (float32 or none -- the launcher sends other mask types to the streaming kernel): the tile's mask rows come in
  // by LDS-DMA with the tile's first slice, like everything else (an ordinary load in this loop would be ordered against
  // the LDS-DMA traffic by the compiler).  Rows past Ld are tile padding: their mask word is NaN for the whole kernel,
  // so their products are NaN and never win a `>`; without masks every real row / token weighs 1.
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
        if (row < a.Ld)  // (lanes past Ld stay out: their words keep the NaN)
          __builtin_amdgcn_global_load_lds(GPTR((const float*)a.d_mask + (int64_t)d * a.Ld + row), (__attribute__((address_space(3))) void*)(dm_dst + j * 256), 4, 0, 0);
      }
    } else if (wave == 7) {
#pragma unroll
      for (int j = 0; j < NQ / 2; ++j) {
        const int slot = 2 * j + (lane >> 5);
        const int qq = min(q0 + slot, a.nq - 1), t = min(lane & 31, a.Lq - 1);  // past nq / Lq: any valid word (weight 0 below)
        __builtin_amdgcn_global_load_lds(GPTR((const float*)a.q_mask + (int64_t)qq * a.Lq + t), (__attribute__((address_space(3))) void*)(qm_dst + j * 256), 4, 0, 0);
      }
    }
  };
  // the slice stream: (tile, slice) of the next slice to ISSUE and the tensor offset of its tile's doc / query block
  int is_ti = 0, is_s = 0;
  uint32_t is_base = 0;
  auto issue_setup = [&]() __attribute__((always_inline)) {  // before the first instruction of a slice
    if (is_s == 0) {
      const int u = l + is_ti * nl;
      is_base = (uint32_t)(first_half ? tile_doc(u) * a.Ld : tile_q0(u) * a.Lq) * rowb;
    }
  };
  auto issue_done = [&]() __attribute__((always_inline)) {
    if (++is_s == nslices) { is_s = 0; ++is_ti; }
  };
  // instruction j of this wave's part of the slice (is_ti, is_s) into ring stage st
#define AP_DMA(st, j)                                                                                                 \
  do {                                                                                                                \
    if (!AP_ABLATE_DMA)                                                                                               \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(src, (__attribute__((address_space(3))) void*)(lds3 + (st) * STAGE + dst_off + (j) * 4096), 16, \
                                               (int)(lane_off + (is_base + (uint32_t)is_s * 128u + (uint32_t)(j) * j_stride)), 0, 0, 0); \
  } while (0)

  // ---- compute side.  A fragment of row block b, k-step ks: row 32 b + r, chunk 2 ks + hh at position chunk ^ ((row >> 1) & 7)
  //      = (hh ^ swz) ^ 2 ks: the k-step is an XOR of bits 5-6 of the byte address
  const uint32_t swz = (uint32_t)((r >> 1) & 7);
  const uint32_t fa0 = (uint32_t)(((wm * R) * 32 + r) * 128) + ((hh ^ swz) << 4);
  const uint32_t fb0 = (uint32_t)((TM + (wn * QB) * 32 + r) * 128) + ((hh ^ swz) << 4);
  f32x16 acc[R][QB];
#pragma unroll
  for (int b = 0; b < R; ++b)
#pragma unroll
    for (int q = 0; q < QB; ++q) acc[b][q] = (f32x16)(0.0f);
  u32x4 fa[2][R], fb[2][QB];  // fragment sets, double-buffered by k-step
  // ds_read_b128 by hand: the compiler does not know what an LDS-DMA instruction writes and orders its own LDS reads
  // against them with s_waitcnt vmcnt(0); and it would wait for a set with lgkmcnt(0) right where it issues the next
#define AP_READ(set, i, aa, ab)                                                                                       \
  do {                                                                                                                \
    if ((i) < R) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][(i) < R ? (i) : 0]) : "v"(aa), "n"(((i) < R ? (i) : 0) * 4096)); \
    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][(i) >= R && (i) < NF ? (i) - R : 0]) : "v"(ab), "n"(((i) >= R && (i) < NF ? (i) - R : 0) * 4096)); \
  } while (0)
  auto wait_frags = [&](int set) __attribute__((always_inline)) {  // every LDS read issued so far has returned
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < R; ++b) asm volatile("" : "+v"(fa[set][b]));  // (the MFMAs below depend on this point)
#pragma unroll
    for (int q = 0; q < QB; ++q) asm volatile("" : "+v"(fb[set][q]));
  };
#define AP_MFMA(set, i, c0)                                                                                           \
  do {                                                                                                                \
    if (AP_ABLATE_MFMA) break;                                                                                        \
    const int q_ = (i) / R, b_ = (i) % R;                                                                             \
    if (AP_MFMA16) { /* timing experiment, WRONG results: the same flops, operands and cycles as two 16x16x32 */      \
      f32x4 c0_, c1_;                                                                                                 \
      _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) { c0_[t_] = acc[b_][q_][8 * (set) + t_]; c1_[t_] = acc[b_][q_][8 * (set) + 4 + t_]; } \
      if constexpr (DT == MAXSIM_F16) {                                                                               \
        c0_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[set][b_]), __builtin_bit_cast(f16x8, fb[set][q_]), c0_, 0, 0, 0); \
        c1_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, fa[set][b_]), __builtin_bit_cast(f16x8, fb[set][q_]), c1_, 0, 0, 0); \
      } else {                                                                                                        \
        c0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[set][b_]), __builtin_bit_cast(bf16x8, fb[set][q_]), c0_, 0, 0, 0); \
        c1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[set][b_]), __builtin_bit_cast(bf16x8, fb[set][q_]), c1_, 0, 0, 0); \
      }                                                                                                               \
      _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) { acc[b_][q_][8 * (set) + t_] = c0_[t_]; acc[b_][q_][8 * (set) + 4 + t_] = c1_[t_]; } \
      break;                                                                                                          \
    }                                                                                                                 \
    const f32x16 c_ = (c0) ? (f32x16)(0.0f) : acc[b_][q_];                                                            \
    if constexpr (DT == MAXSIM_F16)                                                                                   \
      acc[b_][q_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][b_]), __builtin_bit_cast(f16x8, fb[set][q_]), c_, 0, 0, 0); \
    else                                                                                                              \
      acc[b_][q_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][b_]), __builtin_bit_cast(bf16x8, fb[set][q_]), c_, 0, 0, 0); \
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

  // ---- the loop: one slice = one barrier + four k-steps of R QB MFMAs.  Between the MFMAs of a k-step: the R + QB fragment
  //      reads of the next k-step (none in the last: the next slice is only known to have landed after the barrier), then
  //      this wave's LDS-DMA instructions of the NEXT slice -- into the other stage, which everybody finished reading
  //      before the barrier -- in the k-steps of its half.
  int cs = 0, cti = 0;  // (slice in tile, tile) being computed
  for (int g = 0; g < total; ++g) {
    const int st = g & 1;
    // slice g has landed: this wave's part (everything it has in flight: the slice, a new tile's mask rows, the previous
    // tile's result stores), then everybody's.  The barrier also says: every wave is done reading slice g - 1.
    wait_vmcnt<0>();
    wg_barrier();
    const bool first = cs == 0;
    const bool do_issue = g + 1 < total;
    if (do_issue) issue_setup();
    if (masked && first && g > 0) issue_masks(cti);  // (tile 0's mask rows are issued in the prologue)
    const uint32_t aa = fa0 + (uint32_t)(st * STAGE), ab = fb0 + (uint32_t)(st * STAGE);
#pragma unroll
    for (int i = 0; i < NF; ++i) AP_READ(0, i, aa, ab);
    // DMA instructions of this wave placed behind MFMA i of k-step ks (slots NF .. NM - 1 of a k-step carry no read)
    constexpr int SLOTS = NM > NF ? NM - NF : 1;
    // when the second half fetches: all its instructions in one k-step, or split over two (measured per tile shape:
    // R = 3 k-step 1 alone, R = 2 k-steps 0 and 1 (+4 %), R = 1 no difference)
    constexpr bool BSPLIT = AP_DMA_BSPLIT != 0 || R == 2;
    constexpr int B0 = R == 2 ? 0 : AP_DMA_B0;
    constexpr int PER_A = (NDA / 2 + SLOTS - 1) / SLOTS, PER_B = (NDB + SLOTS - 1) / SLOTS, PER_B2 = (NDB / 2 + SLOTS - 1) / SLOTS;
#define AP_KSTEP(ks, C0)                                                                                              \
  {                                                                                                                   \
    wait_frags((ks) & 1);                                                                                             \
    const uint32_t an_ = aa ^ (((ks) + 1) << 5), bn_ = ab ^ (((ks) + 1) << 5);                                        \
    _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                                  \
      AP_MFMA((ks) & 1, i, C0);                                                                                       \
      if ((ks) + 1 < KS && i < NF) AP_READ(((ks) + 1) & 1, i, an_, bn_);                                              \
      if (i >= (NM > NF ? NF : NM - 1) && do_issue) {                                                                 \
        const int sl_ = NM > NF ? i - NF : 0;                                                                         \
        if (first_half && ((ks) == AP_DMA_A0 || (ks) == AP_DMA_A0 + 1)) {                                             \
          _Pragma("unroll") for (int t = 0; t < (NM > NF ? PER_A : NDA / 2); ++t) {                                   \
            const int j = ((ks) - AP_DMA_A0) * (NDA / 2) + sl_ * PER_A + t;                                           \
            if (sl_ * PER_A + t < NDA / 2) AP_DMA(st ^ 1, j < NDA ? j : 0);                                           \
          }                                                                                                           \
        }                                                                                                             \
        if (!first_half && !BSPLIT && (ks) == B0) {                                                       \
          _Pragma("unroll") for (int t = 0; t < (NM > NF ? PER_B : NDB); ++t) {                                       \
            const int j = sl_ * PER_B + t;                                                                            \
            if (j < NDB) AP_DMA(st ^ 1, j < NDB ? j : 0);                                                             \
          }                                                                                                           \
        }                                                                                                             \
        if (!first_half && BSPLIT && ((ks) == B0 || (ks) == B0 + 1)) {                                              \
          _Pragma("unroll") for (int t = 0; t < (NM > NF ? PER_B2 : NDB / 2); ++t) {                                  \
            const int j = ((ks) - B0) * (NDB / 2) + sl_ * PER_B2 + t;                                                 \
            if (sl_ * PER_B2 + t < NDB / 2) AP_DMA(st ^ 1, j < NDB ? j : 0);                                          \
          }                                                                                                           \
        }                                                                                                             \
      }                                                                                                               \
    }                                                                                                                 \
    if ((ks) + 1 < KS) { _Pragma("unroll") for (int i = NM; i < NF; ++i) AP_READ(((ks) + 1) & 1, i, an_, bn_); }      \
  }
    // (ONE copy of every k-step: the fragment reads are asynchronous behind the compiler's back -- where two branches
    //  that both issue them meet, it may unify their destination registers with copies, i.e. read them before the data
    //  is there.  A tile starts from accumulators zeroed at the end of the previous tile's epilogue instead of from a
    //  C = 0 copy of its first k-step.)
    AP_KSTEP(0, false)
    AP_KSTEP(1, false)
    AP_KSTEP(2, false)
    AP_KSTEP(3, false)
#undef AP_KSTEP
    if (do_issue) issue_done();
    if (++cs < nslices) continue;

    // ---- epilogue of tile cti: similarities complete ---------------------------------------------------------------
    const int u = l + cti * nl;
    const int d = tile_doc(u), q0 = tile_q0(u);
    const float* const dm = dm_lds;
    const float* const qmp = qm_lds;
    // What the d_mask words of this wave's 32-row blocks are, block by block (wave-uniform): all 1 (a token row of the
    // doc: nothing to multiply), all 0 (padding inside Ld: every similarity is 0, the block's first row is its only
    // candidate), all NaN (tile padding past Ld: no candidate), or mixed.  With the reference's prefix masks
    // (tokenizers.py:57) at most one block of a doc is mixed.
    enum { BLK_MIXED = 0, BLK_ONES = 1, BLK_ZEROS = 2, BLK_NONE = 3 };
    int kind[R];
#pragma unroll
    for (int b = 0; b < R; ++b) {
      const float w = dm[(wm * R + b) * 32 + r];
      const bool ones = __builtin_amdgcn_ballot_w64(w == 1.0f) == ~0ull, zeros = __builtin_amdgcn_ballot_w64(w == 0.0f) == ~0ull;
      const bool nans = __builtin_amdgcn_ballot_w64(w != w) == ~0ull;
      kind[b] = ones ? BLK_ONES : zeros ? BLK_ZEROS : nans ? BLK_NONE : BLK_MIXED;
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {  // one query at a time: the accumulators leave few registers for anything else
      const int qq = q0 + wn * QB + q;
      // tokens past Lq and query slots past nq: weight 0 (similarity 0, nothing written)
      const float qm = (qq < a.nq && r < a.Lq) ? qmp[(wn * QB + q) * 32 + r] : 0.0f;
      // q_mask >= 0 (the reference's masks are 0/1: tokenizers.py:36,57) commutes with the max: max_n(qm x_n) = qm max_n(x_n),
      // one multiplication per token instead of one per similarity.  A negative weight anywhere in the wave (never built
      // by the reference, allowed by its interface) takes the multiplication back into the mask words of every block.
      const bool premul = __builtin_amdgcn_ballot_w64(qm < 0.0f) != 0;
      float best = NEG_INF;
      int bidx = 0;  // position b * 16 + v of the winner among this lane's values (compile-time numbers: no arithmetic per
                     // value; turned into a row number once, below)
      // (copies of the scan chosen by wave-uniform branches: a conditional multiplication inside one copy cost 12
      //  registers -> spills)
#define AP_PICK(sim_, pos_)                                                                                           \
  do {                                                                                                                \
    if constexpr (AM) {                                                                                               \
      const bool better = (sim_) > best; /* strict >: the first maximal token wins (torch.max); a NaN never does */    \
      best = better ? (sim_) : best;                                                                                  \
      bidx = better ? (pos_) : bidx;                                                                                  \
    } else {                                                                                                          \
      best = __builtin_fmaxf(best, (sim_)); /* (maxNum: a NaN operand is dropped; pairs fuse into v_max3_f32) */      \
    }                                                                                                                 \
  } while (0)
#define AP_SCAN(PREMUL)                                                                                               \
  _Pragma("unroll") for (int b = 0; b < R; ++b) {                                                                     \
    if (!PREMUL && kind[b] == BLK_ONES) {                                                                             \
      _Pragma("unroll") for (int v = 0; v < 16; ++v) AP_PICK(acc[b][q][v], b * 16 + v);                               \
    } else if (!PREMUL && kind[b] == BLK_ZEROS) {                                                                     \
      AP_PICK(0.0f, b * 16);                                                                                          \
    } else if (kind[b] != BLK_NONE) {                                                                                 \
      int rowbase = (wm * R + b) * 32 + 4 * hh;                                                                       \
      /* opaque: otherwise the mask words are shared by the QB unrolled query iterations and stay live across them */ \
      asm volatile("" : "+v"(rowbase));                                                                               \
      /* d_mask of this lane's 16 rows of the block: four 16-byte reads up front, ONE wait; NaN past Ld */            \
      f32x4 d4[4];                                                                                                    \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) d4[k] = *(const f32x4*)(dm + rowbase + 8 * k);                    \
      _Pragma("unroll") for (int v = 0; v < 16; v += 2) { /* rows in increasing order, two at a time (v_pk_mul_f32) */ \
        f32x2 w = {d4[v >> 2][v & 3], d4[v >> 2][(v & 3) + 1]};                                                       \
        if (PREMUL) w *= qm;                                                                                          \
        f32x2 sim = {acc[b][q][v], acc[b][q][v + 1]};                                                                 \
        sim *= w; /* . d_mask (. q_mask: in w, or after the scan) */                                                  \
        AP_PICK(sim[0], b * 16 + v);                                                                                  \
        AP_PICK(sim[1], b * 16 + v + 1);                                                                              \
      }                                                                                                               \
    }                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0); /* (one block's mask words at a time) */                                       \
  }
      if (premul) { AP_SCAN(true) } else { AP_SCAN(false) }
#undef AP_SCAN
#undef AP_PICK
      // (Q q_mask) . (D d_mask), BaseModel.py:41-43.  q_mask = 0 makes every similarity of the token 0: the first row wins
      const bool qzero = !premul && qm == 0.0f;
      if (!premul) best = qzero ? 0.0f : best * qm;
      if (AM) bidx = qzero ? (wm == 0 && hh == 0 ? 0 : 0x7fffffff) : (wm * R + (bidx >> 4)) * 32 + 4 * hh + (bidx & 3) + 8 * ((bidx & 15) >> 2);
      // the two lane halves hold interleaved rows of the same query token
      const auto sv = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
      const float va = __uint_as_float(sv[0]), vb = __uint_as_float(sv[1]);
      float v2;
      int i2 = 0;
      if constexpr (AM) {
        const auto si = __builtin_amdgcn_permlane32_swap((uint32_t)bidx, (uint32_t)bidx, false, false);
        const int ia = (int)si[0], ib = (int)si[1];
        const bool take_b = (vb > va) || (vb == va && ib < ia);
        v2 = take_b ? vb : va;
        i2 = take_b ? ib : ia;
      } else {
        v2 = fmaxf(va, vb);
      }
      if (lane < 32) {
        ex_v[((wn * QB + q) * 4 + wm) * 32 + lane] = v2;
        if (AM) ex_i[((wn * QB + q) * 4 + wm) * 32 + lane] = i2;
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
    // (the exchange area and the mask rows are rewritten only after the next barrier)
#pragma unroll
    for (int b = 0; b < R; ++b)
#pragma unroll
      for (int q = 0; q < QB; ++q) acc[b][q] = (f32x16)(0.0f);
    cs = 0;
    ++cti;
  }
#undef AP_DMA
#undef AP_READ
#undef AP_MFMA
}

}  // namespace maxsim
