// maxsim_allpairs.h -- the all-pairs (training-form) MaxSim as a register-blocked, K-sliced GEMM with a fused
// max / arg-max / sum epilogue: 16-bit Q and D of any width h that is a multiple of 32 (the reference trains at dim 768,
// proj_conf/dense.yaml:8), Lq <= 32 query tokens, Ld <= 384 doc tokens (doc_maxlen, dense.yaml:7).
//
// Reference: BaseModel.score (colbert/modeling/BaseModel.py:39-46) as ColbertModel.forward calls it on the gathered batch
// (colbert/modeling/colbert_model.py:87-90): simmat[q,d,m,n] = <Q[q,m] q_mask[q,m], D[d,n] d_mask[d,n]>, max over n
// (torch.max: the FIRST maximal index is what autograd routes the gradient through), sum over m.  At the reference's
// step (Q 272 x 32 x 768, D 544 x 384 x 768) that is a 208 896 x 8 704 x 768 GEMM = 2.79 PFLOP whose result never has
// to exist: only 148 k scores and 4.7 M arg-max indices leave the kernel.
//
// The streaming kernel (maxsim_stream_bigh.h) computes this with one wave per 32-row doc tile and whole query images
// in LDS: every 8 KB A fragment set is read for QB x 8 MFMAs, 2 KB of LDS traffic per MFMA at QB = 2 -- the LDS peak
// (128 B/clk/CU) at full matrix rate, measured 22 % of the bf16 peak.  Here the blocking is a GEMM's:
//   workgroup  8 waves = 4 (doc rows) x 2 (queries); tile = ONE doc (up to 128 R rows, R = 1..3) x 2 QB queries
//   wave       R row blocks x QB queries of 32x32x16 MFMAs: R + QB fragment reads (1 KB each) per R QB MFMAs, 16 R QB
//              accumulator registers.  (R, QB) = (1, 4), (2, 4), (3, 3): with 8 waves a wave has 256 registers, and
//              3 x 4 blocks (192 accumulators) spilled; 3 x 3 = 0.67 KB of LDS reads per MFMA
//   K          sliced by 32 dims: a slice is (128 R + 64 QB) rows x 64 B, fetched by LDS-DMA (global_load_lds_dwordx4, no
//              register staging; source addresses = wave-uniform 64-bit base + a 32-bit lane offset computed once) into a
//              4-stage ring: slice g being read, g + 1 complete (its first fragments are requested across the barrier),
//              g + 2 and g + 3 in flight.  ONE bare s_barrier per slice (__syncthreads would wait for every DMA in
//              flight), counted s_waitcnt vmcnt for the slice that must have landed; fragment reads are ds_read_b128 by
//              hand, double-buffered, placed between the MFMAs of the previous k-step
//   tiles      workgroups are persistent (one per CU: ~150 KB of LDS) and walk (doc, query block) tiles; the ring runs
//              across tile boundaries, so the next tile's first slices arrive during the epilogue.  XCD x takes the docs
//              x, x + 8, ... and walks their query blocks in order: the ~32 workgroups of an XCD work on one or two docs at
//              a time (the doc comes from that XCD's L2; the 13 MB of queries from the Infinity Cache)
//   epilogue   per lane and query: running (max, first index) over its rows in increasing row order, lane halves combined
//              with v_permlane32_swap, the four row-waves through LDS; masks multiply the finished similarities (exact
//              for 0/1 masks, the only ones training uses: tokenizers.py:36,57); 0-padding rows past Ld never win.
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
#ifdef MAXSIM_DIAG
  int stamp_g0, stamp_wg;  // diagnostic build: s_memtime stamps of slices stamp_g0 .. + 23 of workgroup stamp_wg
#endif
};

#ifdef MAXSIM_DIAG
// In-kernel stamps (diagnostic build only; tools/allpairs_stamps.py reads them): [2 waves][24 slices][9 points].
// A stamp is ONE scalar instruction whose result is not waited for (a wait would also drain the LDS reads in flight).
// The nine results of a slice -- the ninth is the top of the NEXT iteration -- are stored right after that top stamp, so
// the cost of storing them falls into the first segment of the next slice (top -> vmcnt wait) and nowhere else.  Kept in
// LDS while the loop runs (a global store would count in vmcnt and move the loop's counted waits), dumped at the end.
__device__ uint64_t g_ap_stamps[2 * 24 * 9];
#define AP_STAMP(k) asm volatile("s_memtime %0" : "=s"(tS[k]))
#define AP_STAMP_TOP()                                                                           \
  do {                                                                                           \
    AP_STAMP(8);                                                                                 \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
    if (stamp_w >= 0 && g > a.stamp_g0 && g <= a.stamp_g0 + 24 && lane < 9)                      \
      st_lds[(stamp_w * 24 + (g - 1 - a.stamp_g0)) * 9 + lane] = lane == 0 ? tS[0] : lane == 1 ? tS[1] : lane == 2 ? tS[2] : lane == 3 ? tS[3] : lane == 4 ? tS[4] : lane == 5 ? tS[5] : lane == 6 ? tS[6] : lane == 7 ? tS[7] : tS[8]; \
    tS[0] = tS[8];                                                                               \
    tS[7] = 0;                                                                                   \
  } while (0)
#else
#define AP_STAMP(k) do {} while (0)
#define AP_STAMP_TOP() do {} while (0)
#endif

template <int CTRL>
__device__ __forceinline__ float ap_dpp(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, false));
}

// The bare hardware barrier.  __syncthreads() is a workgroup fence + barrier, and the fence waits for EVERY outstanding
// memory operation of the wave (s_waitcnt vmcnt(0)) -- including the LDS-DMA slices this kernel keeps in flight on
// purpose: with it the ring never ran ahead and every slice exposed the full memory latency (35 % MFMA busy).  The loop
// orders its data by hand: counted vmcnt for the slice that must have landed, then the barrier.
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

// (timing experiments of the diagnostic build: -DAP_ABLATE_DMA=1 drops the loop's LDS-DMA instructions, -DAP_ABLATE_MFMA=1
//  its MFMAs; results are then wrong, only the time is of interest)
#ifndef AP_ABLATE_DMA
#define AP_ABLATE_DMA 0
#endif
#ifndef AP_ABLATE_MFMA
#define AP_ABLATE_MFMA 0
#endif

template <int DT, int R, int QB, bool AM>
__global__ void __launch_bounds__(512) k_maxsim_allpairs(const AllPairsArgs a) {
  constexpr int WV = 8;  // waves: 4 (doc rows) x 2 (queries).  (Four waves of twice the rows, one per SIMD with 512 registers,
                         // were slower: the accumulators must then fit the 256 AGPRs, i.e. 6 x 2 blocks at best.)
  static_assert(DT == MAXSIM_F16 || DT == MAXSIM_BF16, "16-bit operands");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  constexpr int NQ = 2 * QB;                     // queries of a tile
  constexpr int TM = 128 * R, TN = 32 * NQ;      // tile rows (doc tokens) / columns (query tokens)
  constexpr int ROWS = TM + TN;                  // rows of a K slice image
  constexpr int STAGE = ROWS * 64;               // bytes: 32 dims x 2 B per row
  constexpr int NST = 4;                         // ring stages: slice g being read, g + 1 ready (its first fragments are
                                                 // prefetched across the barrier), g + 2 and g + 3 in flight
  constexpr int NI = ROWS / 16;                  // LDS-DMA instructions per slice (16 rows each)
  constexpr int WM = WV / 2;                     // row-waves
  constexpr int RW = 4 * R / WM;                 // 32-row blocks of a wave
  constexpr int NAI = TM / 16;                   // instructions < NAI move doc rows, the others query rows
  constexpr int HW = WV / 2;                     // waves 0 .. HW - 1 fetch the doc rows, HW .. WV - 1 the query rows: a wave
  constexpr int NDA = NAI / HW;                  // and its SIMD partner (w, w + HW) sit in different halves
  constexpr int NDB = (NI - NAI) / HW;           // LDS-DMA instructions per slice of a doc-row wave (2 R) / query-row wave (QB)
  constexpr int NOFF = NDA > 2 * NDB ? NDA : 2 * NDB;
  float* const ex_v = (float*)(lds + NST * STAGE);      // [NQ][4 wm][32]: per-wave (max) ...
  int* const ex_i = (int*)(ex_v + NQ * 4 * 32);         // ... and (first index)
  float* const dm_lds = (float*)(ex_i + NQ * 4 * 32);   // [TM]: d_mask row of the tile's doc
  float* const qm_lds = dm_lds + TM;                    // [NQ * 32]: q_mask rows of the tile's queries
  // (LDS-space views for the DMA destinations, cast here in uniform control flow: the generic -> LDS cast inside a
  //  divergent branch trips a code-generation bug of this compiler)
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const lds_ptr_t dm_dst = (lds_ptr_t)LPTR(dm_lds), qm_dst = (lds_ptr_t)LPTR(qm_lds);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uni(tid >> 6), wm = wave & (WM - 1), wn = wave / WM;
  const int r = lane & 31, hh = lane >> 5;
  const int nslices = a.h >> 5;
  const int64_t rowb = (int64_t)a.h * 2;
  const int nqb = (a.nq + NQ - 1) / NQ;
  const bool fetch_docs = wave < HW;
  const int hw = wave & (HW - 1);
#ifdef MAXSIM_DIAG
  uint64_t* const st_lds = (uint64_t*)(qm_lds + NQ * 32);
  const int stamp_w = (int)blockIdx.x != a.stamp_wg ? -1 : wave == 0 ? 0 : wave == WV - 1 ? 1 : -1;
  if ((int)blockIdx.x == a.stamp_wg) {
    for (int i = tid; i < 2 * 24 * 9; i += WV * 64) st_lds[i] = 0;
  }
#endif

  // ---- this workgroup's tiles: XCD x = id % 8 owns docs x, x + 8, ...; its workgroups walk (doc, query block) in order
  const int x = blockIdx.x & 7, l = blockIdx.x >> 3, nl = max(1, (int)gridDim.x >> 3);
  const int ndx = (a.nd - x + 7) >> 3;
  const int ntx = ndx * nqb;  // tiles of this XCD
  auto tile_doc = [&](int u) { return x + 8 * (u / nqb); };
  auto tile_q0 = [&](int u) { return NQ * (u % nqb); };
  const int my_tiles = l < ntx ? (ntx - l + nl - 1) / nl : 0;
  const int total = my_tiles * nslices;  // slices this workgroup streams

  // ---- fetch side.  DMA instruction j of this wave moves rows 16 (wave + 8 j) .. + 15 of the slice image: lane i -> row
  //      + i / 4, 16-byte position i % 4, which receives source chunk (i % 4) ^ ((row >> 2) & 3) (fragment reads are
  //      conflict-free).  Everything lane-dependent is a 32-bit offset computed ONCE (the address of an instruction is a
  //      wave-uniform 64-bit base + that offset: no vector arithmetic per slice -- with two waves per SIMD in the same
  //      phase, every VALU cycle spent here is a cycle the matrix pipe idles).
  //      Doc-row wave hw moves image rows 16 (hw + HW j) .. + 15 (j < NDA), query-row wave hw the rows TM + 16 (hw + HW j) ..
  //      (j < NDB).  off[j]: doc rows min(row, Ld - 1) * rowb + chunk; query rows (slot * Lq + min(token, Lq - 1)) * rowb +
  //      chunk and, at NDB + j, the same for slot 0 (used for slots past nq in the last, partial query block).
  uint32_t off[NOFF];
#pragma unroll
  for (int j = 0; j < NOFF; ++j) {
    const int jj = fetch_docs ? j : j % NDB;
    const int lr = 16 * (hw + HW * jj) + (lane >> 2);  // row within the doc part / the query part of the image
    const uint32_t chunk = (uint32_t)(((lane & 3) ^ ((lr >> 2) & 3)) * 16);  // (TM % 16 == 0: the same swizzle either way)
    const int slot = lr >> 5, t = min(lr & 31, a.Lq - 1);  // tokens past Lq re-read the last token (weight 0)
    const uint32_t od = (uint32_t)(min(lr, a.Ld - 1) * (int)rowb);  // rows past Ld re-read the last row (never candidates)
    const uint32_t oq = (uint32_t)(((j < NDB ? slot * a.Lq : 0) + t) * (int)rowb);
    off[j] = (fetch_docs ? od : oq) + chunk;
  }
  // the slice stream: (tile, slice) of the next slice to ISSUE, kept incrementally
  int is_ti = 0, is_s = 0;
  const char* is_dbase = nullptr;  // D + doc * Ld * rowb
  const char* is_qbase = nullptr;  // Q + q0 * Lq * rowb
  int is_nvalid = 0;               // query slots of the block that exist
  auto issue_tile_setup = [&]() __attribute__((always_inline)) {
    const int u = l + is_ti * nl;
    const int d = tile_doc(u), q0 = tile_q0(u);
    is_dbase = (const char*)a.D + ((int64_t)d * a.Ld) * rowb;
    is_qbase = (const char*)a.Q + ((int64_t)q0 * a.Lq) * rowb;
    is_nvalid = min(NQ, a.nq - q0);
  };
  auto issue = [&](int g) __attribute__((always_inline)) {  // issues slice g of this workgroup's stream (called with g = 0, 1, 2, ...)
    if (is_s == 0) issue_tile_setup();
    char* const dst = lds + (g % NST) * STAGE + ((fetch_docs ? 0 : NAI) + hw) * 1024;
    const char* const base = (fetch_docs ? is_dbase : is_qbase) + is_s * 64;
    if (fetch_docs) {
#pragma unroll
      for (int j = 0; j < NDA; ++j) __builtin_amdgcn_global_load_lds(GPTR(base + off[j]), LPTR(dst + j * (HW * 1024)), 16, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < NDB; ++j)  // (16 rows = half a query slot: the slot number is wave-uniform; slots past nq re-read slot 0)
        __builtin_amdgcn_global_load_lds(GPTR(base + (((hw + HW * j) >> 1) < is_nvalid ? off[j] : off[NDB + j])), LPTR(dst + j * (HW * 1024)), 16, 0, 0);
    }
    if (++is_s == nslices) { is_s = 0; ++is_ti; }
  };

  // ---- compute side: fragment addresses inside a stage
  //      A fragment of row block b, k-step ks: row 32 b + r, chunk 2 ks + hh at position chunk ^ ((row >> 2) & 3)
  const int swz = (r >> 2) & 3;
  f32x16 acc[RW][QB];
#pragma unroll
  for (int b = 0; b < RW; ++b)
#pragma unroll
    for (int q = 0; q < QB; ++q) acc[b][q] = (f32x16)(0.0f);

  // fragment sets, double-buffered: while the MFMAs of one k-step run, the next k-step's fragments are on their way from
  // LDS (a workgroup's waves hit the barrier together; without this overlap the LDS read phase -- 96 KB per slice and CU,
  // 768 cycles at 128 B/clk -- and the MFMA phase -- 1152 cycles per SIMD -- serialise: measured 35 % MFMA busy)
  u32x4 fa[2][RW], fb[2][QB];
  auto load_frags = [&](int set, int g, int ks) __attribute__((always_inline)) {
    const char* const st = lds + (g % NST) * STAGE;
    const int pos = ((2 * ks + hh) ^ swz) * 16;
    // ds_read_b128 by hand: the compiler would wait for a set with s_waitcnt lgkmcnt(0) -- i.e. also for the set it has
    // just issued -- because it cannot count LDS returns across the loop back edge; wait_frags counts them instead
    const uint32_t aa = (uint32_t)(size_t)(st + ((wm * RW) * 32 + r) * 64 + pos - lds);
    const uint32_t ab = (uint32_t)(size_t)(st + (TM + (wn * QB) * 32 + r) * 64 + pos - lds);
#pragma unroll
    for (int b = 0; b < RW; ++b) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[set][b]) : "v"(aa), "n"(b * 2048));
#pragma unroll
    for (int q = 0; q < QB; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[set][q]) : "v"(ab), "n"(q * 2048));
  };
  // every LDS read issued so far has returned (the reads of a set are issued a whole k-step before they are needed)
  auto wait_frags = [&](int set) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < RW; ++b) asm volatile("" : "+v"(fa[set][b]));  // (the MFMAs below depend on this point)
#pragma unroll
    for (int q = 0; q < QB; ++q) asm volatile("" : "+v"(fb[set][q]));
  };
  // masks (float32 or none -- the launcher sends other mask types to the streaming kernel): the tile's mask rows come in
  // by LDS-DMA with the tile's first slices, like everything else.  An ordinary load anywhere in this loop would make the
  // compiler wait for its result with s_waitcnt vmcnt(0), i.e. for every slice in flight: the ring would never run ahead
  // (measured: that alone held the kernel at 35 % MFMA busy).
  const bool masked = a.mask_dtype != MAXSIM_MASK_NONE;
  // rows past Ld are tile padding: their mask word is NaN for the whole kernel, so their products are NaN and never win a
  // `>` (no bound test per element in the epilogue); without masks every real row / token weighs 1
  for (int i = tid; i < TM + NQ * 32; i += WV * 64) {
    const bool pad = i >= a.Ld && i < TM;
    if (pad || !masked) dm_lds[i] = pad ? __builtin_nanf("") : 1.0f;
  }
  auto issue_masks = [&](int ti) __attribute__((always_inline)) {  // the last two waves, after their part of a tile's first slice
    const int u = l + ti * nl;
    const int d = tile_doc(u), q0 = tile_q0(u);
    if (wave == WV - 2) {
#pragma unroll
      for (int j = 0; j < TM / 64; ++j) {
        const int row = j * 64 + lane;
        if (row < a.Ld)  // (lanes past Ld stay out: their words keep the NaN)
          __builtin_amdgcn_global_load_lds(GPTR((const float*)a.d_mask + (int64_t)d * a.Ld + row), (__attribute__((address_space(3))) void*)(dm_dst + j * 256), 4, 0, 0);
      }
    } else if (wave == WV - 1) {
#pragma unroll
      for (int j = 0; j < NQ / 2; ++j) {
        const int slot = 2 * j + (lane >> 5);
        const int qq = min(q0 + slot, a.nq - 1), t = min(lane & 31, a.Lq - 1);  // past nq / Lq: any valid word (weight 0 below)
        __builtin_amdgcn_global_load_lds(GPTR((const float*)a.q_mask + (int64_t)qq * a.Lq + t), (__attribute__((address_space(3))) void*)(qm_dst + j * 256), 4, 0, 0);
      }
    }
  };
  if (total > 0) issue(0);
  if (total > 1) issue(1);
  if (total > 2) issue(2);
  if (total > 0 && masked) issue_masks(0);
  if (total > 0) {  // slice 0 for everybody, its first fragments on their way
    if (masked && wave >= WV - 2) wait_vmcnt<0>();  // (their mask rows sit behind the slices in the queue: once, at start)
    else if (total > 2) { if (fetch_docs) wait_vmcnt<2 * NDA>(); else wait_vmcnt<2 * NDB>(); }
    else if (total > 1) { if (fetch_docs) wait_vmcnt<NDA>(); else wait_vmcnt<NDB>(); }
    else wait_vmcnt<0>();
    wg_barrier();
    load_frags(0, 0, 0);
  }

  // Per-iteration bookkeeping (loop position, which wait, the LDS-DMA bases of slice g + 3, fragment addresses): ~100 scalar
  // and vector instructions.  A wave issues at most one instruction every 4 cycles, an MFMA keeps the matrix pipe busy for
  // 32, and after every barrier all eight waves are in the same phase: whatever is not placed BETWEEN a wave's MFMAs in
  // program order adds to the 1152 cycles of MFMA time per slice instead of hiding under it (measured: the loop skeleton
  // alone, with the MFMAs and every memory instruction taken out, cost 1040 cycles per slice).  So the state of
  // iteration g + 1 is prepared in the middle of iteration g's second k-step.
  struct Step {
    int s, ti;
    bool first, do_issue, do_masks, vm_all;
    char* dma_dst;
    const char* dma_src;  // doc rows or query rows of slice g + 3, by the wave's half
    int nvalid;
    uint32_t a1, b1, a0, b0;
  };
  int lp_s = -1, lp_ti = 0;
  auto prep = [&](int g) __attribute__((always_inline)) -> Step {
    Step t;
    if (++lp_s == nslices) { lp_s = 0; ++lp_ti; }
    t.s = lp_s;
    t.ti = lp_ti;
    t.first = lp_s == 0;
    t.do_issue = g + 3 < total;
    t.do_masks = masked && t.first && g > 0;  // (tile 0's mask rows are issued in the prologue)
    t.vm_all = g + 2 >= total;
    t.dma_dst = nullptr;
    t.dma_src = nullptr;
    if (t.do_issue) {
      if (is_s == 0) issue_tile_setup();
      t.dma_dst = lds + ((g + 3) % NST) * STAGE + ((fetch_docs ? 0 : NAI) + hw) * 1024;
      t.dma_src = (fetch_docs ? is_dbase : is_qbase) + is_s * 64;
      if (++is_s == nslices) { is_s = 0; ++is_ti; }
    }
    t.nvalid = is_nvalid;
    // fragment addresses: (g, k-step 1) for set 1, (g + 1, k-step 0) for set 0 (past the last slice: a stale stage, dropped)
    const uint32_t st1 = (uint32_t)((g % NST) * STAGE), st0 = (uint32_t)(((g + 1) % NST) * STAGE);
    const uint32_t pos1 = (uint32_t)(((2 + hh) ^ swz) * 16), pos0 = (uint32_t)((hh ^ swz) * 16);
    t.a1 = st1 + ((wm * RW) * 32 + r) * 64 + pos1;
    t.b1 = st1 + (TM + (wn * QB) * 32 + r) * 64 + pos1;
    t.a0 = st0 + ((wm * RW) * 32 + r) * 64 + pos0;
    t.b0 = st0 + (TM + (wn * QB) * 32 + r) * 64 + pos0;
    return t;
  };
  Step cur = prep(0);
#ifdef MAXSIM_DIAG
  uint64_t tS[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (int g = 0; g < total; ++g) {
    const int s = cur.s, ti = cur.ti;
    // slice g + 1 has landed: this wave's part (counted: the instructions of slice g + 2 may still be in flight.  Loads retire
    // in order, so "at most one slice's worth outstanding" implies slice g + 1 is in whatever the epilogue's younger stores and the
    // mask rows are doing: at worst the wait runs a few instructions into slice g + 2), then everybody's.  The barrier also
    // says: every wave is done reading slice g - 1, whose stage slice g + 3 overwrites.
    AP_STAMP_TOP();
    if (cur.vm_all) wait_vmcnt<0>(); else if (fetch_docs) wait_vmcnt<NDA>(); else wait_vmcnt<NDB>();
    AP_STAMP(1);
    wg_barrier();
    AP_STAMP(2);
    // One slice = two k-steps of R QB MFMAs, with the R + QB fragment reads of the next k-step and the LDS-DMA
    // instructions of slice g + 3 between them.
    constexpr int NM = RW * QB, NF = RW + QB;
    constexpr int SLOTS = NM > NF ? NM - NF : 1;  // MFMAs of a k-step that have no fragment read behind them
    constexpr int PER_A = (NDA + SLOTS - 1) / SLOTS, PER_B = (NDB + SLOTS - 1) / SLOTS;  // LDS-DMA instructions per such MFMA
    // The two waves of a SIMD (w and w + HW) issue their LDS-DMA instructions in DIFFERENT k-steps -- the doc-row waves in
    // k-step 0, the query-row waves in k-step 1: an LDS-DMA instruction holds the issuing wave for 100-180 cycles (stamped:
    // a k-step with 4-5 of them took 780-1080 cycles, one without 200-260), and with both partners stalled at the same
    // point of the slice nobody fed the matrix pipe meanwhile.  The doc rows are 2/3 of a slice: k-step 1 also carries
    // the next iteration's bookkeeping.
    const bool do_issue0 = cur.do_issue && fetch_docs, do_issue1 = cur.do_issue && !fetch_docs;
    const bool first = cur.first;
    char* const dma_dst = cur.dma_dst;
    const char* const dma_src = cur.dma_src;
#define AP_DMA_A(j)                                                                                                   \
  do {                                                                                                                \
    if (!AP_ABLATE_DMA) __builtin_amdgcn_global_load_lds(GPTR(dma_src + off[j]), LPTR(dma_dst + (j) * (HW * 1024)), 16, 0, 0); \
  } while (0)
#define AP_DMA_B(j)                                                                                                   \
  do {                                                                                                                \
    if (!AP_ABLATE_DMA)                                                                                               \
      __builtin_amdgcn_global_load_lds(GPTR(dma_src + (((hw + HW * (j)) >> 1) < is_nvalid_cur ? off[j] : off[NDB + (j)])), LPTR(dma_dst + (j) * (HW * 1024)), 16, 0, 0); \
  } while (0)
#define AP_MFMA(set, i, c0)                                                                                           \
  do {                                                                                                                \
    if (AP_ABLATE_MFMA) break;                                                                                        \
    const int q_ = (i) / RW, b_ = (i) % RW;                                                                            \
    const f32x16 c_ = (c0) ? (f32x16)(0.0f) : acc[b_][q_];                                                            \
    if constexpr (DT == MAXSIM_F16)                                                                                   \
      acc[b_][q_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[set][b_]), __builtin_bit_cast(f16x8, fb[set][q_]), c_, 0, 0, 0); \
    else                                                                                                              \
      acc[b_][q_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[set][b_]), __builtin_bit_cast(bf16x8, fb[set][q_]), c_, 0, 0, 0); \
  } while (0)
    const int is_nvalid_cur = cur.nvalid;
    const uint32_t a1 = cur.a1, b1 = cur.b1, a0 = cur.a0, b0 = cur.b0;
    wait_frags(0);  // set 0 (requested during the previous k-step) is in: no younger LDS read is outstanding here
    if (cur.do_masks) issue_masks(ti);
    AP_STAMP(3);
    // ---- k-step 0: MFMAs on set 0, between them the reads of set 1, then the LDS-DMA instructions of slice g + 3
    //      (two copies, so that "a tile's first k-step starts from C = 0" is ONE branch per slice, not one per MFMA)
#define AP_KSTEP0(C0)                                                                                                 \
  _Pragma("unroll") for (int i = 0; i < NM; ++i) {                                                                    \
    AP_MFMA(0, i, C0);                                                                                                \
    if (i < RW) {                                                                                                     \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[1][i < RW ? i : 0]) : "v"(a1), "n"(i * 2048));           \
    } else if (i < NF) {                                                                                              \
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[1][i >= RW && i < NF ? i - RW : 0]) : "v"(b1), "n"((i - RW) * 2048)); \
    } else if (do_issue0) {                                                                                           \
      _Pragma("unroll") for (int t = 0; t < PER_A; ++t) {                                                             \
        const int j = (i - NF) * PER_A + t;                                                                           \
        if (j < NDA) AP_DMA_A(j < NDA ? j : 0);                                                                       \
      }                                                                                                               \
    }                                                                                                                 \
  }
    if (first) { AP_KSTEP0(true) } else { AP_KSTEP0(false) }
#undef AP_KSTEP0
    // (what did not fit between the MFMAs: R QB < R + QB happens for R = 1)
#pragma unroll
    for (int i = NM; i < NF; ++i) {
      if (i < RW) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[1][i < RW ? i : 0]) : "v"(a1), "n"(i * 2048));
      else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[1][i >= RW && i < NF ? i - RW : 0]) : "v"(b1), "n"((i - RW) * 2048));
    }
    if (NM <= NF && do_issue0) {
#pragma unroll
      for (int j = 0; j < NDA; ++j) AP_DMA_A(j);
    }
    AP_STAMP(4);
    wait_frags(1);
    AP_STAMP(5);
    // ---- k-step 1: MFMAs on set 1, between them the reads of the next slice's set 0
    Step nxt = cur;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      AP_MFMA(1, i, false);
      if (i < RW) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[0][i < RW ? i : 0]) : "v"(a0), "n"(i * 2048));
      else if (i < NF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[0][i >= RW && i < NF ? i - RW : 0]) : "v"(b0), "n"((i - RW) * 2048));
      else if (do_issue1) {
#pragma unroll
        for (int t = 0; t < PER_B; ++t) {
          const int j = (i - NF) * PER_B + t;
          if (j < NDB) AP_DMA_B(j < NDB ? j : 0);
        }
      }
      if (i == (NM > 2 ? 2 : NM - 1)) nxt = prep(g + 1);  // the next iteration's bookkeeping, under this k-step's MFMAs
    }
#pragma unroll
    for (int i = NM; i < NF; ++i) {
      if (i < RW) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[0][i < RW ? i : 0]) : "v"(a0), "n"(i * 2048));
      else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[0][i >= RW && i < NF ? i - RW : 0]) : "v"(b0), "n"((i - RW) * 2048));
    }
    if (NM <= NF && do_issue1) {
#pragma unroll
      for (int j = 0; j < NDB; ++j) AP_DMA_B(j);
    }
#undef AP_DMA_A
#undef AP_DMA_B
#undef AP_MFMA
    AP_STAMP(6);
    if (s + 1 < nslices) {
      cur = nxt;
      continue;
    }

    // ---- epilogue of tile ti: similarities complete ----------------------------------------------------------------
    const int u = l + ti * nl;
    const int d = tile_doc(u), q0 = tile_q0(u);
    // What the d_mask words of this wave's 32-row blocks are, block by block (wave-uniform): all 1 (a token row of the
    // doc: nothing to multiply), all 0 (padding inside Ld: every similarity is 0, the block's first row is its only
    // candidate), all NaN (tile padding past Ld: no candidate), or mixed.  With the reference's prefix masks
    // (tokenizers.py:57) at most one block of a doc is mixed.
    enum { BLK_MIXED = 0, BLK_ONES = 1, BLK_ZEROS = 2, BLK_NONE = 3 };
    int kind[RW];
#pragma unroll
    for (int b = 0; b < RW; ++b) {
      const float w = dm_lds[(wm * RW + b) * 32 + r];
      const bool ones = __builtin_amdgcn_ballot_w64(w == 1.0f) == ~0ull, zeros = __builtin_amdgcn_ballot_w64(w == 0.0f) == ~0ull;
      const bool nans = __builtin_amdgcn_ballot_w64(w != w) == ~0ull;
      kind[b] = ones ? BLK_ONES : zeros ? BLK_ZEROS : nans ? BLK_NONE : BLK_MIXED;
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {  // one query at a time: the accumulators leave few registers for anything else
      const int qq = q0 + wn * QB + q;
      // tokens past Lq and query slots past nq: weight 0 (similarity 0, nothing written)
      const float qm = (qq < a.nq && r < a.Lq) ? qm_lds[(wn * QB + q) * 32 + r] : 0.0f;
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
    const bool better = (sim_) > best; /* strict >: the first maximal token wins (torch.max); a NaN never does */      \
    best = better ? (sim_) : best;                                                                                    \
    if (AM) bidx = better ? (pos_) : bidx;                                                                            \
  } while (0)
#define AP_SCAN(PREMUL)                                                                                               \
  _Pragma("unroll") for (int b = 0; b < RW; ++b) {                                                                    \
    if (!PREMUL && kind[b] == BLK_ONES) {                                                                             \
      _Pragma("unroll") for (int v = 0; v < 16; ++v) AP_PICK(acc[b][q][v], b * 16 + v);                               \
    } else if (!PREMUL && kind[b] == BLK_ZEROS) {                                                                     \
      AP_PICK(0.0f, b * 16);                                                                                          \
    } else if (kind[b] != BLK_NONE) {                                                                                 \
      int rowbase = (wm * RW + b) * 32 + 4 * hh;                                                                      \
      /* opaque: otherwise the mask words are shared by the QB unrolled query iterations and stay live across them */ \
      asm volatile("" : "+v"(rowbase));                                                                               \
      /* d_mask of this lane's 16 rows of the block: four 16-byte reads up front, ONE wait (a read + wait per pair  */ \
      /* of rows made the epilogue cost as much as the whole K loop: 14 us per tile); NaN past Ld (tile padding) */    \
      f32x4 d4[4];                                                                                                    \
      _Pragma("unroll") for (int k = 0; k < 4; ++k) d4[k] = *(const f32x4*)(dm_lds + rowbase + 8 * k);                \
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
      if (AM) bidx = qzero ? (wm == 0 && hh == 0 ? 0 : 0x7fffffff) : (wm * RW + (bidx >> 4)) * 32 + 4 * hh + (bidx & 3) + 8 * ((bidx & 15) >> 2);
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
        ex_v[((wn * QB + q) * WM + wm) * 32 + lane] = v2;
        if (AM) ex_i[((wn * QB + q) * WM + wm) * 32 + lane] = i2;
      }
    }
    lds_barrier();
    // the row-waves' results meet: wave w finishes query slots w, w + WV, ... of the tile
#pragma unroll
    for (int qs = wave; qs < NQ; qs += WV) {
      const int qq = q0 + qs;
      float best = NEG_INF;
      int bidx = 0;
#pragma unroll
      for (int w = 0; w < WM; ++w) {  // increasing row ranges: strict > keeps the first maximal token
        const float v = ex_v[(qs * WM + w) * 32 + r];
        const bool better = v > best;
        best = better ? v : best;
        if (AM) bidx = better ? ex_i[(qs * WM + w) * 32 + r] : bidx;
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
    // (the exchange area is rewritten only after the next tile's slices, i.e. after many more barriers)
    AP_STAMP(7);
    cur = nxt;
  }
#ifdef MAXSIM_DIAG
  if ((int)blockIdx.x == a.stamp_wg) {
    __syncthreads();
    for (int i = tid; i < 2 * 24 * 9; i += WV * 64) g_ap_stamps[i] = st_lds[i];
  }
#endif
}

}  // namespace maxsim
