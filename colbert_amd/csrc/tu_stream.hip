// tu_stream.hip -- instantiations + launch heuristics of the h = 128 register-query streaming kernel (maxsim_stream.h).
#include "maxsim_launch.h"
#include "maxsim_stream.h"

namespace maxsim {
namespace {

template <int MODE, int DT, int WAVES, int NT, int ABLATE = 0, int QT = 32>
int launch_stream_v(Params& p, hipStream_t st) {
  int dpwv = MAXSIM_KNOB("MAXSIM_DPW", 0);  // tuning knob (diagnostic builds): docs per wave
  if (dpwv <= 0 || dpwv > 64) {
    dpwv = pick_docs_per_wave(p, WAVES);
    if (MODE == MODE_RERANK) dpwv = refine_docs_per_wave(p, dpwv, WAVES, WAVES * NT * StreamTraits<DT>::TILE > 80 * 1024 ? 256 : 512);
  }
  p.dpw = dpwv * WAVES;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = WAVES * NT * StreamTraits<DT>::TILE;
  if constexpr (MODE == MODE_RERANK && ABLATE == 0 && (DT == MAXSIM_F16 || DT == MAXSIM_BF16)) {
    // a ragged 16-bit index (no fixed-length promise), a workgroup's docs fitting one per lane: the token-balanced cut (BAL).
    // Measured on ragged docs N(120, 40), 256 x 1000 (interleaved, tools/run_bal_ab.sh): fp16 index 0.777 -> 0.791 of the HBM
    // peak (+1.0 .. +3.3 % in every one of six pairs), bf16 +0.5 .. +3.8 %; the fp32 index does not gain (exact: 0.629 vs 0.628,
    // bf16x3 -1 %: power-limited, an idle wave only raises the others' clock) and keeps the equal-count cut.
    // (diagnostic: MAXSIM_BAL=0 keeps the equal-count cut)
    if (p.uniform_len == 0 && p.dpw <= 64 && dpwv >= 2 && MAXSIM_KNOB("MAXSIM_BAL", 1) != 0) {
      auto kb = k_maxsim_stream<MODE, DT, WAVES, NT, ABLATE, QT, false, false, true>;
      int rc = allow_lds(kb, ldsb);
      if (rc) return rc;
      hipLaunchKernelGGL(kb, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
      return check_launch();
    }
  }
  auto kern = k_maxsim_stream<MODE, DT, WAVES, NT, ABLATE, QT>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}

template <int WAVES, int NCB, int NT>
int launch_stream_f32h(Params& p, hipStream_t st) {
  int dpwv = MAXSIM_KNOB("MAXSIM_DPW", 0);
  if (dpwv <= 0 || dpwv > 64) dpwv = pick_docs_per_wave(p, WAVES);
  p.dpw = dpwv * WAVES;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = WAVES * NT * 8192;
  auto kern = k_maxsim_stream_f32h<WAVES, NCB, NT>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}

// every doc exactly L tokens (maxsim_index_view.uniform_len): the fixed-length kernel, one 8 KiB tile per wave
template <int WAVES, int NCB, int L, int ABLATE = 0>
int launch_stream_uni(Params& p, hipStream_t st) {
  int dpwv = MAXSIM_KNOB("MAXSIM_DPW", 0);
  if (dpwv <= 0 || dpwv > 64) dpwv = pick_docs_per_wave(p, WAVES);
  p.dpw = dpwv * WAVES;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = WAVES * 8192;
  auto kern = k_maxsim_stream_uni<WAVES, NCB, L, ABLATE>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}
template <int L>
int launch_stream_uni_l(Params& p, bool many, hipStream_t st) {
#ifdef MAXSIM_DIAG
  const int variant = MAXSIM_KNOB("MAXSIM_VARIANT", 0);
  if (variant == 1) return launch_stream_uni<8, 1, L, 1>(p, st);  // no MFMA (timing only, wrong results)
  if (variant == 2) return launch_stream_uni<8, 1, L, 2>(p, st);  // no DMA  (timing only, wrong results)
#endif
  const int waves = MAXSIM_KNOB("MAXSIM_UNI_WAVES", 0);  // (diagnostic: force 4- or 8-wave workgroups)
  if (waves == 8 || (waves == 0 && many)) return p.Lq <= 16 ? launch_stream_uni<8, 1, L>(p, st) : launch_stream_uni<8, 2, L>(p, st);
  return p.Lq <= 16 ? launch_stream_uni<4, 1, L>(p, st) : launch_stream_uni<4, 2, L>(p, st);
}

// the same for a 16-bit index (the reference's storage dtype): k_maxsim_stream_uni16, NT tiles of 8 KiB per wave
template <int DT, int WAVES, int NCB, int L, int NT>
int launch_stream_uni16(Params& p, hipStream_t st) {
  int dpwv = MAXSIM_KNOB("MAXSIM_DPW", 0);
  if (dpwv <= 0 || dpwv > 64) dpwv = pick_docs_per_wave(p, WAVES);
  p.dpw = dpwv * WAVES;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = WAVES * NT * 8192;
  auto kern = k_maxsim_stream_uni16<DT, WAVES, NCB, L, NT>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}
template <int DT, int L>
int launch_stream_uni16_l(Params& p, hipStream_t st) {
  // Workgroups of 8 waves when there are enough of them to fill the chip (fewer dispatches), of 4 waves otherwise; one 8 KiB
  // tile per wave in the ring.  Measured at 256 / 2048 queries x 1000 eight-token fp16 docs (tools/probe_multiview.py): 8 x 1
  // 0.788 / 0.827 of the HBM peak, 4 x 1 0.791 / 0.830, 4 waves x 2 tiles 0.802 / 0.831 -- no shape matters; the general kernel
  // 0.511 / 0.624.  (diagnostic: MAXSIM_UNI16_SHAPE = 1 / 2 forces 8 x 1 / 4 x 1)
  const int shape = MAXSIM_KNOB("MAXSIM_UNI16_SHAPE", 0);
  const bool many = (int64_t)p.nq * ((p.ncand + 511) / 512) >= 256;
  const int pick = shape ? shape : (many ? 1 : 2);
  if (pick == 2) return p.Lq <= 16 ? launch_stream_uni16<DT, 4, 1, L, 1>(p, st) : launch_stream_uni16<DT, 4, 2, L, 1>(p, st);
  return p.Lq <= 16 ? launch_stream_uni16<DT, 8, 1, L, 1>(p, st) : launch_stream_uni16<DT, 8, 2, L, 1>(p, st);
}
// every doc of the index exactly 4 / 8 / 16 tokens and none padded (maxsim_index_view.uniform_len): that length, else 0
inline int uniform_short_len(const Params& p) {
  const int L = p.uniform_len;
  return (p.n_docs > 0 && (L == 4 || L == 8 || L == 16) && p.n_tokens == (int64_t)L * p.n_docs) ? L : 0;
}

// Per-wave LDS ring: fp32 1 x 16 KiB tile, 16-bit 2 x 8 KiB tiles; 4 waves per workgroup = 64 KiB, two
// workgroups per CU.  MAXSIM_VARIANT exists in diagnostic builds only (-DMAXSIM_DIAG; DESIGN.md "Tuning knobs"):
// 1/2 = ablation kernels (timing only, WRONG scores) -- the shipped library does not even contain them.
template <int MODE, int DT>
int launch_stream(Params& p, hipStream_t st) {
  constexpr int NT0 = (StreamTraits<DT>::TILE == 16384) ? 1 : 2;
  const int variant = MAXSIM_KNOB("MAXSIM_VARIANT", 0);
  (void)variant;
  if constexpr (MODE == MODE_RERANK && DT == MAXSIM_F32) {
    // fp32 rerank runs on v_mfma_f32_16x16x4_f32: <= 16 query tokens as one 16-column block (half the matrix work),
    // otherwise as two.  Same flop rate as the 32x32x2 form, but half the accumulator register traffic per flop:
    // the chip is power-limited on this kernel (1.9 GHz with fetch + f32 MFMA together, 2.35 GHz with either alone),
    // and the lighter form buys 2-3 % of clock.  (diagnostic: MAXSIM_VARIANT=4 forces the 32x32x2 form)
    // short docs (the 8-token multi-view config): 16-row half tiles, two per wave in the ring -- the first rows of a
    // short stream arrive sooner and tiles straddle fewer docs: +5-7 % at 8-16 tokens per doc, -2 % from 32 up.
    // (diagnostic: MAXSIM_VARIANT=6 forces this kernel, 8 disables it)
    const bool short_docs = p.n_docs > 0 && p.n_tokens <= 24 * p.n_docs;
    if (variant == 6 || (short_docs && (variant == 0 || variant == 1 || variant == 2))) {
      // Very short docs (<= 16 tokens on average; the 8-token multi-view config): ONE half tile per wave and twice the
      // waves -- 16 per CU -- instead of two tiles per wave and 8 waves.  These launches are bound by how many independent
      // doc streams are in flight, not by bytes in flight per stream: 256 queries x 1000 eight-token docs 0.200 -> 0.185 ms
      // (4-token docs 0.134 -> 0.111, 16-token docs 0.355 -> 0.351; 24-token docs 0.570 -> 0.575: those keep two tiles).
      // Workgroups of 8 waves when there are enough of them to fill the chip (fewer dispatches), of 4 waves otherwise.
      // (diagnostic: MAXSIM_F32H_SHAPE=1/2/3 forces 4 x 1 / 8 x 1 / 4 x 2)
      const int shape = MAXSIM_KNOB("MAXSIM_F32H_SHAPE", 0);
      const bool very_short = p.n_tokens <= 16 * p.n_docs;
      const bool many = (int64_t)p.nq * ((p.ncand + 511) / 512) >= 256;
      // every doc the same 4 / 8 / 16 tokens (the multi-view configuration): the kernel with the length compiled in
      // (diagnostic: MAXSIM_F32H_SHAPE != 0 keeps the general half-tile kernels)
      if (shape == 0 && variant != 6 && p.n_tokens == (int64_t)p.uniform_len * p.n_docs) {
        if (p.uniform_len == 8) return launch_stream_uni_l<8>(p, many, st);
        if (p.uniform_len == 4) return launch_stream_uni_l<4>(p, many, st);
        if (p.uniform_len == 16) return launch_stream_uni_l<16>(p, many, st);
      }
      const int pick = shape ? shape : (!very_short ? 3 : many ? 2 : 1);
      if (pick == 2) return p.Lq <= 16 ? launch_stream_f32h<8, 1, 1>(p, st) : launch_stream_f32h<8, 2, 1>(p, st);
      if (pick == 1) return p.Lq <= 16 ? launch_stream_f32h<4, 1, 1>(p, st) : launch_stream_f32h<4, 2, 1>(p, st);
      return p.Lq <= 16 ? launch_stream_f32h<4, 1, 2>(p, st) : launch_stream_f32h<4, 2, 2>(p, st);
    }
    if (p.Lq <= 16 && variant != 4) {  // (Lq <= 16 implies a single query slice)
#ifdef MAXSIM_DIAG
      if (variant == 1) return launch_stream_v<MODE, DT, 4, NT0, 1, 16>(p, st);
      if (variant == 2) return launch_stream_v<MODE, DT, 4, NT0, 2, 16>(p, st);
#endif
      return launch_stream_v<MODE, DT, 4, NT0, 0, 16>(p, st);
    }
#ifdef MAXSIM_DIAG
    if (variant == 0 || variant == 8)
#endif
      return launch_stream_v<MODE, DT, 4, NT0, 0, QT_2X16>(p, st);
  }
  if constexpr (MODE == MODE_RERANK && (DT == MAXSIM_F16 || DT == MAXSIM_BF16)) {
    // a 16-bit index whose every doc has the same 4 / 8 / 16 tokens (the reference's multi-view deployments: fp16 storage,
    // colbert_ranker.py:62, d_view viewer tokens per doc, dense.yaml:29-32): the kernel with the length compiled in
    // (diagnostic: MAXSIM_VARIANT != 0 keeps the general kernel)
    if (variant == 0) {
      switch (uniform_short_len(p)) {
        case 8: return launch_stream_uni16_l<DT, 8>(p, st);
        case 4: return launch_stream_uni16_l<DT, 4>(p, st);
        case 16: return launch_stream_uni16_l<DT, 16>(p, st);
        default: break;
      }
    }
  }
  if constexpr (MODE == MODE_RERANK && DT != MAXSIM_F32 && DT != MAXSIM_F32_FAST) {
    // 16-bit index and the 3 x bf16 mode of an fp32 index: v_mfma_f32_16x16x32 in two 16-column blocks (+2-3 %: a
    // higher sustained clock than 32x32x16 on the power cap; the fp16-split fast mode measured 1 % slower in this form
    // and keeps 32x32x16).  (diagnostic: MAXSIM_VARIANT=4 forces the 32x32x16 form)
#ifdef MAXSIM_DIAG
    if constexpr (DT == MAXSIM_F16) {   // five waves per workgroup: 2 x 80 KiB of rings fill the LDS, 10 waves per CU (3 + 3 + 2 + 2 per SIMD)
      if (variant == 0 && MAXSIM_KNOB("MAXSIM_WG_WAVES", 4) == 5) return launch_stream_v<MODE, DT, 5, NT0, 0, QT_2X16>(p, st);
    }
    if (variant == 0)
#endif
      return launch_stream_v<MODE, DT, 4, NT0, 0, QT_2X16>(p, st);
  }
#ifdef MAXSIM_DIAG
  switch (variant) {
    case 1: return launch_stream_v<MODE, DT, 4, NT0, 1>(p, st);  // no MFMA  (timing only, wrong results)
    case 2: return launch_stream_v<MODE, DT, 4, NT0, 2>(p, st);  // no DMA   (timing only, wrong results)
    case 3: return launch_stream_v<MODE, DT, 4, NT0 * 2>(p, st); // deeper ring, 1 workgroup per CU
    default: break;
  }
  if constexpr (MODE == MODE_RERANK && DT == MAXSIM_F32) {   // workgroups of 1 / 2 waves (a workgroup's slots are free only when its slowest wave ends)
    const int wgw = MAXSIM_KNOB("MAXSIM_WG_WAVES", 4);
    if (wgw == 1) return launch_stream_v<MODE, DT, 1, NT0>(p, st);
    if (wgw == 2) return launch_stream_v<MODE, DT, 2, NT0>(p, st);
  }
#endif
  return launch_stream_v<MODE, DT, 4, NT0>(p, st);
}

}  // namespace

int launch_stream_rerank(Params& p, int index_dtype, hipStream_t st) {
  switch (index_dtype) {
    case MAXSIM_F32: return launch_stream<MODE_RERANK, MAXSIM_F32>(p, st);
    case MAXSIM_F32_FAST: return launch_stream<MODE_RERANK, MAXSIM_F32_FAST>(p, st);
    case MAXSIM_F32_BF16X3: return launch_stream<MODE_RERANK, MAXSIM_F32_BF16X3>(p, st);
    case MAXSIM_F16: return launch_stream<MODE_RERANK, MAXSIM_F16>(p, st);
    default: return launch_stream<MODE_RERANK, MAXSIM_BF16>(p, st);
  }
}

int launch_stream_dense_f32(Params& p, hipStream_t st) { return launch_stream<MODE_DENSE, MAXSIM_F32>(p, st); }

// ---- counted candidate rows: the work-list form (maxsim_worklist.h) ------------------------------------------------
namespace {
template <int DT, int QT>
int launch_list_v(Params& p, int max_wgs, hipStream_t st) {
  constexpr int WAVES = 4;
  constexpr int NT = (StreamTraits<DT>::TILE == 16384) ? 1 : 2;
  const int ldsb = WAVES * NT * StreamTraits<DT>::TILE;
  auto kern = k_maxsim_stream<MODE_RERANK, DT, WAVES, NT, 0, QT, false, true>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)max_wgs), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}
}  // namespace

namespace {
template <int DT, int L>
int launch_list_uni16(Params& p, int wgs, hipStream_t st) {
  constexpr int WAVES = 4;
  const int ldsb = WAVES * 8192;
  if (p.Lq <= 16) {
    hipLaunchKernelGGL((k_maxsim_stream_uni16<DT, WAVES, 1, L, 1, true>), dim3((unsigned)wgs), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  } else {
    hipLaunchKernelGGL((k_maxsim_stream_uni16<DT, WAVES, 2, L, 1, true>), dim3((unsigned)wgs), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  }
  return check_launch();
}
template <int DT>
int launch_list_uni16_dt(Params& p, int L, int wgs, hipStream_t st) {
  if (L == 8) return launch_list_uni16<DT, 8>(p, wgs, st);
  if (L == 4) return launch_list_uni16<DT, 4>(p, wgs, st);
  return launch_list_uni16<DT, 16>(p, wgs, st);
}
template <int L>
int launch_list_uni(Params& p, int wgs, hipStream_t st) {
  constexpr int WAVES = 4;
  const int ldsb = WAVES * 8192;
  if (p.Lq <= 16) {
    hipLaunchKernelGGL((k_maxsim_stream_uni<WAVES, 1, L, 0, true>), dim3((unsigned)wgs), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  } else {
    hipLaunchKernelGGL((k_maxsim_stream_uni<WAVES, 2, L, 0, true>), dim3((unsigned)wgs), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  }
  return check_launch();
}
}  // namespace

// Does the list form serve this index?  Always: a uniform 4 / 8 / 16-token fp32 index gets the fixed-length kernel,
// everything else the general one.  (Ragged short docs on an fp32 index run 5-7 % faster on the static grid's half-tile
// kernel when every slot is live -- but counted rows come from doc shards and ANN lists, where most slots are not: one
// rank's share of an 8-way step on docs of 1..24 tokens, 2048 rows with 125 live of 1000 slots, 0.756 -> 0.411 ms with
// the fp32 index, 0.460 -> 0.180 ms with the fp16 index; tools/probe_share_short_docs.py.)
bool stream_list_serves(const Params&, int) { return true; }

int stream_list_docs_per_item(const Params& p) {
  double avg = p.n_docs > 0 ? (double)p.n_tokens / (double)p.n_docs : 1.0;
  if (avg < 1.0) avg = 1.0;
  int d = (int)(1440.0 / avg + 0.5);
  const int knob = MAXSIM_KNOB("MAXSIM_DPW", 0);
  if (knob > 0) d = knob;
  return d < 1 ? 1 : (d > 64 ? 64 : d);
}

// The grid: enough workgroups for every item the rows could hold (4 wave items each), capped -- past the cap the waves
// loop.  A few resident rounds keep the end of the launch balanced by the dispatcher without paying one dispatch per item.
int launch_stream_list(Params& p, int index_dtype, int64_t max_items, hipStream_t st) {
  const int cap = MAXSIM_KNOB("MAXSIM_LIST_WGS", 4096);
  int64_t wgs = (max_items + 3) / 4;
  if (wgs < 1) wgs = 1;
  if (wgs > cap) wgs = cap;
  wgs = (wgs + 7) & ~(int64_t)7;  // the kernel's slot -> item map assumes a workgroup's slots share s % 8
  if (const int L = uniform_short_len(p); L != 0) {  // a uniform short-doc index: the fixed-length kernels
    if (index_dtype == MAXSIM_F16) return launch_list_uni16_dt<MAXSIM_F16>(p, L, (int)wgs, st);
    if (index_dtype == MAXSIM_BF16) return launch_list_uni16_dt<MAXSIM_BF16>(p, L, (int)wgs, st);
    if (index_dtype == MAXSIM_F32) {
      if (L == 8) return launch_list_uni<8>(p, (int)wgs, st);
      if (L == 4) return launch_list_uni<4>(p, (int)wgs, st);
      return launch_list_uni<16>(p, (int)wgs, st);
    }
  }
  if (p.Lq <= 16 && index_dtype == MAXSIM_F32) return launch_list_v<MAXSIM_F32, 16>(p, (int)wgs, st);
  switch (index_dtype) {
    case MAXSIM_F32: return launch_list_v<MAXSIM_F32, QT_2X16>(p, (int)wgs, st);
    case MAXSIM_F32_FAST: return launch_list_v<MAXSIM_F32_FAST, 32>(p, (int)wgs, st);
    case MAXSIM_F32_BF16X3: return launch_list_v<MAXSIM_F32_BF16X3, QT_2X16>(p, (int)wgs, st);
    case MAXSIM_F16: return launch_list_v<MAXSIM_F16, QT_2X16>(p, (int)wgs, st);
    case MAXSIM_BF16: return launch_list_v<MAXSIM_BF16, QT_2X16>(p, (int)wgs, st);
    default: return MAXSIM_EINVAL;
  }
}

}  // namespace maxsim
