// libmaxsim.so -- MI355X (gfx950 / CDNA4) kernels + C ABI for the ColBERT MaxSim rerank path.
//
// What is computed (reference: wuyaoxuehun/colbert, colbert/modeling/BaseModel.py:39-46 and
// colbert/ranking/colbert_ranker.py:88-118):
//     score(q, doc) = sum_m  max_n  <Q[q,m,:], D[doc,n,:]>
// Kernels: maxsim_stream.h / maxsim_stream_bigh.h (MFMA + LDS-DMA streaming kernels, the hot path; compiled in the
// tu_*.hip units so that they build in parallel), maxsim_generic.h (any-shape correctness kernel), maxsim_topk.h,
// maxsim_candidates.h, maxsim_backward.h.  This file holds the dispatch and the C ABI declared in include/maxsim.h.
// gfx950 only: no CUDA, no hipify, no dual paths.
#include "maxsim_backward.h"
#include "maxsim_candidates.h"
#include "maxsim_common.h"
#include "maxsim_generic.h"
#include "maxsim_launch.h"
#include "maxsim_probe.h"
#include "maxsim_shard.h"
#include "maxsim_topk.h"
#include "maxsim_worklist.h"

using namespace maxsim;

namespace {

// one spin of a host-side polling loop (maxsim_rank_forward waits for the top-k kernel's completion word)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}

// Queries longer than 32 tokens: score(Q) = sum over 32-token slices of score(slice) (the sum over query tokens
// is additive), one launch per slice, the later ones accumulating into `scores`.
template <typename F>
int for_query_slices(Params& p, F&& launch) {
  for (int t0 = 0; t0 < p.Lq; t0 += 32) {
    p.q_tok0 = t0;
    p.accum = t0 > 0;
    int rc = launch();
    if (rc != MAXSIM_OK) return rc;
  }
  p.q_tok0 = 0;
  p.accum = 0;
  return MAXSIM_OK;
}
constexpr int MAX_LQ_SLICED = 1024;
#ifdef MAXSIM_STAMP
const void* g_stamp_buffer = nullptr;  // timing builds: where the stream kernel's waves write their phase stamps
#endif

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
int64_t maxsim_index_view_bytes(void) { return (int64_t)sizeof(maxsim_index_view); }

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

static int score_dense_impl(const void* Q, const void* D, const void* q_mask, const void* d_mask, int nq, int nd,
                            int Lq, int Ld, int h, int dtype, int mask_dtype, float* out, int32_t* argmax,
                            void* stream) {
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
  p.q_dtype = dtype;
  p.nq = nq; p.ncand = nd; p.Lq = Lq; p.h = h;
  p.scores = out;
  p.argmax = argmax;
  p.q_mask = q_mask; p.d_mask = d_mask; p.mask_dtype = mask_dtype;
  p.Ld = Ld;
  // the streaming kernels move 16-byte pieces: operands that are not 16-byte aligned take the generic kernel
  const bool aligned = (((uintptr_t)Q | (uintptr_t)D) & 15) == 0;
  const bool stream_ok = aligned && (Lq <= 32 || (!argmax && Lq <= MAX_LQ_SLICED)) && p.n_tokens <= 0xffffffffLL;
  if (!argmax && stream_ok && dtype == MAXSIM_F32 && h == 128)
    return for_query_slices(p, [&] { return launch_stream_dense_f32(p, st); });
  if (dtype != MAXSIM_F32 && aligned && Lq <= 32) {  // big all-pairs problems (the training step): GEMM blocking
    const int rc = launch_allpairs(p, dtype, argmax != nullptr, st);
    if (rc != MAXSIM_ERANGE) return rc;
  }
  const int esz = dtype == MAXSIM_F32 ? 4 : 2;
  if (stream_ok && h >= 16 && h <= 1024 && ((h * esz) & 15) == 0) {
    int rc = argmax ? launch_bigh_dense(p, dtype, true, st)
                    : for_query_slices(p, [&] { return launch_bigh_dense(p, dtype, false, st); });
    if (rc != MAXSIM_ERANGE) return rc;
  }
  return launch_generic<MODE_DENSE>(p, dtype, st);
}

int maxsim_score_dense_kernel(int nq, int nd, int Lq, int Ld, int h, int dtype, int mask_dtype) {
  if (nq < 0 || nd < 0 || Lq < 1 || Ld < 1 || h < 1 || dtype < MAXSIM_F32 || dtype > MAXSIM_BF16) return MAXSIM_EINVAL;
  return allpairs_serves(dtype, dtype, mask_dtype, nq, nd, Lq, Ld, h) ? 1 : 0;
}

int maxsim_score_dense(const void* Q, const void* D, const void* q_mask, const void* d_mask, int nq, int nd,
                       int Lq, int Ld, int h, int dtype, int mask_dtype, float* out, void* stream) {
  return score_dense_impl(Q, D, q_mask, d_mask, nq, nd, Lq, Ld, h, dtype, mask_dtype, out, nullptr, stream);
}

int maxsim_score_dense_fwd(const void* Q, const void* D, const void* q_mask, const void* d_mask, int nq, int nd,
                           int Lq, int Ld, int h, int dtype, int mask_dtype, float* out, int32_t* argmax,
                           void* stream) {
  if (!argmax && (int64_t)nq * nd * Lq > 0) return MAXSIM_EINVAL;
  return score_dense_impl(Q, D, q_mask, d_mask, nq, nd, Lq, Ld, h, dtype, mask_dtype, out, argmax, stream);
}

int64_t maxsim_score_dense_bwd_workspace(int nq, int nd, int Lq, int Ld) {
  if (nq < 0 || nd < 0 || Lq < 0 || Ld < 0) return 0;
  return ((int64_t)nd * ((int64_t)nq * Lq) + (int64_t)nd * (Ld + 1)) * (int64_t)sizeof(int32_t);
}

int maxsim_score_dense_bwd(const void* Q, const void* D, const void* q_mask, const void* d_mask,
                           const int32_t* argmax, const float* grad_out, int nq, int nd, int Lq, int Ld, int h,
                           int dtype, int mask_dtype, float* dQ, float* dD, void* workspace, int64_t workspace_bytes,
                           void* stream) {
  if (nq < 0 || nd < 0 || Lq < 0 || Ld < 0 || h < 0) return MAXSIM_EINVAL;
  if (dtype < MAXSIM_F32 || dtype > MAXSIM_BF16) return MAXSIM_EINVAL;
  if (mask_dtype < MAXSIM_MASK_NONE || mask_dtype > MAXSIM_MASK_U8) return MAXSIM_EINVAL;
  if (h > 1024) return MAXSIM_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  const int64_t nQ = (int64_t)nq * Lq * h, nD = (int64_t)nd * Ld * h;
  const bool degenerate = nq == 0 || nd == 0 || Lq == 0 || Ld == 0 || h == 0;
  const bool dd_lds = Ld * 256 <= 150 * 1024;  // the doc's [Ld][64] fp32 slab fits in LDS
  // preferred: per-doc inverse index in the caller's workspace (then dD is fully overwritten row by row)
  // per-wave histograms + ranking scratch + bucket starts + the doc's keys parked as 16-bit values
  const int64_t idx_lds = (int64_t)(2 * BWD_INDEX_WAVES + 1) * ((int64_t)Ld + 1) * 4 +
                          ((int64_t)nq * Lq <= BWD_INDEX_MAX_PARKED ? (((int64_t)nq * Lq * 2 + 15) & ~15LL) : 0);
  const bool vec8 = (h & 7) == 0;                                            // 16-byte row chunks
  const bool dd_idx = vec8 && workspace && workspace_bytes >= maxsim_score_dense_bwd_workspace(nq, nd, Lq, Ld) &&
                      idx_lds <= 150 * 1024 && Ld <= 1024 && ((int64_t)nd * Ld + 3) / 4 <= 0x7fffffffLL;
  if (dQ && nQ > 0 && degenerate && hipMemsetAsync(dQ, 0, nQ * sizeof(float), st) != hipSuccess) return MAXSIM_ELAUNCH;
  if (dD && nD > 0 && (degenerate || (!dd_lds && !dd_idx)) && hipMemsetAsync(dD, 0, nD * sizeof(float), st) != hipSuccess)
    return MAXSIM_ELAUNCH;
  if (degenerate) return MAXSIM_OK;
  if (!Q || !D || !argmax || !grad_out) return MAXSIM_EINVAL;
  if (mask_dtype != MAXSIM_MASK_NONE && (!q_mask || !d_mask)) return MAXSIM_EINVAL;
  if ((int64_t)nq * nd > 0x7fffffffLL || (int64_t)nq * Lq > 0x7fffffffLL) return MAXSIM_ERANGE;
  const int nchunk = (h + 63) / 64;
  if (dD && dd_lds && (int64_t)nd * nchunk > 0x7fffffffLL) return MAXSIM_ERANGE;
  int rc = MAXSIM_OK;
#define BWD_LAUNCH(DT)                                                                                              \
  do {                                                                                                              \
    if (dQ && vec8)                                                                                                 \
      hipLaunchKernelGGL((k_maxsim_bwd_dq_v8<DT>), dim3((unsigned)(nq * Lq)), dim3(128), 0, st, D, q_mask, d_mask,  \
                         mask_dtype, argmax, grad_out, dQ, nd, Lq, Ld, h);                                          \
    else if (dQ)                                                                                                    \
      hipLaunchKernelGGL((k_maxsim_bwd_dq<DT>), dim3((unsigned)(nq * Lq)), dim3(256), 0, st, D, q_mask, d_mask,     \
                         mask_dtype, argmax, grad_out, dQ, nd, Lq, Ld, h);                                          \
    if (dD && dd_idx) {                                                                                             \
      int32_t* ws_start = (int32_t*)workspace;                                                                      \
      int32_t* ws_items = ws_start + (int64_t)nd * (Ld + 1);                                                        \
      const int lb = (int)idx_lds;                                                                                  \
      rc = allow_lds(k_maxsim_bwd_index, lb);                                                                       \
      if (rc == MAXSIM_OK) {                                                                                        \
        hipLaunchKernelGGL(k_maxsim_bwd_index, dim3((unsigned)nd), dim3(64 * BWD_INDEX_WAVES), lb, st, q_mask, d_mask, mask_dtype,  \
                           argmax, grad_out, ws_start, ws_items, nq, nd, Lq, Ld);                                   \
        hipLaunchKernelGGL((k_maxsim_bwd_dd_rows<DT>), dim3((unsigned)(((int64_t)nd * Ld + 3) / 4)), dim3(256), 0,  \
                           st, Q, q_mask, d_mask, mask_dtype, grad_out, ws_start, ws_items, dD, nq, nd, Lq, Ld, h); \
      }                                                                                                             \
    } else if (dD && dd_lds) {                                                                                      \
      rc = allow_lds(k_maxsim_bwd_dd_lds<DT>, Ld * 256);                                                            \
      if (rc == MAXSIM_OK)                                                                                          \
        hipLaunchKernelGGL((k_maxsim_bwd_dd_lds<DT>), dim3((unsigned)(nd * nchunk)), dim3(1024), Ld * 256, st, Q,   \
                           q_mask, d_mask, mask_dtype, argmax, grad_out, dD, nq, nd, Lq, Ld, h, nchunk);            \
    } else if (dD) {                                                                                                \
      hipLaunchKernelGGL((k_maxsim_bwd_dd_atomic<DT>), dim3((unsigned)((int64_t)nq * nd)), dim3(256), 0, st, Q,     \
                         q_mask, d_mask, mask_dtype, argmax, grad_out, dD, nd, Lq, Ld, h);                          \
    }                                                                                                               \
  } while (0)
  switch (dtype) {
    case MAXSIM_F32: BWD_LAUNCH(MAXSIM_F32); break;
    case MAXSIM_F16: BWD_LAUNCH(MAXSIM_F16); break;
    default: BWD_LAUNCH(MAXSIM_BF16); break;
  }
#undef BWD_LAUNCH
  if (rc) return rc;
  return check_launch();
}

int64_t maxsim_worklist_bytes(int nq, int ncand) {
  if (nq < 0 || ncand < 0) return 0;
  // header + item_start[nq + 1] + at most one wave item per candidate slot
  return (worklist_items_word(nq) + 2 * ((int64_t)nq * ncand + 1)) * (int64_t)sizeof(int32_t);
}

constexpr int LIST_FUSED_FILL_NQ = 8;  // up to this many rows the scan kernel also fills (one launch instead of two)
constexpr int LIST_MIN_ITEMS = 1792;  // 448 workgroups of 4 waves: pick_docs_per_wave's rule, applied on the device

static int rerank_impl(const maxsim_index_view& iv, const void* Q, int q_dtype, const int32_t* q_len,
                       const uint8_t* q_mask, const int64_t* cand_pids, int nq, int ncand, int Lq, float* scores,
                       hipStream_t st, const int32_t* cand_count = nullptr, void* worklist = nullptr,
                       int64_t worklist_bytes = 0) {
  const int h = iv.h, index_dtype = iv.index_dtype;
  const int64_t n_tokens = iv.n_tokens, n_docs = iv.n_docs;
  if (nq < 0 || ncand < 0 || Lq < 0 || h < 0 || n_tokens < 0 || n_docs < 0) return MAXSIM_EINVAL;
  if (index_dtype < MAXSIM_F32 || index_dtype > MAXSIM_F32_BF16X3) return MAXSIM_EINVAL;
  if (q_dtype < MAXSIM_F32 || q_dtype > MAXSIM_BF16) return MAXSIM_EINVAL;
  if (ncand == 0) return MAXSIM_EEMPTY;  // assert len(pids) > 0, colbert_ranker.py:76
  if (nq == 0) return MAXSIM_OK;
  if (!scores || !cand_pids || !Q) return MAXSIM_EINVAL;
  if (!iv.doc_table && (!iv.tok_offsets || !iv.doclens)) return MAXSIM_EINVAL;
  if (((uintptr_t)iv.doc_table & 15) != 0) return MAXSIM_EINVAL;
  if (n_tokens > 0 && !iv.index) return MAXSIM_EINVAL;
  if ((int64_t)nq * ncand > 0x7fffffffLL) return MAXSIM_ERANGE;
  Params p{};
  p.index = iv.index;
  p.n_tokens = n_tokens;
  p.tok_offsets = iv.tok_offsets;
  p.doclens = iv.doclens;
  p.pad_len = iv.pad_len;
  p.doc_table = iv.doc_table;
  p.uniform_len = iv.uniform_len;
  p.n_docs = n_docs;
  p.Q = Q;
  p.q_dtype = q_dtype;
  p.q_len = q_len;
  p.q_mask = q_mask;  // rerank mode: uint8 keep-predicate (maxsim_common.h q_token_live)
  p.cand = cand_pids;
  p.nq = nq; p.ncand = ncand; p.Lq = Lq; p.h = h;
  p.scores = scores;
  p.mask_dtype = MAXSIM_MASK_NONE;
#ifdef MAXSIM_STAMP
  p.d_mask = g_stamp_buffer;  // timing builds: per-wave phase stamps (maxsim_stream.h)
#endif
  const bool aligned = (((uintptr_t)Q | (uintptr_t)iv.index) & 15) == 0;  // the streaming kernels move 16-byte pieces
  const bool stream_ok = aligned && Lq >= 1 && Lq <= MAX_LQ_SLICED && n_tokens > 0 && n_tokens <= 0xffffffffLL;
  // counted rows (doc shards, ANN lists): the device builds a dense list of wave items and a fixed grid walks it (every
  // shape without a streaming kernel for h = 128 keeps the static grid).
  if (h == 128 && stream_ok && cand_count && worklist && ((uintptr_t)worklist & 15) == 0 && stream_list_serves(p, index_dtype) &&
      worklist_bytes >= maxsim_worklist_bytes(nq, ncand) && ncand < (1 << WL_SLOT_BITS)) {
    p.worklist = worklist;
    const int D0 = stream_list_docs_per_item(p);
    int32_t* const wl = (int32_t*)worklist;
    // (wave slots the list kernel keeps resident: 2 workgroups of 4 waves per CU; the fixed-length short-doc kernel 4)
    const bool uni_short = (index_dtype == MAXSIM_F32 || index_dtype == MAXSIM_F16 || index_dtype == MAXSIM_BF16) &&
                           p.n_tokens == (int64_t)p.uniform_len * p.n_docs && (p.uniform_len == 4 || p.uniform_len == 8 || p.uniform_len == 16);
    const int slots_knob = MAXSIM_KNOB("MAXSIM_LIST_SLOTS", -1);  // (diagnostic builds: 0 switches the one-round rule off)
    const int list_slots = slots_knob >= 0 ? slots_knob : (uni_short ? 4096 : 2048);
    if (nq <= LIST_FUSED_FILL_NQ) {  // a handful of rows: scan + fill in one launch
      hipLaunchKernelGGL(k_worklist_scan, dim3(1), dim3(1024), 0, st, cand_count, nq, ncand, D0, 64, LIST_MIN_ITEMS, list_slots, wl, scores);
    } else {
      hipLaunchKernelGGL(k_worklist_scan, dim3(1), dim3(1024), 0, st, cand_count, nq, ncand, D0, 64, LIST_MIN_ITEMS, list_slots, wl,
                         (float*)nullptr);
      hipLaunchKernelGGL(k_worklist_fill, dim3((unsigned)nq), dim3(256), 0, st, cand_count, nq, ncand, wl, scores, 1);
    }
    if (check_launch() != MAXSIM_OK) return MAXSIM_ELAUNCH;
    // k_worklist_scan may settle on any docs-per-item in [D0 / 2, 2 D0] (and keeps halving while the launch is below its
    // minimum): the grid is sized for the SMALLEST it can pick, so that no workgroup runs two items while CU slots idle
    const int Dlow = D0 / 2 > 1 ? D0 / 2 : 1;
    const int64_t by_rows = (int64_t)nq * ((ncand + Dlow - 1) / Dlow), small = 2 * LIST_MIN_ITEMS + nq;
    const int64_t max_items = by_rows > small ? by_rows : small;
    return for_query_slices(p, [&] { return launch_stream_list(p, index_dtype, max_items, st); });
  }
  if (h == 128 && stream_ok) {
    if (Lq <= 32) {  // small launches of a 16-bit index: docs split over several waves
      const int rc = launch_stream_small(p, index_dtype, st);
      if (rc != MAXSIM_ERANGE) return rc;
    }
    return for_query_slices(p, [&] { return launch_stream_rerank(p, index_dtype, st); });
  }
  const int esz = (index_dtype == MAXSIM_F32 || index_dtype >= MAXSIM_F32_FAST) ? 4 : 2;
  if (h >= 16 && h <= 1024 && ((h * esz) & 15) == 0 && stream_ok) {
    const int dt = index_dtype >= MAXSIM_F32_FAST ? MAXSIM_F32 : index_dtype;
    // counted rows on wide embeddings (the reference's default dim 768): workgroup items -- the waves of a workgroup share
    // the staged query image -- of waves x (a ~1.4 k-token stream per wave) docs
    const int lw = (cand_count && worklist && ((uintptr_t)worklist & 15) == 0 && worklist_bytes >= maxsim_worklist_bytes(nq, ncand) &&
                    ncand < (1 << WL_SLOT_BITS)) ? bigh_list_waves(p, dt) : 0;
    if (lw > 0) {
      p.worklist = worklist;
      const int D0 = stream_list_docs_per_item(p) * lw;
      int32_t* const wl = (int32_t*)worklist;
      const int slots768 = MAXSIM_KNOB("MAXSIM_LIST_SLOTS", -1) >= 0 ? MAXSIM_KNOB("MAXSIM_LIST_SLOTS", -1) : 256;  // (one workgroup per CU is resident)
      if (nq <= LIST_FUSED_FILL_NQ) {
        hipLaunchKernelGGL(k_worklist_scan, dim3(1), dim3(1024), 0, st, cand_count, nq, ncand, D0, 64 * lw, 256, slots768, wl, scores);
      } else {
        hipLaunchKernelGGL(k_worklist_scan, dim3(1), dim3(1024), 0, st, cand_count, nq, ncand, D0, 64 * lw, 256, slots768, wl,
                           (float*)nullptr);
        hipLaunchKernelGGL(k_worklist_fill, dim3((unsigned)nq), dim3(256), 0, st, cand_count, nq, ncand, wl, scores, 1);
      }
      if (check_launch() != MAXSIM_OK) return MAXSIM_ELAUNCH;
      const int Dlow = D0 / 2 > 1 ? D0 / 2 : 1;   // (the smallest docs-per-item the scan can settle on, as above)
      const int64_t by_rows = (int64_t)nq * ((ncand + Dlow - 1) / Dlow), small = 2 * 256 + nq;
      const int64_t max_items = by_rows > small ? by_rows : small;
      int rc = for_query_slices(p, [&] { return launch_bigh_rerank_list(p, dt, max_items, st); });
      if (rc != MAXSIM_ERANGE) return rc;
      p.worklist = nullptr;  // (not reached: bigh_list_waves said the form serves this launch)
    }
    if (Lq <= 32) {  // small launches (the online call): docs split over several waves
      const int rc = launch_bigh_rerank_small(p, dt, st);
      if (rc != MAXSIM_ERANGE) return rc;
    }
    int rc = for_query_slices(p, [&] { return launch_bigh_rerank(p, dt, st); });
    if (rc != MAXSIM_ERANGE) return rc;
  }
  return launch_generic<MODE_RERANK>(p, index_dtype >= MAXSIM_F32_FAST ? MAXSIM_F32 : index_dtype, st);
}

int maxsim_rerank(const void* index, int index_dtype, int64_t n_tokens, const int64_t* tok_offsets,
                  const int32_t* doclens, const int32_t* pad_len, int64_t n_docs, const void* Q, int q_dtype,
                  const int32_t* q_len, const int64_t* cand_pids, int nq, int ncand, int Lq, int h,
                  float* scores, void* stream) {
  maxsim_index_view iv{};
  iv.index = index;
  iv.index_dtype = index_dtype;
  iv.h = h;
  iv.n_tokens = n_tokens;
  iv.tok_offsets = tok_offsets;
  iv.doclens = doclens;
  iv.pad_len = pad_len;
  iv.n_docs = n_docs;
  return rerank_impl(iv, Q, q_dtype, q_len, nullptr, cand_pids, nq, ncand, Lq, scores, (hipStream_t)stream);
}

int maxsim_rerank_ex(const maxsim_index_view* iv, const void* Q, int q_dtype, const int32_t* q_len,
                     const uint8_t* q_mask, const int64_t* cand_pids, int nq, int ncand, int Lq, float* scores,
                     void* stream) {
  if (!iv || (iv->struct_size != 0 && iv->struct_size != (int32_t)sizeof(maxsim_index_view))) return MAXSIM_EINVAL;
  return rerank_impl(*iv, Q, q_dtype, q_len, q_mask, cand_pids, nq, ncand, Lq, scores, (hipStream_t)stream);
}

#ifdef MAXSIM_STAMP
void maxsim_diag_set_stamp_buffer(const void* p) { g_stamp_buffer = p; }
#endif

int maxsim_rerank_counted(const maxsim_index_view* iv, const void* Q, int q_dtype, const int32_t* q_len,
                          const uint8_t* q_mask, const int64_t* cand_pids, const int32_t* cand_count, int nq, int ncand,
                          int Lq, float* scores, void* worklist, int64_t worklist_bytes, void* stream) {
  if (!iv || (iv->struct_size != 0 && iv->struct_size != (int32_t)sizeof(maxsim_index_view))) return MAXSIM_EINVAL;
  return rerank_impl(*iv, Q, q_dtype, q_len, q_mask, cand_pids, nq, ncand, Lq, scores, (hipStream_t)stream, cand_count,
                     worklist, worklist_bytes);
}

int64_t maxsim_doc_table_bytes(int64_t n_docs) { return n_docs > 0 ? n_docs * 16 : 0; }

int maxsim_build_doc_table(const int64_t* tok_offsets, const int32_t* doclens, const int32_t* pad_len,
                           int64_t n_docs, void* table, void* stream) {
  if (n_docs < 0) return MAXSIM_EINVAL;
  if (n_docs == 0) return MAXSIM_OK;
  if (!tok_offsets || !doclens || !table || ((uintptr_t)table & 15) != 0) return MAXSIM_EINVAL;
  if ((n_docs + 255) / 256 > 0x7fffffffLL) return MAXSIM_ERANGE;
  hipLaunchKernelGGL(k_build_doc_table, dim3((unsigned)((n_docs + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     tok_offsets, doclens, pad_len, n_docs, (int4*)table);
  return check_launch();
}

int maxsim_shard_candidates(const int64_t* cand_global, int nq, int ncand, int64_t lo, int64_t hi,
                            int64_t* out_local, int64_t* out_global, int32_t* out_count, void* stream) {
  if (nq < 0 || ncand < 0 || lo > hi) return MAXSIM_EINVAL;
  if (nq == 0 || ncand == 0) return MAXSIM_OK;
  if (!cand_global || !out_local) return MAXSIM_EINVAL;
  hipLaunchKernelGGL(k_shard_candidates, dim3((unsigned)nq), dim3(256), 0, (hipStream_t)stream, cand_global, ncand, lo,
                     hi, out_local, out_global, out_count);
  return check_launch();
}

// short lists: rank by counting, ncand / 16 workgroups per query (maxsim_topk.h)
static int topk_count(const float* scores, const int64_t* pids, int nq, int ncand, int k, float* out_scores,
                      int64_t* out_pids, int32_t* counter, uint32_t* done_flag, uint32_t ticket, hipStream_t st,
                      const int32_t* counts = nullptr) {
  int groups = (ncand + TOPK_CAND_PER_WG - 1) / TOPK_CAND_PER_WG;
  // counted rows of a big batch: at most 8 workgroups per row, each looping over its candidate groups (see k_topk_count)
  if (counts && !done_flag && nq >= 64 && groups > 8) groups = 8;
  if ((int64_t)nq * groups > 0x7fffffffLL) return MAXSIM_ERANGE;
  hipLaunchKernelGGL(k_topk_count, dim3((unsigned)(nq * groups)), dim3(256), 0, st, scores, pids, ncand, k, out_scores,
                     out_pids, groups, counter, done_flag, ticket, counts, groups);
  return check_launch();
}

static int topk_impl(const float* scores, const int64_t* pids, const int32_t* counts, int nq, int ncand, int k,
                     float* out_scores, int64_t* out_pids, void* stream);

int maxsim_topk(const float* scores, const int64_t* pids, int nq, int ncand, int k, float* out_scores,
                int64_t* out_pids, void* stream) {
  return topk_impl(scores, pids, nullptr, nq, ncand, k, out_scores, out_pids, stream);
}

int maxsim_topk_counted(const float* scores, const int64_t* pids, const int32_t* counts, int nq, int ncand, int k,
                        float* out_scores, int64_t* out_pids, void* stream) {
  return topk_impl(scores, pids, counts, nq, ncand, k, out_scores, out_pids, stream);
}

static int topk_impl(const float* scores, const int64_t* pids, const int32_t* counts, int nq, int ncand, int k,
                     float* out_scores, int64_t* out_pids, void* stream) {
  if (nq < 0 || ncand < 0 || k < 1) return MAXSIM_EINVAL;
  if (ncand == 0) return MAXSIM_EEMPTY;
  if (ncand > 16384) return MAXSIM_ERANGE;
  if (nq == 0) return MAXSIM_OK;
  if (!scores || !out_scores || !out_pids) return MAXSIM_EINVAL;
  if (ncand <= 2048)
    return topk_count(scores, pids, nq, ncand, k, out_scores, out_pids, nullptr, nullptr, 0, (hipStream_t)stream, counts);
  int P = 2;
  while (P < ncand) P <<= 1;
  const int ldsb = P * 8;
  int rc = allow_lds(k_topk, ldsb);
  if (rc) return rc;
  int threads = P / 2 < 64 ? 64 : (P / 2 > 1024 ? 1024 : P / 2);
  hipLaunchKernelGGL(k_topk, dim3((unsigned)nq), dim3(threads), ldsb, (hipStream_t)stream, scores, pids, ncand, k,
                     P, out_scores, out_pids, counts);
  return check_launch();
}

int64_t maxsim_rank_forward_workspace_bytes(int n) { return n > 0 ? (int64_t)n * 4 + 64 : 64; }

int maxsim_rank_forward(const maxsim_index_view* iv, const void* Q, int q_dtype, int Lq, const int64_t* pids, int n,
                        int depth, void* workspace, int64_t* out_pids, float* out_scores, uint32_t* done_flag,
                        int sync, void* stream) {
  if (!iv || n < 0 || depth < 1 || (iv->struct_size != 0 && iv->struct_size != (int32_t)sizeof(maxsim_index_view))) return MAXSIM_EINVAL;
  if (n == 0) return MAXSIM_EEMPTY;  // assert len(pids) > 0, colbert_ranker.py:76
  if (n > 16384) return MAXSIM_ERANGE;
  if (!workspace || ((uintptr_t)workspace & 15) != 0 || !out_pids || !out_scores) return MAXSIM_EINVAL;
  // workspace: 64 bytes of counters (zero between calls) | n floats (the score vector, colbert_ranker.py:122)
  int32_t* const counter = (int32_t*)workspace;
  float* const scores = (float*)((char*)workspace + 64);
  hipStream_t st = (hipStream_t)stream;
  const int k = depth < n ? depth : n;
  int rc = rerank_impl(*iv, Q, q_dtype, nullptr, nullptr, pids, 1, n, Lq, scores, st);
  if (rc != MAXSIM_OK) return rc;
  // colbert_ranker.py:128-130.  Short lists: the counting kernel, whose last workgroup stores a ticket to done_flag.
  // (Running that ranking at the END OF THE RERANK LAUNCH instead -- its first n / 16 workgroups waiting in-kernel for the
  //  launch's scores -- was built and measured: the GPU span of a call grew by 2-4 us and the call took as long as with
  //  two launches, tools/attic/README.md "fused top-k"; the second launch overlaps the first kernel and costs nothing.)
  const bool poll = sync && done_flag && n <= 2048 && MAXSIM_KNOB("MAXSIM_POLL", 1) != 0;
  static std::atomic<uint32_t> tickets{0};
  uint32_t ticket = ++tickets;
  if (ticket == 0) ticket = ++tickets;
  if (n <= 2048)
    rc = topk_count(scores, pids, 1, n, k, out_scores, out_pids, counter, poll ? done_flag : nullptr, ticket, st);
  else
    rc = maxsim_topk(scores, pids, 1, n, k, out_scores, out_pids, stream);
  if (rc != MAXSIM_OK) return rc;
  if (!sync) return MAXSIM_OK;
  if (poll) {
    // polling the host-visible word costs a fraction of a stream synchronisation.  Bounded: if the ticket does not
    // show up (the memory is not host-coherent after all), fall back to the runtime's wait.
    volatile uint32_t* f = done_flag;
    for (int spins = 0; spins < (1 << 22); ++spins) {
      if (*f == ticket) {
        std::atomic_thread_fence(std::memory_order_acquire);
        return MAXSIM_OK;
      }
      cpu_relax();
    }
  }
  return hipStreamSynchronize(st) == hipSuccess ? MAXSIM_OK : MAXSIM_ELAUNCH;
}

void* maxsim_host_alloc_coherent(int64_t bytes) {
  void* p = nullptr;
  if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocCoherent) != hipSuccess) return nullptr;
  return p;
}

void maxsim_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int maxsim_hbm_read_probe_scattered(const void* buf, int64_t bytes, int granule, int variant, int64_t read_bytes, void* stream) {
  if (!buf || bytes < 0 || read_bytes < 0 || ((uintptr_t)buf & 15) != 0) return MAXSIM_EINVAL;
  if (granule < 1024 || (granule & (granule - 1)) != 0 || granule > (1 << 20) || variant < 0 || variant > 3) return MAXSIM_EINVAL;
  const int64_t per_wave = 256 * 1024;
  int64_t ngran = 1;  // granules of the buffer, rounded down to a power of two (the kernel masks a hash)
  while (2 * ngran * granule <= bytes && ngran < (1LL << 31)) ngran *= 2;
  const int64_t wgs = read_bytes / (4 * per_wave);
  if (wgs == 0 || ngran * granule > bytes) return MAXSIM_OK;
  if (wgs > 0x7fffffffLL) return MAXSIM_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL((k_read_probe_scatter<16384, 1>), dim3((unsigned)wgs), dim3(256), 4 * 16384, st, (const char*)buf, per_wave, ngran, granule); break;
    case 1: hipLaunchKernelGGL((k_read_probe_scatter<8192, 2>), dim3((unsigned)wgs), dim3(256), 4 * 2 * 8192, st, (const char*)buf, per_wave, ngran, granule); break;
    case 2: hipLaunchKernelGGL((k_read_probe_scatter<8192, 1>), dim3((unsigned)wgs), dim3(256), 4 * 8192, st, (const char*)buf, per_wave, ngran, granule); break;
    default: hipLaunchKernelGGL((k_read_probe_scatter<8192, 1>), dim3((unsigned)(wgs / 2)), dim3(256), 4 * 8192, st, (const char*)buf, 2 * per_wave, ngran, granule); break;
  }
  return check_launch();
}

int maxsim_hbm_read_probe(const void* buf, int64_t bytes, int variant, int64_t* bytes_read, void* stream) {
  if (!buf || bytes < 0 || variant < 0 || variant > 2 || ((uintptr_t)buf & 15) != 0) return MAXSIM_EINVAL;
  // a wave reads 32 tiles of 16 KiB (variant 1: 64 tiles of 8 KiB) = 512 KiB, a workgroup of 4 waves 2 MiB
  const int64_t per_wave = 512 * 1024;
  const int64_t wgs = bytes / (4 * per_wave);
  if (bytes_read) *bytes_read = wgs * 4 * per_wave;
  if (wgs == 0) return MAXSIM_OK;
  if (wgs > 0x7fffffffLL) return MAXSIM_ERANGE;
  hipStream_t st = (hipStream_t)stream;
  int rc = MAXSIM_OK;
  switch (variant) {
    case 0:  // the fp32 rerank kernel's shape: one 16 KiB tile per wave, 8 waves per CU
      hipLaunchKernelGGL((k_read_probe<16384, 1>), dim3((unsigned)wgs), dim3(256), 4 * 16384, st, (const char*)buf, per_wave);
      break;
    case 1:  // the 16-bit kernels' shape: two 8 KiB tiles per wave
      hipLaunchKernelGGL((k_read_probe<8192, 2>), dim3((unsigned)wgs), dim3(256), 4 * 2 * 8192, st, (const char*)buf, per_wave);
      break;
    default:  // two 16 KiB tiles per wave, one workgroup per CU
      rc = allow_lds(k_read_probe<16384, 2>, 4 * 2 * 16384);
      if (rc) return rc;
      hipLaunchKernelGGL((k_read_probe<16384, 2>), dim3((unsigned)wgs), dim3(256), 4 * 2 * 16384, st, (const char*)buf, per_wave);
      break;
  }
  return check_launch();
}

int64_t maxsim_row_blocks_bytes(int64_t n_tokens) {
  if (n_tokens < 0) return MAXSIM_EINVAL;
  return (((n_tokens + (1 << kRowBlockShift) - 1) >> kRowBlockShift) + 1) * 8;
}

int maxsim_build_row_blocks(const int64_t* tok_offsets, int64_t n_docs, int64_t n_tokens, void* row_blocks, void* stream) {
  if (n_docs < 0 || n_tokens < 0 || !row_blocks || (n_docs > 0 && !tok_offsets)) return MAXSIM_EINVAL;
  if (n_docs > 0xfffffffeLL) return MAXSIM_ERANGE;
  const int64_t nblocks = (n_tokens + (1 << kRowBlockShift) - 1) >> kRowBlockShift;
  const int64_t wgs = (nblocks + 1 + 255) / 256;
  if (wgs > 0x7fffffffLL) return MAXSIM_ERANGE;
  if (((uintptr_t)row_blocks & 7) != 0) return MAXSIM_EINVAL;
  hipLaunchKernelGGL(k_build_row_blocks, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, tok_offsets, n_docs, n_tokens,
                     nblocks, (uint64_t*)row_blocks);
  return check_launch();
}

int maxsim_embedding_ids_to_pids_ex(const int64_t* emb_ids, int nq, int n, int ids_per_token, const uint8_t* tok_keep,
                                    int64_t id_base, const int64_t* tok_offsets, int64_t n_docs, int64_t n_tokens,
                                    const void* row_blocks, int64_t* out_pids, int32_t* out_count, void* stream) {
  if (nq < 0 || n < 0 || n_docs < 0 || n_tokens < 0) return MAXSIM_EINVAL;
  if (n == 0) return MAXSIM_EEMPTY;
  if (n > 16384 || n_docs > 0xfffffffeLL) return MAXSIM_ERANGE;
  if (tok_keep && (ids_per_token <= 0 || n % ids_per_token != 0)) return MAXSIM_EINVAL;
  if (nq == 0) return MAXSIM_OK;
  if (!emb_ids || !out_pids || !out_count || (n_docs > 0 && !tok_offsets) || ((uintptr_t)row_blocks & 7) != 0) return MAXSIM_EINVAL;
  int P = 2048;
  while (P < n) P <<= 1;
  int log_ts = 12;                                         // hash set: min(16384, 2 P) slots
  while ((1 << log_ts) < 2 * P && log_ts < 14) ++log_ts;
  const int ldsb = ((1 << log_ts) + 1024 + 4) * 4;
  int ipt_arg = tok_keep ? ids_per_token : 1;              // a power of two travels as -(log2) - 1: a shift in the kernel
  if ((ipt_arg & (ipt_arg - 1)) == 0) ipt_arg = -__builtin_ctz((unsigned)ipt_arg) - 1;
  int rc = allow_lds(k_unique_pids, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(k_unique_pids, dim3((unsigned)nq), dim3(1024), ldsb, (hipStream_t)stream, emb_ids, n, P, log_ts, id_base,
                     tok_keep, ipt_arg, tok_offsets, n_docs, n_tokens, (const uint64_t*)row_blocks, out_pids,
                     out_count);
  return check_launch();
}

int maxsim_embedding_ids_to_pids(const int64_t* emb_ids, int nq, int n, const int64_t* tok_offsets, int64_t n_docs,
                                 int64_t n_tokens, int64_t* out_pids, int32_t* out_count, void* stream) {
  return maxsim_embedding_ids_to_pids_ex(emb_ids, nq, n, 1, nullptr, 0, tok_offsets, n_docs, n_tokens, nullptr, out_pids,
                                         out_count, stream);
}

}  // extern "C"
