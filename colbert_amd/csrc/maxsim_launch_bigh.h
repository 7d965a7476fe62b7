// maxsim_launch_bigh.h -- launch heuristics of the LDS-query streaming kernel (included by the two tu_bigh_*.hip units).
#pragma once
#include <type_traits>

#include "maxsim_launch.h"
#include "maxsim_stream_bigh.h"

namespace maxsim {
namespace {

// Query-in-LDS streaming kernel (h = 128 * KB): QB query images of NPQ x KB x sub-tile bytes each, the rest of the
// 160 KiB goes to the waves' rings: as many waves (<= 8) as fit with NT sub-tiles each.
template <int MODE, int DT, int NPQ, bool AM, int QB, bool PART = false>
int launch_stream_bigh_q(Params& p, hipStream_t st) {
  constexpr int SUB = StreamTraits<DT>::TILE;
  const int KB = (p.h + 127) / 128;
  const int qbytes = QB * NPQ * KB * SUB;
  const int avail = 160 * 1024 - qbytes;
  int dpwv = MAXSIM_KNOB("MAXSIM_DPW", 0);
  if (dpwv <= 0 || dpwv > 64) dpwv = pick_docs_per_wave(p, 4);
  const bool knob_dpw = MAXSIM_KNOB("MAXSIM_DPW", 0) > 0;
  auto launch = [&](auto kern, int waves, int ldsb) {
    int rc = allow_lds(kern, ldsb);
    if (rc) return rc;
    const int nqblk = (p.nq + QB - 1) / QB;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nqblk * p.nchunk)), dim3(waves * 64), ldsb, st, KARGS_PASS(p));
    return check_launch();
  };
  // kern_bal: the same kernel with the token-balanced cut (ragged 16-bit index, a workgroup's docs one per lane), or nullptr
  auto go3 = [&](auto kern, auto kern_bal, int waves, int nt, int qb) {   // qb: bytes of the staged query image(s)
    // (workgroups resident at once: one per CU above 80 KiB of LDS, else two)
    const int dw = (MODE == MODE_RERANK && !knob_dpw) ? refine_docs_per_wave(p, dpwv, waves, qb + waves * nt * SUB > 80 * 1024 ? 256 : 512) : dpwv;
    p.dpw = dw * waves;
    p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
    const int ldsb = qb + waves * nt * SUB;
    if constexpr (!std::is_same<decltype(kern_bal), std::nullptr_t>::value) {
      if (p.uniform_len == 0 && p.dpw <= 64 && dw >= 2 && MAXSIM_KNOB("MAXSIM_BAL", 1) != 0) return launch(kern_bal, waves, ldsb);
    }
    return launch(kern, waves, ldsb);
  };
  auto go2 = [&](auto kern, auto kern_bal, int waves, int nt) { return go3(kern, kern_bal, waves, nt, qbytes); };
  auto go = [&](auto kern, int waves, int nt) { return go3(kern, nullptr, waves, nt, qbytes); };
  if constexpr (PART) {  // odd widths: one configuration (keeps the number of instantiations down)
    if (avail >= 4 * 1 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 4, 1, AM, QB, true>, 4, 1);
    return MAXSIM_ERANGE;
  } else {
    if constexpr (QB == 1) {
#ifdef MAXSIM_DIAG
      if constexpr (MODE == MODE_RERANK) {  // diagnostic: MAXSIM_BIGH_SHAPE = waves * 10 + sub-tiles per wave
        const int shape = MAXSIM_KNOB("MAXSIM_BIGH_SHAPE", 0);
        if (shape == 81 && avail >= 8 * 1 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 1, AM, QB>, 8, 1);
        if (shape == 42 && avail >= 4 * 2 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 4, 2, AM, QB>, 4, 2);
        if (shape == 41 && avail >= 4 * 1 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 4, 1, AM, QB>, 4, 1);
        if (shape == 121 && avail >= 12 * 1 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 12, 1, AM, QB>, 12, 1);
        if (shape == 141 && avail >= 14 * 1 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 14, 1, AM, QB>, 14, 1);
        if (shape == 161 && avail >= 16 * 1 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 16, 1, AM, QB>, 16, 1);
        if (shape == 62 && avail >= 6 * 2 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 6, 2, AM, QB>, 6, 2);
      }
#endif
      constexpr bool BALOK = MODE == MODE_RERANK && DT != MAXSIM_F32 && !AM;   // (ragged 16-bit indexes: the reference's deployment)
      // docs of a few tokens (the multi-view configuration: 16 per doc): the launch is bound by how many waves stream, not by
      // bytes in flight per wave -- mv768 with a one-piece image: 4 waves x 2 sub-tiles 0.684 of the HBM peak, 8 x 1 0.788,
      // 12 x 1 0.796 (long docs: all shapes within 1 %)
      const bool short_docs = p.n_docs > 0 && p.n_tokens <= 64 * p.n_docs;
      if constexpr (BALOK) {
        // at most 16 query tokens (q_view = 16, dense.yaml:31): the 16-row query image (HALFQ) -- half the LDS for the image,
        // the rest goes to the rings; bit-identical scores (diagnostic: MAXSIM_HALFQ=0 keeps the 32-row image)
        if (p.Lq <= 16 && MAXSIM_KNOB("MAXSIM_HALFQ", 1) != 0) {
          const int qh = qbytes / 2, availh = 160 * 1024 - qh;
          if (short_docs && availh >= 12 * 1 * SUB)
            return go3(k_maxsim_stream_bigh<MODE, DT, NPQ, 12, 1, AM, QB, false, false, false, false, true>, nullptr, 12, 1, qh);
          if (availh >= 8 * 2 * SUB)
            return go3(k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 2, AM, QB, false, false, false, false, true>, nullptr, 8, 2, qh);
          if (availh >= 8 * 1 * SUB)
            return go3(k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 1, AM, QB, false, false, false, false, true>, nullptr, 8, 1, qh);
        }
      }
      if (avail >= 8 * 2 * SUB) {
        if constexpr (BALOK) return go2(k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 2, AM, QB>, k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 2, AM, QB, false, false, false, true>, 8, 2);
        else return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 2, AM, QB>, 8, 2);
      }
      // A two-piece query image (an fp32 query on a 16-bit index: the reference's deployment, dim 768 fp16 -> 96 KiB)
      // leaves 64 KiB for the rings: eight waves with one sub-tile each beat four waves with two -- the same bytes in
      // flight, but twice the matrix work per byte has twice the waves to hide behind (ragged dim-768 fp16 docs:
      // 13.45 -> 11.81 ms, 0.73 -> 0.83 of peak; with a one-piece image, e.g. C5, the two shapes measure the same)
      if constexpr (MODE == MODE_RERANK)
        if ((NPQ == 2 || short_docs) && avail >= 8 * 1 * SUB) {
          if constexpr (BALOK) return go2(k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 1, AM, QB>, k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 1, AM, QB, false, false, false, true>, 8, 1);
          else return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 1, AM, QB>, 8, 1);
        }
      if (avail >= 4 * 2 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 4, 2, AM, QB>, 4, 2);
    } else {  // several queries per workgroup: the matrix work per sub-tile is QB x longer, one sub-tile ahead suffices
      if (avail >= 8 * 1 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 8, 1, AM, QB>, 8, 1);
    }
    if (avail >= 4 * 1 * SUB) return go(k_maxsim_stream_bigh<MODE, DT, NPQ, 4, 1, AM, QB>, 4, 1);
    return MAXSIM_ERANGE;
  }
}

// All-pairs (dense) launches share each doc sub-tile between QB queries of a workgroup: the largest QB whose query
// images leave room for >= 4 waves x 1 sub-tile and whose accumulators (16 VGPRs per query and piece) stay <= 64
// (QB = 8 spills).
template <int MODE, int DT, int NPQ, bool AM>
int launch_stream_bigh(Params& p, hipStream_t st) {
  if (p.h & 127) return launch_stream_bigh_q<MODE, DT, NPQ, AM, 1, true>(p, st);  // partial last block
  if constexpr (MODE == MODE_DENSE) {
    constexpr int SUB = StreamTraits<DT>::TILE;
    const int qimg = NPQ * (p.h / 128) * SUB;
    const int qb_env = MAXSIM_KNOB("MAXSIM_QB", 0);  // tuning knob (diagnostic builds)
    auto fits = [&](int qb) { return qb * qimg + 4 * SUB <= 160 * 1024 && qb * NPQ <= 4 && (qb_env == 0 || qb <= qb_env); };
    if (p.nq >= 4 && fits(4)) return launch_stream_bigh_q<MODE, DT, NPQ, AM, 4>(p, st);
    if (p.nq >= 2 && fits(2)) return launch_stream_bigh_q<MODE, DT, NPQ, AM, 2>(p, st);
  }
  return launch_stream_bigh_q<MODE, DT, NPQ, AM, 1>(p, st);
}

template <int MODE, bool AM>
int launch_bigh(Params& p, int dt, hipStream_t st) {
  const bool same16 = dt != MAXSIM_F32 && p.q_dtype == dt;  // query already in the index's 16-bit type: one piece
  switch (dt) {
    case MAXSIM_F32: return launch_stream_bigh<MODE, MAXSIM_F32, 1, AM>(p, st);
    case MAXSIM_F16:
      return same16 ? launch_stream_bigh<MODE, MAXSIM_F16, 1, AM>(p, st) : launch_stream_bigh<MODE, MAXSIM_F16, 2, AM>(p, st);
    default:
      return same16 ? launch_stream_bigh<MODE, MAXSIM_BF16, 1, AM>(p, st) : launch_stream_bigh<MODE, MAXSIM_BF16, 2, AM>(p, st);
  }
}

}  // namespace
}  // namespace maxsim
