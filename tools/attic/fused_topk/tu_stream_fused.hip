// tu_stream_fused.hip -- the FUSED form of the h = 128 streaming kernel (maxsim_stream.h): rerank of ONE query and the
// counting top-k in one launch, for maxsim_rank_forward (the reference's online call, faiss_indexers.py:234).
#include "maxsim_launch.h"
#include "maxsim_stream.h"

namespace maxsim {
namespace {

template <int DT, int QT, bool SPLITK>
int launch_fused_v(Params& p, hipStream_t st) {
  constexpr int WAVES = 4;
  constexpr int NT = (StreamTraits<DT>::TILE == 16384) ? 1 : 2;
  const int ldsb = WAVES * NT * StreamTraits<DT>::TILE + (SPLITK ? WAVES * SPLIT_MAX_DOCS * 32 * (int)sizeof(float) : 0);
  auto kern = k_maxsim_stream<MODE_RERANK, DT, WAVES, NT, 0, QT, SPLITK, false, true>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)p.nchunk), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}

}  // namespace

// MAXSIM_ERANGE = not a launch this form serves (the caller enqueues rerank and top-k as two launches).
// p.aux_ptr / aux0 / aux1 / worklist carry the top-k arguments (maxsim_common.h Params).
int launch_stream_fused(Params& p, int index_dtype, hipStream_t st) {
  if (p.nq != 1 || p.Lq < 1 || p.Lq > 32 || p.ncand < 1 || p.ncand > 2048 || p.accum || p.q_tok0) return MAXSIM_ERANGE;
  if (index_dtype != MAXSIM_F32 && index_dtype != MAXSIM_F16 && index_dtype != MAXSIM_BF16) return MAXSIM_ERANGE;
  if (p.n_docs > 0 && p.n_tokens <= 24 * p.n_docs) return MAXSIM_ERANGE;  // short docs: the half-tile kernel (two launches)
  const double avg = p.n_docs > 0 ? (double)p.n_tokens / (double)p.n_docs : 1.0;
  int split = 1;  // 16-bit index: docs split over 2 / 4 waves, by launch_stream_small's rule
  if (index_dtype != MAXSIM_F32) {
    if ((p.ncand + 3) / 4 <= 320 && avg >= 128.0) split = 2;
    if ((p.ncand + 1) / 2 <= 320 && avg >= 256.0) split = 4;
  }
  const int teams = 4 / split;
  int dpwv = pick_docs_per_wave(p, teams);
  if (split > 1 && dpwv > SPLIT_MAX_DOCS) dpwv = SPLIT_MAX_DOCS;
  // the first ceil(n / 16) workgroups rank: the grid must hold at least that many
  const int groups = (p.ncand + 15) / 16;
  while (dpwv > 1 && (p.ncand + dpwv * teams - 1) / (dpwv * teams) < groups) dpwv = (dpwv + 1) / 2;
  p.split = split;
  p.dpw = dpwv * teams;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  if (p.nchunk < groups || p.nchunk > FUSED_MAX_WGS) {
    p.split = 0;
    return MAXSIM_ERANGE;
  }
  int rc;
  if (split > 1) {
    rc = index_dtype == MAXSIM_F16 ? launch_fused_v<MAXSIM_F16, QT_2X16, true>(p, st) : launch_fused_v<MAXSIM_BF16, QT_2X16, true>(p, st);
  } else if (index_dtype == MAXSIM_F32) {
    rc = p.Lq <= 16 ? launch_fused_v<MAXSIM_F32, 16, false>(p, st) : launch_fused_v<MAXSIM_F32, QT_2X16, false>(p, st);
  } else {
    rc = index_dtype == MAXSIM_F16 ? launch_fused_v<MAXSIM_F16, QT_2X16, false>(p, st) : launch_fused_v<MAXSIM_BF16, QT_2X16, false>(p, st);
  }
  p.split = 0;
  return rc;
}

}  // namespace maxsim
