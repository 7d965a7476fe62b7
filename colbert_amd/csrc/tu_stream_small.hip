// tu_stream_small.hip -- the small-launch forms of the h = 128 streaming kernel (maxsim_stream.h: SPLITK, EPI): the
// reference's online call is ONE query x ~1000 candidates per rank_forward (colbert/indexing/faiss_indexers.py:234).
#include "maxsim_launch.h"
#include "maxsim_stream.h"

namespace maxsim {
namespace {

template <int DT, bool SPLITK, bool EPI>
int launch_small_v(Params& p, int dpwv, hipStream_t st) {
  constexpr int WAVES = 4;
  constexpr int NT = (StreamTraits<DT>::TILE == 16384) ? 1 : 2;
  const int teams = SPLITK ? WAVES / p.split : WAVES;
  p.dpw = dpwv * teams;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = WAVES * NT * StreamTraits<DT>::TILE + (SPLITK ? WAVES * SPLIT_MAX_DOCS * 32 * (int)sizeof(float) : 0);
  auto kern = k_maxsim_stream<MODE_RERANK, DT, WAVES, NT, 0, QT_2X16, SPLITK, EPI>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}

template <int DT>
int launch_small_dt(Params& p, bool epi, int dpwv, hipStream_t st) {
  if (p.split > 1) return epi ? launch_small_v<DT, true, true>(p, dpwv, st) : launch_small_v<DT, true, false>(p, dpwv, st);
  return launch_small_v<DT, false, true>(p, dpwv, st);  // (unsplit without the epilogue is the regular kernel)
}

}  // namespace

// Returns MAXSIM_ERANGE when the launch is not one these forms serve (the caller then takes the regular path):
// index types other than F32 (exact) / F16 / BF16, queries longer than 32 tokens, and -- without the fused top-k -- any
// launch big enough that one wave per doc already fills the chip.
int launch_stream_small(Params& p, int index_dtype, bool epi, hipStream_t st) {
  if (index_dtype != MAXSIM_F32 && index_dtype != MAXSIM_F16 && index_dtype != MAXSIM_BF16) return MAXSIM_ERANGE;
  if (p.Lq < 1 || p.Lq > 32 || p.accum || p.q_tok0) return MAXSIM_ERANGE;
  if (epi && (p.ncand > 2048 || p.ep.k > p.ncand || p.ep.k < 1)) return MAXSIM_ERANGE;
  const double avg = p.n_docs > 0 ? (double)p.n_tokens / (double)p.n_docs : 1.0;
  // waves per doc: as many as it takes to put a workgroup on (nearly) every one of the 512 resident slots, as long as
  // every wave keeps >= 2 tiles of its doc (measured: tools/sweep_small.sh)
  const int64_t docs = (int64_t)p.nq * p.ncand;
  int split = MAXSIM_KNOB("MAXSIM_SPLIT", 0);
  if (split != 1 && split != 2 && split != 4) {
    split = 1;
    if ((docs + 3) / 4 <= 320 && avg >= 128.0) split = 2;
    if ((docs + 1) / 2 <= 320 && avg >= 256.0) split = 4;
  }
  if (split == 1 && !epi) return MAXSIM_ERANGE;
  p.split = split;
  int dpwv = MAXSIM_KNOB("MAXSIM_DPW", 0);
  if (dpwv <= 0 || dpwv > 64) dpwv = pick_docs_per_wave(p, 4 / split);
  if (split > 1 && dpwv > SPLIT_MAX_DOCS) dpwv = SPLIT_MAX_DOCS;
  int rc;
  switch (index_dtype) {
    case MAXSIM_F32: rc = launch_small_dt<MAXSIM_F32>(p, epi, dpwv, st); break;
    case MAXSIM_F16: rc = launch_small_dt<MAXSIM_F16>(p, epi, dpwv, st); break;
    default: rc = launch_small_dt<MAXSIM_BF16>(p, epi, dpwv, st); break;
  }
  p.split = 0;
  return rc;
}

}  // namespace maxsim
