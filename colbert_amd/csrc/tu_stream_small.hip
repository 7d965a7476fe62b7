// tu_stream_small.hip -- the split-doc form of the h = 128 streaming kernel (maxsim_stream.h: SPLITK) for small launches:
// the reference's online call is ONE query x ~1000 candidates per rank_forward (colbert/indexing/faiss_indexers.py:234).
#include "maxsim_launch.h"
#include "maxsim_stream.h"

namespace maxsim {
namespace {

template <int DT>
int launch_split(Params& p, int dpwv, hipStream_t st) {
  constexpr int WAVES = 4;
  constexpr int NT = (StreamTraits<DT>::TILE == 16384) ? 1 : 2;
  p.dpw = dpwv * (WAVES / p.split);
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = WAVES * NT * StreamTraits<DT>::TILE + WAVES * SPLIT_MAX_DOCS * 32 * (int)sizeof(float);
  auto kern = k_maxsim_stream<MODE_RERANK, DT, WAVES, NT, 0, QT_2X16, true>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}

}  // namespace

// Returns MAXSIM_ERANGE when the launch is not one this form serves (the caller then takes the regular path): it is for
// launches so small that one wave per doc leaves most wave slots of the chip empty AND whose tiles are short enough for
// that to matter -- the 16-bit index types (8 KiB tiles): measured 31 -> 17 us at 1 query x 1000 docs x 180 tokens with
// the fp16 index, while the fp32 index (16 KiB tiles, 16 MB in flight at one wave per doc) is bandwidth-bound already
// and gets 2-25 % SLOWER when split (tools/bench_small.py).  Scores are bit-identical either way.
int launch_stream_small(Params& p, int index_dtype, hipStream_t st) {
  const int forced = MAXSIM_KNOB("MAXSIM_SPLIT", 0);  // diagnostic builds: 1 = never, 2 / 4 = always (any of the types)
  if (index_dtype != MAXSIM_F32 && index_dtype != MAXSIM_F16 && index_dtype != MAXSIM_BF16) return MAXSIM_ERANGE;
  if (index_dtype == MAXSIM_F32 && forced < 2) return MAXSIM_ERANGE;
  if (p.Lq < 1 || p.Lq > 32 || p.accum || p.q_tok0 || forced == 1) return MAXSIM_ERANGE;
  const double avg = p.n_docs > 0 ? (double)p.n_tokens / (double)p.n_docs : 1.0;
  const int64_t docs = (int64_t)p.nq * p.ncand;
  int split = forced;
  if (split != 2 && split != 4) {
    split = 1;
    if ((docs + 3) / 4 <= 320 && avg >= 128.0) split = 2;  // a workgroup on (nearly) every one of the 512 resident slots,
    if ((docs + 1) / 2 <= 320 && avg >= 256.0) split = 4;  // as long as every wave keeps >= 2 tiles of its doc
  }
  if (split == 1) return MAXSIM_ERANGE;
  p.split = split;
  int dpwv = MAXSIM_KNOB("MAXSIM_DPW", 0);
  if (dpwv <= 0 || dpwv > 64) dpwv = pick_docs_per_wave(p, 4 / split);
  if (dpwv > SPLIT_MAX_DOCS) dpwv = SPLIT_MAX_DOCS;
  int rc;
  switch (index_dtype) {
    case MAXSIM_F32: rc = launch_split<MAXSIM_F32>(p, dpwv, st); break;
    case MAXSIM_F16: rc = launch_split<MAXSIM_F16>(p, dpwv, st); break;
    default: rc = launch_split<MAXSIM_BF16>(p, dpwv, st); break;
  }
  p.split = 0;
  return rc;
}

}  // namespace maxsim
