// tu_bigh_rerank_small.hip -- the split-doc form of the LDS-query streaming kernel (maxsim_stream_bigh.h: SPLITK) for small
// launches on rows wider than 128 dims: the reference's online call is ONE query x ~1000 candidates per rank_forward
// (colbert/indexing/faiss_indexers.py:234) and its default deployment is dim 768 (proj_conf/dense.yaml:8).
#include "maxsim_launch_bigh.h"

namespace maxsim {
namespace {

template <int DT, int NPQ>
int launch_small(Params& p, int split, hipStream_t st) {
  constexpr int SUB = StreamTraits<DT>::TILE, WAVES = 8;
  const int KB = (p.h + 127) / 128;
  const int qbytes = NPQ * KB * SUB;
  if (160 * 1024 - qbytes < WAVES * SUB) return MAXSIM_ERANGE;  // eight rings of one sub-tile next to the query image
  const int teams = WAVES / split;
  int dpwv = pick_docs_per_wave(p, teams);
  if (dpwv > SPLIT_MAX_DOCS) dpwv = SPLIT_MAX_DOCS;
  p.split = split;
  p.dpw = dpwv * teams;
  p.nchunk = (p.ncand + p.dpw - 1) / p.dpw;
  const int ldsb = qbytes + WAVES * SUB;
  auto kern = k_maxsim_stream_bigh<MODE_RERANK, DT, NPQ, WAVES, 1, false, 1, false, false, true>;
  int rc = allow_lds(kern, ldsb);
  if (rc == MAXSIM_OK) {
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
    rc = check_launch();
  }
  p.split = 0;
  return rc;
}

}  // namespace

// MAXSIM_ERANGE = not a launch this form serves (the caller takes the regular path).  Served: so few (query, doc) pairs that
// one wave per doc leaves wave slots empty -- and, above all, makes the launch last as long as its LONGEST doc takes one
// wave -- on docs long enough to be cut into slices of >= 1.5 tiles.  Scores are bit-identical to the unsplit kernel's.
int launch_bigh_rerank_small(Params& p, int dt, hipStream_t st) {
  const int forced = MAXSIM_KNOB("MAXSIM_SPLIT", 0);  // diagnostic builds: 1 = never, 2 / 4 = always
  if ((p.h & 127) || p.Lq < 1 || p.Lq > 32 || p.accum || p.q_tok0 || forced == 1) return MAXSIM_ERANGE;
  const double avg = p.n_docs > 0 ? (double)p.n_tokens / (double)p.n_docs : 1.0;
  const int64_t docs = (int64_t)p.nq * p.ncand;
  int split = forced;
  if (split != 2 && split != 4) {
    // two waves per doc: dim 768 fp16, 1 query x 1000 ragged docs (avg 200, up to 384 tokens): 101 -> 80 us; four waves per
    // doc (twice the workgroups: two rounds of one workgroup per CU, each staging the 96 KiB query image): 88 us
    split = (docs * 2 <= 4096 && avg >= 96.0) ? 2 : 1;
  }
  if (split == 1) return MAXSIM_ERANGE;
  const bool same16 = dt != MAXSIM_F32 && p.q_dtype == dt;
  switch (dt) {
    case MAXSIM_F32: return launch_small<MAXSIM_F32, 1>(p, split, st);
    case MAXSIM_F16: return same16 ? launch_small<MAXSIM_F16, 1>(p, split, st) : launch_small<MAXSIM_F16, 2>(p, split, st);
    default: return same16 ? launch_small<MAXSIM_BF16, 1>(p, split, st) : launch_small<MAXSIM_BF16, 2>(p, split, st);
  }
}

}  // namespace maxsim
