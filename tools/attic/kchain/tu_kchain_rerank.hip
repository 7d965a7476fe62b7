// tu_kchain_rerank.hip -- the register-query chain kernel (maxsim_stream_kchain.h) for rerank on wide 16-bit rows.
#include "maxsim_launch.h"
#include "maxsim_stream_kchain.h"

namespace maxsim {
namespace {

template <int DT, int NPQ, int KB>
int launch_kchain(Params& p, hipStream_t st) {
  constexpr int SUB = StreamTraits<DT>::TILE, HOP = NPQ * 4096;
  constexpr int NT = (KB * 2 * SUB + (KB - 1) * HOP + KCHAIN_MAPD * KCHAIN_MAPB <= 160 * 1024) ? 2 : 1;
  constexpr int MAPS = KCHAIN_MAPD * KCHAIN_MAPB;
  constexpr int NT_FITS = KB * 2 * SUB + (KB - 1) * HOP + MAPS <= 160 * 1024;
  static_assert(NT_FITS || NT == 1, "ring depth");
  constexpr int ldsb = KB * NT * SUB + (KB - 1) * HOP + MAPS;
  // docs per workgroup: one token stream of ~6.4 k tokens (200 tiles against KB - 1 steps of pipeline fill), shortened
  // while the launch would not give every CU a workgroup; at most 64 (scores are parked one per lane)
  double avg = p.n_docs > 0 ? (double)p.n_tokens / (double)p.n_docs : 1.0;
  if (avg < 1.0) avg = 1.0;
  int dpw = MAXSIM_KNOB("MAXSIM_DPW", 0);
  if (dpw <= 0 || dpw > 64) {
    dpw = (int)(6400.0 / avg + 0.5);
    dpw = dpw < 1 ? 1 : dpw > 64 ? 64 : dpw;
    // (a launch of 200-256 workgroups is one round on the 256 CUs; halving its docs per workgroup would make it two)
    while (dpw > 1 && (int64_t)p.nq * ((p.ncand + dpw - 1) / dpw) < 200) dpw = (dpw + 1) / 2;
  }
  p.dpw = dpw;
  p.nchunk = (p.ncand + dpw - 1) / dpw;
  auto kern = k_maxsim_stream_kchain<DT, NPQ, KB, NT>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.nq * p.nchunk)), dim3((KB + 1) * 64), ldsb, st, KARGS_PASS(p));
  return check_launch();
}

}  // namespace

// MAXSIM_ERANGE = not a launch this kernel serves (the caller takes the LDS-query kernel).
int launch_kchain_rerank(Params& p, int dt, hipStream_t st) {
  if (p.h != 768 || p.Lq < 1 || p.Lq > 32 || p.accum || p.q_tok0 || p.worklist) return MAXSIM_ERANGE;
  if (dt != MAXSIM_F16 && dt != MAXSIM_BF16) return MAXSIM_ERANGE;
  const bool same16 = p.q_dtype == dt;
  if (dt == MAXSIM_F16) return same16 ? launch_kchain<MAXSIM_F16, 1, 6>(p, st) : launch_kchain<MAXSIM_F16, 2, 6>(p, st);
  return same16 ? launch_kchain<MAXSIM_BF16, 1, 6>(p, st) : launch_kchain<MAXSIM_BF16, 2, 6>(p, st);
}

}  // namespace maxsim
