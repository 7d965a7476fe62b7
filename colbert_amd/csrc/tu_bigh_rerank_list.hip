// tu_bigh_rerank_list.hip -- the LDS-query streaming kernel (maxsim_stream_bigh.h) walking a device-built work list of
// WORKGROUP items: counted candidate rows (a doc shard's share of every list, ANN pid lists) for rows wider than 128
// dims -- the reference's default deployment is dim 768 (proj_conf/dense.yaml:8).
#include "maxsim_launch_bigh.h"

namespace maxsim {
namespace {

// The ring shape launch_stream_bigh_q would pick for a rerank launch of this width / type (waves, sub-tiles per wave).
// halfq: at most 16 query tokens on a 16-bit index of short docs (the multi-view configuration): the 16-row query image and 12
// waves x 1 sub-tile, as the static grid (maxsim_launch_bigh.h; bit-identical scores)
template <int DT, int NPQ>
bool list_shape(const Params& p, int& waves, int& nt, bool& halfq) {
  constexpr int SUB = StreamTraits<DT>::TILE;
  const int qfull = NPQ * ((p.h + 127) / 128) * SUB;
  halfq = false;
  if (DT != MAXSIM_F32 && p.Lq <= 16 && (p.h & 127) == 0 && p.n_docs > 0 && p.n_tokens <= 64 * p.n_docs &&
      160 * 1024 - qfull / 2 >= 12 * SUB && MAXSIM_KNOB("MAXSIM_HALFQ", 1) != 0) {
    halfq = true; waves = 12; nt = 1;
    return true;
  }
  const int avail = 160 * 1024 - qfull;
  if (avail >= 8 * 2 * SUB) { waves = 8; nt = 2; return true; }
  if (NPQ == 2 && avail >= 8 * 1 * SUB) { waves = 8; nt = 1; return true; }
  if (avail >= 4 * 2 * SUB) { waves = 4; nt = 2; return true; }
  if (avail >= 4 * 1 * SUB) { waves = 4; nt = 1; return true; }
  return false;
}

template <int DT, int NPQ>
int launch_list(Params& p, int64_t max_items, hipStream_t st) {
  constexpr int SUB = StreamTraits<DT>::TILE;
  int waves = 0, nt = 0;
  bool halfq = false;
  if (!list_shape<DT, NPQ>(p, waves, nt, halfq)) return MAXSIM_ERANGE;
  const bool part = (p.h & 127) != 0;  // last block partial: the one ring shape the static grid uses for such widths
  if (part) { waves = 4; nt = 1; }
  const int KB = (p.h + 127) / 128;
  const int ldsb = NPQ * KB * SUB / (halfq ? 2 : 1) + waves * nt * SUB;
  // one workgroup per CU is resident (the query image + rings take most of the LDS): a few rounds of them
  int64_t wgs = max_items < 1 ? 1 : max_items;
  const int cap = MAXSIM_KNOB("MAXSIM_LIST_WGS", 1024);
  if (wgs > cap) wgs = cap;
  wgs = (wgs + 7) & ~(int64_t)7;
  auto go = [&](auto kern) {
    int rc = allow_lds(kern, ldsb);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(waves * 64), ldsb, st, KARGS_PASS(p));
    return check_launch();
  };
  if constexpr (DT != MAXSIM_F32) {
    if (halfq) return go(k_maxsim_stream_bigh<MODE_RERANK, DT, NPQ, 12, 1, false, 1, false, true, false, false, true>);
  }
  if (part) return go(k_maxsim_stream_bigh<MODE_RERANK, DT, NPQ, 4, 1, false, 1, true, true>);
  if (waves == 8 && nt == 2) return go(k_maxsim_stream_bigh<MODE_RERANK, DT, NPQ, 8, 2, false, 1, false, true>);
  if (waves == 8) return go(k_maxsim_stream_bigh<MODE_RERANK, DT, NPQ, 8, 1, false, 1, false, true>);
  if (nt == 2) return go(k_maxsim_stream_bigh<MODE_RERANK, DT, NPQ, 4, 2, false, 1, false, true>);
  return go(k_maxsim_stream_bigh<MODE_RERANK, DT, NPQ, 4, 1, false, 1, false, true>);
}

}  // namespace

// Waves per workgroup of the list form for this launch (0: not served -- a query image that does not fit), so that the
// caller can size the workgroup items (docs per item = waves x docs per wave).  Widths that are not a multiple of 128
// (partial last block) run the 4-wave shape the static grid uses for them.
int bigh_list_waves(const Params& p, int dt) {
  const bool same16 = dt != MAXSIM_F32 && p.q_dtype == dt;
  int waves = 0, nt = 0;
  bool ok, halfq = false;
  switch (dt) {
    case MAXSIM_F32: ok = list_shape<MAXSIM_F32, 1>(p, waves, nt, halfq); break;
    case MAXSIM_F16: ok = same16 ? list_shape<MAXSIM_F16, 1>(p, waves, nt, halfq) : list_shape<MAXSIM_F16, 2>(p, waves, nt, halfq); break;
    default: ok = same16 ? list_shape<MAXSIM_BF16, 1>(p, waves, nt, halfq) : list_shape<MAXSIM_BF16, 2>(p, waves, nt, halfq); break;
  }
  return !ok ? 0 : (p.h & 127) ? 4 : waves;
}

int launch_bigh_rerank_list(Params& p, int dt, int64_t max_items, hipStream_t st) {
  const bool same16 = dt != MAXSIM_F32 && p.q_dtype == dt;
  switch (dt) {
    case MAXSIM_F32: return launch_list<MAXSIM_F32, 1>(p, max_items, st);
    case MAXSIM_F16: return same16 ? launch_list<MAXSIM_F16, 1>(p, max_items, st) : launch_list<MAXSIM_F16, 2>(p, max_items, st);
    default: return same16 ? launch_list<MAXSIM_BF16, 1>(p, max_items, st) : launch_list<MAXSIM_BF16, 2>(p, max_items, st);
  }
}

}  // namespace maxsim
