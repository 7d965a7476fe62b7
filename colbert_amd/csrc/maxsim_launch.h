// maxsim_launch.h -- host-side launch helpers shared by the translation units of libmaxsim, and the entry points each
// unit exports to the C-ABI file (the kernels are split over several .hip files only to compile them in parallel).
#pragma once
#include <atomic>

#include "maxsim_common.h"

namespace maxsim {

// Kernels that need more than the default 64 KiB of dynamic LDS must be granted it once per (device, kernel):
// hipFuncSetAttribute costs a few microseconds, which is a fifth of a single-query rerank call, so the grant is
// remembered in a small lock-free table (a lost race only repeats the idempotent call).
struct LdsGrant {
  std::atomic<const void*> fn{nullptr};
  std::atomic<int> bytes{0};
};
inline int allow_lds_slow(const void* fn, int bytes) {
  static LdsGrant table[8][128];  // [device][open addressing over the kernels of this library]
  auto grant = [&] {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? MAXSIM_OK : MAXSIM_ELAUNCH;
  };
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MAXSIM_ELAUNCH;
  if (dev < 0 || dev >= 8) return grant();
  const uintptr_t key = (uintptr_t)fn;
  for (unsigned probe = 0; probe < 128; ++probe) {
    LdsGrant& g = table[dev][(unsigned)((key >> 4) + probe) & 127];
    const void* cur = g.fn.load(std::memory_order_acquire);
    if (cur == fn) {
      if (g.bytes.load(std::memory_order_acquire) >= bytes) return MAXSIM_OK;
      const int rc = grant();
      if (rc == MAXSIM_OK) g.bytes.store(bytes, std::memory_order_release);
      return rc;
    }
    if (cur == nullptr) {
      const int rc = grant();
      const void* expect = nullptr;
      if (rc == MAXSIM_OK && g.fn.compare_exchange_strong(expect, fn, std::memory_order_acq_rel))
        g.bytes.store(bytes, std::memory_order_release);
      return rc;
    }
  }
  return grant();  // table full (cannot happen with this library's kernel count)
}
template <typename K>
inline int allow_lds(K kernel, int bytes) {
  if (bytes <= 64 * 1024) return MAXSIM_OK;
  return allow_lds_slow((const void*)kernel, bytes);
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? MAXSIM_OK : MAXSIM_ELAUNCH; }

// Tuning / ablation knobs exist only in diagnostic builds (build.sh -DMAXSIM_DIAG, loaded through MAXSIM_LIB by the
// tools/ scripts).  The shipped library never consults the environment: a stray variable cannot select an ablation
// kernel (those return wrong scores by design) and no launch pays for a getenv.
#ifdef MAXSIM_DIAG
inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#define MAXSIM_KNOB(name, dflt) ([] { static const int v = ::maxsim::env_int(name, dflt); return v; }())
#else
#define MAXSIM_KNOB(name, dflt) (dflt)
#endif

// Docs per wave for the streaming kernels: a wave's token stream should be long (~1.5k tokens) so that the one partly
// filled last tile, the 16 KiB query-tile load and the dependent descriptor loads of its start-up are noise;
// small launches shorten it only as far as it takes to fill (nearly) one round of the 512 resident workgroup slots --
// measured (tools/sweep_minwgs.sh, 2..32 queries x 1000 docs): against the earlier "at least 2048 workgroups" rule
// this is 3-13 % faster on 32x180 docs, 7-17 % with the fp16 index, up to 6x on 8-token docs, within +-9 % on ragged
// docs.  At most 64 docs per wave (scores are parked one per lane).
inline int pick_docs_per_wave(const Params& p, int waves) {
  double avg = p.n_docs > 0 ? (double)p.n_tokens / (double)p.n_docs : 1.0;
  if (avg < 1.0) avg = 1.0;
  int dpwv = (int)(1440.0 / avg + 0.5);
  if (dpwv < 1) dpwv = 1;
  if (dpwv > 64) dpwv = 64;
  const int min_wgs = MAXSIM_KNOB("MAXSIM_MIN_WGS", 448);
  while (dpwv > 1 && (int64_t)p.nq * ((p.ncand + dpwv * waves - 1) / (dpwv * waves)) < min_wgs) dpwv = (dpwv + 1) / 2;
  return dpwv;
}

// Mid-size launches: a static grid a little larger than a whole number of rounds of the kernel's resident workgroup
// slots (`slots` = workgroups the chip holds at once: 256 CUs x workgroups per CU) ends with a nearly empty round --
// 16 queries x 1000 docs at dim 768: 288 workgroups on 256 slots, the last 32 alone for as long as the first 256 took.
// A workgroup's duration is its own serial stream, ~ proportional to its docs: among the docs-per-wave values between half
// and twice the stream-length rule's choice, take the one with the smallest rounds x docs when that saves >= 10 %.
// Scores do not depend on the cut (bit-identical); large launches (>= 4 rounds) are left alone.
inline int refine_docs_per_wave(const Params& p, int dpwv, int waves, int slots) {
  auto rounds = [&](int d) {
    const int64_t wgs = (int64_t)p.nq * ((p.ncand + (int64_t)d * waves - 1) / ((int64_t)d * waves));
    return (wgs + slots - 1) / slots;
  };
  const int64_t r0 = rounds(dpwv);
  if (r0 < 2 || r0 > 4 || MAXSIM_KNOB("MAXSIM_NO_REFINE", 0)) return dpwv;
  int best = dpwv;
  int64_t best_cost = r0 * dpwv;
  const int hi = dpwv * 2 < 64 ? dpwv * 2 : 64, lo = dpwv / 2 > 1 ? dpwv / 2 : 1;
  for (int d = hi; d >= lo; --d) {
    const int64_t c = rounds(d) * d;
    if (c < best_cost) { best_cost = c; best = d; }
  }
  return best_cost * 10 <= r0 * dpwv * 9 ? best : dpwv;
}

// tu_stream.hip: the h = 128 register-query kernel.  index_dtype: MAXSIM_F32 / F16 / BF16 / F32_FAST / F32_BF16X3.
int launch_stream_rerank(Params& p, int index_dtype, hipStream_t st);
int launch_stream_dense_f32(Params& p, hipStream_t st);
// tu_stream.hip, counted candidate rows: the same kernel walking a device-built work list (maxsim_worklist.h) on a fixed
// grid.  max_items = an upper bound of the list length known on the host (sizes the grid below its cap).
int launch_stream_list(Params& p, int index_dtype, int64_t max_items, hipStream_t st);
int stream_list_docs_per_item(const Params& p);
bool stream_list_serves(const Params& p, int index_dtype);
// tu_stream_small.hip: small launches of the same kernel with each doc split over several waves (bit-identical scores).
// MAXSIM_ERANGE = not a launch this form serves (take the regular path).
int launch_stream_small(Params& p, int index_dtype, hipStream_t st);
// tu_bigh_rerank.hip / tu_bigh_dense.hip: the LDS-query kernel (any 16 <= h <= 1024); dt: MAXSIM_F32 / F16 / BF16.
// Return MAXSIM_ERANGE when the query image does not fit in LDS.
int launch_bigh_rerank(Params& p, int dt, hipStream_t st);
int launch_bigh_dense(Params& p, int dt, bool argmax, hipStream_t st);
// tu_bigh_rerank_small.hip: small launches of the same kernel with each doc split over 2 / 4 waves (bit-identical scores).
// MAXSIM_ERANGE = not a launch this form serves (take the regular path).
int launch_bigh_rerank_small(Params& p, int dt, hipStream_t st);
// tu_bigh_rerank_list.hip: the same kernel walking a work list of WORKGROUP items (counted rows; maxsim_worklist.h).
// bigh_list_waves: waves per workgroup of that form for this launch, 0 = not served.
int bigh_list_waves(const Params& p, int dt);
int launch_bigh_rerank_list(Params& p, int dt, int64_t max_items, hipStream_t st);
// tu_allpairs.hip: the GEMM-blocked all-pairs kernel (16-bit operands, Lq <= 32, Ld <= 384); MAXSIM_ERANGE = not its shape.
int launch_allpairs(const Params& p, int dt, bool argmax, hipStream_t st);
bool allpairs_serves(int dt, int q_dtype, int mask_dtype, int nq, int nd, int Lq, int Ld, int h);

}  // namespace maxsim
