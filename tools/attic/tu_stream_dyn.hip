// tu_stream_dyn.hip -- launch of the persistent, work-queue form of the short-doc rerank kernel (maxsim_stream_dyn.h).
#include <atomic>

#include "maxsim_launch.h"
#include "maxsim_stream_dyn.h"

namespace maxsim {
namespace {

// The work-queue counters (eight "taken" + one "finished", padded to 16 ints): one set per (device, stream), since
// launches on ONE stream run one after the other (the kernel leaves its set zeroed) while launches on different streams
// may overlap.  64 sets per device in one 4 KB allocation made at the first use; a stream that finds no free slot takes
// the static kernel instead.
constexpr int SLOTS = 64, SLOT_INTS = 128;  // (a slot: up to 127 queue counters + the finished counter)
struct DynSlots {
  std::atomic<int> state{0};  // 0 = untouched, 1 = being set up, 2 = ready, 3 = unavailable
  int* base = nullptr;
  std::atomic<uintptr_t> owner[SLOTS];
};
DynSlots g_dyn[8];

int* dyn_counters_for(hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 8) return nullptr;
  DynSlots& d = g_dyn[dev];
  int s = d.state.load(std::memory_order_acquire);
  if (s == 0) {
    int expect = 0;
    if (d.state.compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) {
      int* b = nullptr;
      const size_t bytes = (size_t)SLOTS * SLOT_INTS * sizeof(int);
      const bool ok = hipMalloc((void**)&b, bytes) == hipSuccess && hipMemset(b, 0, bytes) == hipSuccess &&
                      hipDeviceSynchronize() == hipSuccess;
      for (auto& o : d.owner) o.store(0, std::memory_order_relaxed);
      d.base = ok ? b : nullptr;
      d.state.store(ok ? 2 : 3, std::memory_order_release);
    }
    while ((s = d.state.load(std::memory_order_acquire)) == 1) {
    }
  }
  if (s != 2) return nullptr;
  const uintptr_t key = (uintptr_t)st + 1;  // (the null stream is a stream too)
  const unsigned h0 = (unsigned)((key >> 4) * 2654435761u) % SLOTS;
  for (unsigned i = 0; i < SLOTS; ++i) {
    const unsigned k = (h0 + i) % SLOTS;
    uintptr_t cur = d.owner[k].load(std::memory_order_acquire);
    if (cur == key) return d.base + SLOT_INTS * k;
    if (cur == 0) {
      if (d.owner[k].compare_exchange_strong(cur, key, std::memory_order_acq_rel)) return d.base + SLOT_INTS * k;
      if (cur == key) return d.base + SLOT_INTS * k;
    }
  }
  return nullptr;
}

template <int NCB>
int launch_dyn(Params& p, int* counters, int dpi, hipStream_t st) {
  constexpr int WAVES = 4, NT = 2;
  p.dpw = dpi;
  p.nchunk = (p.ncand + dpi - 1) / dpi;
  p.argmax = counters;
  const int ldsb = WAVES * NT * 8192;
  auto kern = k_maxsim_stream_f32h_dyn<WAVES, NCB, NT>;
  int rc = allow_lds(kern, ldsb);
  if (rc == 0) {
    hipLaunchKernelGGL(kern, dim3(512u), dim3(WAVES * 64), ldsb, st, KARGS_PASS(p));
    rc = check_launch();
  }
  p.argmax = nullptr;
  p.split = 1;
  return rc;
}

}  // namespace

// MAXSIM_ERANGE: not a launch this form serves -- the caller takes the static kernel.  Serves: fp32 index of short docs
// (the static short-doc kernel's domain) with the packed doc table, enough docs in the launch that the tail of a static
// launch matters and every wave of the persistent one gets several items.
int launch_stream_dyn(Params& p, hipStream_t st) {
  const int dpi = MAXSIM_KNOB("MAXSIM_DYN_DPI", 16);
  const int64_t min_docs = MAXSIM_KNOB("MAXSIM_DYN_MIN_DOCS", 100000);
  if (dpi < 1 || dpi > 64 || (int64_t)p.nq * p.ncand < min_docs || p.Lq > 32 || p.ncand < 1) return MAXSIM_ERANGE;
  if (!p.doc_table) return MAXSIM_ERANGE;  // (the kernel fetches descriptors as packed rows, an item ahead)
  if ((int64_t)p.nq * ((p.ncand + dpi - 1) / dpi) > 0x0fffffffLL) return MAXSIM_ERANGE;
  int* const counters = dyn_counters_for(st);
  if (!counters) return MAXSIM_ERANGE;
  return p.Lq <= 16 ? launch_dyn<1>(p, counters, dpi, st) : launch_dyn<2>(p, counters, dpi, st);
}

}  // namespace maxsim
