// tu_allpairs.hip -- launch of the GEMM-blocked all-pairs (training-form) kernel, maxsim_allpairs.h.
#include "maxsim_allpairs.h"
#include "maxsim_launch.h"

namespace maxsim {
namespace {

int cu_count() {  // workgroups are persistent, one per CU; the XCD-aware tile walk wants a multiple of 8
  static int cus[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 8) return 256;
  if (cus[dev] == 0) {
    hipDeviceProp_t prop;
    cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    cus[dev] = (cus[dev] / 8) * 8;
    if (cus[dev] < 8) cus[dev] = 8;
  }
  return cus[dev];
}

template <int DT, int R, int QB, bool AM>
int launch_r(const AllPairsArgs& a, hipStream_t st) {
  // 2 ring stages of 64 dims + the row-waves' (max, index) exchange + the tile's mask rows
  constexpr int ldsb = 2 * (128 * R + 64 * QB) * 128 + 2 * (2 * QB * 4 * 32) * 4 + (128 * R + 64 * QB) * 4;
  auto kern = k_maxsim_allpairs<DT, R, QB, AM>;
  int rc = allow_lds(kern, ldsb);
  if (rc) return rc;
  hipLaunchKernelGGL(kern, dim3((unsigned)cu_count()), dim3(512), ldsb, st, a);
  return check_launch();
}

template <int DT, bool AM>
int launch_dt(const AllPairsArgs& a, hipStream_t st) {
  if (a.Ld <= 128) return launch_r<DT, 1, 4, AM>(a, st);
  if (a.Ld <= 256) return launch_r<DT, 2, 4, AM>(a, st);
  return launch_r<DT, 3, 3, AM>(a, st);
}

}  // namespace

// MAXSIM_ERANGE: not a shape this kernel serves (16-bit operands, h % 64 == 0, h >= 128, Lq <= 32, Ld <= 384, enough work to fill
// the chip with (doc, 8-query) tiles) -- the caller takes the streaming kernel.
int launch_allpairs(const Params& p, int dt, bool argmax, hipStream_t st) {
  if (dt != MAXSIM_F16 && dt != MAXSIM_BF16) return MAXSIM_ERANGE;
  if (p.q_dtype != dt) return MAXSIM_ERANGE;
  if (p.h < 128 || (p.h & 63) || p.Lq < 1 || p.Lq > 32 || p.Ld < 1 || p.Ld > 384) return MAXSIM_ERANGE;
  if ((((uintptr_t)p.Q | (uintptr_t)p.index) & 15) != 0) return MAXSIM_ERANGE;
  // the kernel addresses Q and D with 32-bit offsets from the tensor base (buffer descriptors: reads past the end give 0)
  const uint64_t lim = 0xF0000000ull, rowb = (uint64_t)p.h * 2;
  if ((uint64_t)p.ncand * p.Ld * rowb >= lim || (uint64_t)p.nq * p.Lq * rowb >= lim) return MAXSIM_ERANGE;
  // masks travel by LDS-DMA as float words (see the kernel): float32 masks or none; colbert_amd.score converts
  if (p.mask_dtype != MAXSIM_MASK_NONE && p.mask_dtype != MAXSIM_MASK_F32) return MAXSIM_ERANGE;
  const int64_t tiles = (int64_t)p.ncand * ((p.nq + 7) / 8);  // (doc, query block) tiles, roughly
  const int min_tiles = MAXSIM_KNOB("MAXSIM_ALLPAIRS_MIN_TILES", 128);  // diagnostic builds: 0 = always, huge = never
  if (tiles < min_tiles) return MAXSIM_ERANGE;
  AllPairsArgs a{};
  a.Q = p.Q; a.D = p.index; a.q_mask = p.q_mask; a.d_mask = p.d_mask;
  a.scores = p.scores; a.argmax = p.argmax;
  a.mask_dtype = p.mask_dtype; a.nq = p.nq; a.nd = p.ncand; a.Lq = p.Lq; a.Ld = p.Ld; a.h = p.h;
  if (dt == MAXSIM_F16) return argmax ? launch_dt<MAXSIM_F16, true>(a, st) : launch_dt<MAXSIM_F16, false>(a, st);
  return argmax ? launch_dt<MAXSIM_BF16, true>(a, st) : launch_dt<MAXSIM_BF16, false>(a, st);
}

}  // namespace maxsim
