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
  // (the same tile on v_mfma_f32_16x16x32 -- a higher held clock in the timing experiment -DAP_MFMA16=1, 2.48 -> 2.31 ms -- was
  //  built, bit-identical, and no faster: tools/attic/allpairs16/README.md)
  return launch_r<DT, 3, 3, AM>(a, st);
}

}  // namespace

// Does the GEMM-blocked kernel serve this shape (16-bit operands, h % 64 == 0, h >= 128, Lq <= 32, Ld <= 384, float32
// masks or none, tensors below 3.75 GB, enough work to fill the chip with (doc, 8-query) tiles)?  Otherwise the caller
// takes the streaming kernel.
bool allpairs_serves(int dt, int q_dtype, int mask_dtype, int nq, int nd, int Lq, int Ld, int h) {
  if (dt != MAXSIM_F16 && dt != MAXSIM_BF16) return false;
  if (q_dtype != dt) return false;
  if (h < 128 || (h & 63) || Lq < 1 || Lq > 32 || Ld < 1 || Ld > 384) return false;
  // masks travel by LDS-DMA as float words (see the kernel): float32 masks or none; colbert_amd.score converts
  if (mask_dtype != MAXSIM_MASK_NONE && mask_dtype != MAXSIM_MASK_F32) return false;
  // the kernel addresses Q and D with 32-bit offsets from the tensor base (buffer descriptors: reads past the end give 0)
  const uint64_t lim = 0xF0000000ull, rowb = (uint64_t)h * 2;
  if ((uint64_t)nd * Ld * rowb >= lim || (uint64_t)nq * Lq * rowb >= lim) return false;
  const int64_t tiles = (int64_t)nd * ((nq + 7) / 8);  // (doc, query block) tiles, roughly
  const int min_tiles = MAXSIM_KNOB("MAXSIM_ALLPAIRS_MIN_TILES", 128);  // diagnostic builds: 0 = always, huge = never
  return tiles >= min_tiles;
}

// MAXSIM_ERANGE: not a shape (or an alignment) this kernel serves.
int launch_allpairs(const Params& p, int dt, bool argmax, hipStream_t st) {
  if (!allpairs_serves(dt, p.q_dtype, p.mask_dtype, p.nq, p.ncand, p.Lq, p.Ld, p.h)) return MAXSIM_ERANGE;
  if ((((uintptr_t)p.Q | (uintptr_t)p.index) & 15) != 0) return MAXSIM_ERANGE;
  AllPairsArgs a{};
  a.Q = p.Q; a.D = p.index; a.q_mask = p.q_mask; a.d_mask = p.d_mask;
  a.scores = p.scores; a.argmax = p.argmax;
  a.mask_dtype = p.mask_dtype; a.nq = p.nq; a.nd = p.ncand; a.Lq = p.Lq; a.Ld = p.Ld; a.h = p.h;
  if (dt == MAXSIM_F16) return argmax ? launch_dt<MAXSIM_F16, true>(a, st) : launch_dt<MAXSIM_F16, false>(a, st);
  return argmax ? launch_dt<MAXSIM_BF16, true>(a, st) : launch_dt<MAXSIM_BF16, false>(a, st);
}

}  // namespace maxsim
