// maxsim_launch.h -- host-side launch helpers shared by the translation units of libmaxsim, and the entry points each
// unit exports to the C-ABI file (the kernels are split over several .hip files only to compile them in parallel).
#pragma once
#include "maxsim_common.h"

namespace maxsim {

template <typename K>
inline int allow_lds(K kernel, int bytes) {
  if (bytes <= 64 * 1024) return MAXSIM_OK;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  return e == hipSuccess ? MAXSIM_OK : MAXSIM_ELAUNCH;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? MAXSIM_OK : MAXSIM_ELAUNCH; }

inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// Docs per wave for the streaming kernels: a wave's token stream should be long enough (~1.5k tokens) that the
// one partly filled last tile and the 16 KiB query-tile load are noise, short enough that the grid covers the
// 256 CUs several times over.  At most 64 docs per wave (scores are parked one per lane).
inline int pick_docs_per_wave(const Params& p, int waves) {
  double avg = p.n_docs > 0 ? (double)p.n_tokens / (double)p.n_docs : 1.0;
  if (avg < 1.0) avg = 1.0;
  int dpwv = (int)(1440.0 / avg + 0.5);
  if (dpwv < 1) dpwv = 1;
  if (dpwv > 64) dpwv = 64;
  while (dpwv > 1 && (int64_t)p.nq * ((p.ncand + dpwv * waves - 1) / (dpwv * waves)) < 2048) dpwv = (dpwv + 1) / 2;
  return dpwv;
}

// tu_stream.hip: the h = 128 register-query kernel.  index_dtype: MAXSIM_F32 / F16 / BF16 / F32_FAST / F32_BF16X3.
int launch_stream_rerank(Params& p, int index_dtype, hipStream_t st);
int launch_stream_dense_f32(Params& p, hipStream_t st);
// tu_bigh_rerank.hip / tu_bigh_dense.hip: the LDS-query kernel (any 16 <= h <= 1024); dt: MAXSIM_F32 / F16 / BF16.
// Return MAXSIM_ERANGE when the query image does not fit in LDS.
int launch_bigh_rerank(Params& p, int dt, hipStream_t st);
int launch_bigh_dense(Params& p, int dt, bool argmax, hipStream_t st);

}  // namespace maxsim
