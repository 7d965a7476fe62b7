// tu_bigh_dense_am.hip -- LDS-query streaming kernel, dense instantiations that record arg-max (training-form forward).
#include "maxsim_launch_bigh.h"

namespace maxsim {
int launch_bigh_dense_plain(Params& p, int dt, hipStream_t st);
int launch_bigh_dense(Params& p, int dt, bool argmax, hipStream_t st) {
  return argmax ? launch_bigh<MODE_DENSE, true>(p, dt, st) : launch_bigh_dense_plain(p, dt, st);
}
}  // namespace maxsim
