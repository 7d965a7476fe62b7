// tu_bigh_dense.hip -- LDS-query streaming kernel, dense (all-pairs, masks) instantiations without arg-max.
#include "maxsim_launch_bigh.h"

namespace maxsim {
int launch_bigh_dense_plain(Params& p, int dt, hipStream_t st) { return launch_bigh<MODE_DENSE, false>(p, dt, st); }
}  // namespace maxsim
