// tu_bigh_rerank.hip -- LDS-query streaming kernel, rerank mode instantiations.
#include "maxsim_launch_bigh.h"

namespace maxsim {
int launch_bigh_rerank(Params& p, int dt, hipStream_t st) { return launch_bigh<MODE_RERANK, false>(p, dt, st); }
}  // namespace maxsim
