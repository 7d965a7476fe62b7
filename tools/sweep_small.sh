#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
# On the GPU box: docs-per-wave sweep at small batches (where the grid does not fill the chip many times over).
for nq in 1 4 8 16 32 64; do for d in 0 1 2 3 4 6 8; do
  echo -n "nq=$nq dpw=$d: "; MAXSIM_DPW=$d python bench.py --nq $nq --ndocs 200000 --steps 100 --no-cpu-baseline 2>/dev/null | python tools/ms.py
done; done
