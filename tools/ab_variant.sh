#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
# On the GPU box: kernel variants (MAXSIM_VARIANT) side by side.  usage: tools/ab_variant.sh "<variants>" "<bench args>" ...
VARS="$1"; shift
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for args in "$@"; do for rep in 1 2; do for v in $VARS; do
  echo -n "[$args] variant $v: "; MAXSIM_VARIANT=$v python bench.py $args --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
done; done; done
