#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
# On the GPU box: per-document overhead -- the same number of candidate tokens per query cut into docs of different length.
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for cfg in "32 5632 5625000" "45 4000 4000000" "90 2000 2000000" "128 1408 1400000" "180 1000 1000000" "192 936 936000" "360 500 500000" "720 250 250000"; do
  set -- $cfg
  for extra in "" "--index-dtype fp16"; do
  echo -n "ld=$1 ncand=$2 $extra: "; python bench.py --workload c2 --ld $1 --ncand $2 --ndocs $3 $extra --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
  done
done
