#!/bin/bash
# On the GPU box: the dim-768 ring-shape rules (4 x 2 against 8 x 1 for a two-piece query image and for short docs) on warm
# clocks, interleaved.  Diagnostic build: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for rep in 1 2 3; do
  for s in 42 81 0; do echo -n "dep768 (fp32 query, two-piece image) shape=$s: "; MAXSIM_BIGH_SHAPE=$s run --workload dep768 --steps 30 --warmup 5; done
  for s in 42 81 0; do echo -n "mv768 (fp32 query) shape=$s: "; MAXSIM_BIGH_SHAPE=$s run --workload mv768 --steps 400 --warmup 60; done
  for s in 42 81 0; do echo -n "mv768 (fp16 query, one-piece image) shape=$s: "; MAXSIM_BIGH_SHAPE=$s run --workload mv768 --q-dtype fp16 --steps 400 --warmup 60; done
done
