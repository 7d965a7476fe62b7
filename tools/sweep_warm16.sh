#!/bin/bash
# On the GPU box: the 16-bit kernels' knobs on WARM clocks (see sweep_warm.sh).  Diagnostic build:
#   MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for rep in 1 2; do
  for v in 0 3; do echo -n "c2 fp16 variant=$v: "; MAXSIM_VARIANT=$v run --workload c2 --index-dtype fp16 --steps 120 --warmup 30; done
  for d in 4 6 8 12 16; do echo -n "c2 fp16 dpw=$d: "; MAXSIM_DPW=$d run --workload c2 --index-dtype fp16 --steps 120 --warmup 30; done
  for b in 1 0; do echo -n "ragged fp16 bal=$b: "; MAXSIM_BAL=$b run --workload ragged --index-dtype fp16 --steps 200 --warmup 50; done
  for d in 8 12 16 24; do echo -n "ragged fp16 dpw=$d: "; MAXSIM_DPW=$d run --workload ragged --index-dtype fp16 --steps 200 --warmup 50; done
  for s in 0 81 82 121; do echo -n "dep768 shape=$s: "; MAXSIM_BIGH_SHAPE=$s run --workload dep768 --steps 30 --warmup 5; done
  for b in 1 0; do echo -n "dep768 bal=$b: "; MAXSIM_BAL=$b run --workload dep768 --steps 30 --warmup 5; done
  for s in 0 81 121 141; do echo -n "mv768 shape=$s: "; MAXSIM_BIGH_SHAPE=$s run --workload mv768 --steps 400 --warmup 60; done
  for hq in 1 0; do echo -n "mv768 halfq=$hq: "; MAXSIM_HALFQ=$hq run --workload mv768 --steps 400 --warmup 60; done
  for u in 0 1 2; do echo -n "mv128 uni16 shape=$u: "; MAXSIM_UNI16_SHAPE=$u run --workload mv128 --steps 3000 --warmup 600; done
done
