#!/bin/bash
# On the GPU box: the fp32 stream kernel's knobs re-checked on WARM clocks (15 warm-up steps = 60 ms, 60 timed): earlier sweeps
# (steps 10, warm-up 2) sat on the clock ramp out of idle.  knobs live in the diagnostic build only:
#   MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
run() { python bench.py --workload $1 --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for rep in 1 2; do
  for v in 0 3 4; do echo -n "c2 variant=$v: "; MAXSIM_VARIANT=$v run c2; done
  for d in 3 4 6 8; do echo -n "c2 dpw=$d: "; MAXSIM_DPW=$d run c2; done
  for d in 6 8 12 16 24; do echo -n "ragged dpw=$d: "; MAXSIM_DPW=$d run ragged; done
  for v in 0 3 4; do echo -n "ragged variant=$v: "; MAXSIM_VARIANT=$v run ragged; done
done
