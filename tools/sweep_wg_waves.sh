#!/bin/bash
# On the GPU box (diagnostic build, warm clocks): workgroups of 4 / 2 / 1 waves for the fp32 stream kernel, docs per wave fixed.
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
run() { python bench.py --workload $1 --steps 60 --warmup 15 --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for rep in 1 2 3; do
  for w in 4 2 1; do echo -n "c2 dpw=8 wg_waves=$w: "; MAXSIM_DPW=8 MAXSIM_WG_WAVES=$w run c2; done
  for w in 4 2 1; do echo -n "ragged dpw=12 wg_waves=$w: "; MAXSIM_DPW=12 MAXSIM_WG_WAVES=$w run ragged; done
done
