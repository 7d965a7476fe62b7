#!/bin/bash
# On the GPU box (diagnostic build, warm clocks): fp16 index, workgroups of 4 waves (8 waves per CU) against 5 (10 per CU).
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
run() { python bench.py "$@" --index-dtype fp16 --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for rep in 1 2 3; do
  for w in 4 5; do echo -n "c2 fp16 dpw=8 wg_waves=$w: "; MAXSIM_DPW=8 MAXSIM_WG_WAVES=$w run --workload c2 --steps 120 --warmup 30; done
  for w in 4 5; do echo -n "ragged fp16 dpw=12 wg_waves=$w: "; MAXSIM_DPW=12 MAXSIM_WG_WAVES=$w run --workload ragged --steps 200 --warmup 50; done
done
