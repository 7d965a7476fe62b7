#!/bin/bash
# On the GPU box (diagnostic build, warm clocks, interleaved): the 16x16x32 matrix shape (shipped) against 32x32x16
# (MAXSIM_VARIANT=4) on the fp16 index and in the 3 x bf16 mode of an fp32 index -- round 3 measured +2-3 % on the clock ramp.
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
run() { python bench.py "$@" --no-cpu-baseline 2>/dev/null | python -c "$P"; }
for rep in 1 2 3; do
  for v in 0 4; do echo -n "c2 fp16 variant=$v: "; MAXSIM_VARIANT=$v run --workload c2 --index-dtype fp16 --steps 120 --warmup 30; done
  for v in 4 0; do echo -n "c2 bf16x3 variant=$v: "; MAXSIM_VARIANT=$v run --workload c2 --fp32-mode bf16x3 --steps 60 --warmup 15; done
  for v in 0 4; do echo -n "ragged bf16x3 variant=$v: "; MAXSIM_VARIANT=$v run --workload ragged --fp32-mode bf16x3 --steps 80 --warmup 20; done
done
