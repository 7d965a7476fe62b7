#!/bin/bash
# On the GPU box: the token-balanced cut (BAL) of the static-grid rerank on ragged docs against the equal-count cut, interleaved
# (diagnostic library: MAXSIM_BAL=0 / 1).  usage: tools/run_bal_ab.sh
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
for args in "--workload ragged" "--workload ragged --index-dtype fp16" "--workload ragged --fp32-mode bf16x3"; do
  for rep in 1 2; do
    for b in 0 1; do echo -n "$args BAL=$b: "; MAXSIM_LIB=$PWD/tools/ab/diag.so MAXSIM_BAL=$b python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
  done
done
