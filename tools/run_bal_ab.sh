#!/bin/bash
# On the GPU box: the token-balanced cut (BAL) of the static-grid rerank on ragged docs against the equal-count cut, interleaved
# (diagnostic library: MAXSIM_BAL=0 / 1; warm clocks: STEPS=200 WARMUP=50 by default).  usage: tools/run_bal_ab.sh "<bench args>" [reps]
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
ARGS=${1:-"--workload ragged --index-dtype fp16"}
for rep in $(seq 1 ${2:-3}); do
  for b in 0 1; do echo -n "$ARGS BAL=$b: "; MAXSIM_LIB=$PWD/tools/ab/diag.so MAXSIM_BAL=$b python bench.py $ARGS --steps ${STEPS:-200} --warmup ${WARMUP:-50} --no-cpu-baseline 2>/dev/null | python -c "$P"; done
done
