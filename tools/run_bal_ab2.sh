#!/bin/bash
# as run_bal_ab.sh with the order inside a pair reversed (1 first), to rule out an order effect
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["roofline"]["kernel_ms"], r["roofline"]["frac"])'
ARGS=${1:-"--workload dep768"}
for rep in $(seq 1 ${2:-3}); do
  for b in 1 0; do echo -n "$ARGS BAL=$b: "; MAXSIM_LIB=$PWD/tools/ab/diag.so MAXSIM_BAL=$b python bench.py $ARGS --steps ${STEPS:-30} --warmup ${WARMUP:-6} --no-cpu-baseline 2>/dev/null | python -c "$P"; done
done
