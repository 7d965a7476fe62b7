#!/bin/bash
for v in 0 20 21 22 23 24 25; do for d in 4 8; do
  echo "variant=$v dpw=$d"; MAXSIM_F32_VARIANT=$v MAXSIM_DPW=$d python bench.py --steps 10 --warmup 2 --no-cpu-baseline --ndocs 300000 | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['roofline']['kernel_ms'], r['roofline']['achieved'])"
done; done
