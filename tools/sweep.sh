#!/bin/bash
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for wl in "c2" "c2 --index-dtype fp16" "ragged" "c4" "c4 --lq 32" "c4 --index-dtype fp16"; do
  echo "workload=$wl"; python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline | python -c "$P"
done
