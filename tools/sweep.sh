#!/bin/bash
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for rep in 1 2; do
for lib in tools/ab/libmaxsim_v1.so colbert_amd/libmaxsim.so; do
  echo "workload=c2 lib=$lib"; MAXSIM_LIB=$PWD/$lib python bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --ndocs 400000 | python -c "$P"
done; done
for wl in "c2 --index-dtype fp16" "ragged" "c4" "c4 --index-dtype fp16" "c5"; do
  echo "workload=$wl"; python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline | python -c "$P"
done
