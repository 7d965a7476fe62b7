#!/bin/bash
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for v in 4 0 1 2; do echo "c4 variant=$v (4 = 32-column form)"; MAXSIM_VARIANT=$v python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline | python -c "$P"; done
for v in 4 0; do echo "c4 --lq 16 variant=$v"; MAXSIM_VARIANT=$v python bench.py --workload c4 --lq 16 --steps 20 --warmup 3 --no-cpu-baseline | python -c "$P"; done
for v in 4 0; do echo "c2 --lq 16 variant=$v"; MAXSIM_VARIANT=$v python bench.py --workload c2 --lq 16 --ndocs 400000 --steps 20 --warmup 3 --no-cpu-baseline | python -c "$P"; done
