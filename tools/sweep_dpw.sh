#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
# On the GPU box: docs-per-wave sweep (MAXSIM_DPW) at the metric's batch of 256 queries.
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for d in 4 8 16 32 64; do echo "c4 dpw=$d"; MAXSIM_DPW=$d python bench.py --workload c4 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
for d in 2 3 4 6 8 12; do echo "c2 dpw=$d"; MAXSIM_DPW=$d python bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
for d in 4 8 12 16; do echo "ragged dpw=$d"; MAXSIM_DPW=$d python bench.py --workload ragged --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
