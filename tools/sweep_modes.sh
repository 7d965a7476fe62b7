#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for rep in 1 2; do for m in exact fast bf16x3; do echo "c2 fp32-mode=$m"; python bench.py --workload c2 --fp32-mode $m --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"; done; done
for m in exact fast bf16x3; do echo "ragged fp32-mode=$m"; python bench.py --workload ragged --fp32-mode $m --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
