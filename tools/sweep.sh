#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
# On the GPU box: A/B sweeps through bench.py (kernel ms and algorithmic GB/s per configuration).
# usage: tools/sweep.sh            -> every workload, default kernels
#        MAXSIM_VARIANT=1|2 ...    -> ablation builds of the h=128 stream kernel (DESIGN.md "Tuning knobs")
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for wl in "c2" "c2 --index-dtype fp16" "c2 --fp32-mode fast" "ragged" "c4" "c4 --index-dtype fp16" "c5"; do
  echo "workload=$wl"; python bench.py --workload $wl --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "$P"
done
