#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
# On the GPU box: how the achieved bandwidth depends on the batch (launch ramp / tail vs steady state).
P='import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["roofline"]["kernel_ms"], r["roofline"]["achieved"])'
for wl in "c4" "c2"; do for nq in 64 256 1024 4096; do
  echo "workload=$wl nq=$nq"; python bench.py --workload $wl --nq $nq --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "$P"
done; done
