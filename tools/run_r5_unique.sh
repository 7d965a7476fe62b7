#!/bin/bash
# On the GPU box: ids -> distinct pids with / without the block-level de-duplication (diagnostic build), then the tests.
set -uo pipefail
for v in 0; do
  echo "== shipped kernel (diag build)"
  MAXSIM_LIB=tools/ab/diag.so MAXSIM_UNIQUE_BLOCKS=$v python tools/probe_unique_pids.py 2>&1 | grep -v amdgpu.ids
done
echo "== retrieve step (shipped library)"
python tools/bench_retrieve_step.py 2>&1 | tail -3
python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_parity.py -x -q -k "ids or pids or retrieve" 2>&1 | tail -3
MAXSIM_FUZZ_CASES=300 python -m pytest tests/test_gpu_fuzz.py -x -q -k "retrieve or ids" 2>&1 | tail -3
