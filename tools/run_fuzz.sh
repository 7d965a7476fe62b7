#!/bin/bash
# On the GPU box: a longer hunt of tests/test_gpu_fuzz.py (MAXSIM_FUZZ_CASES cases per sweep, MAXSIM_FUZZ_SEED) -> gpurun_out/
set -u
CASES=${1:-400}; SEED=${2:-4}
mkdir -p gpurun_out
MAXSIM_FUZZ_CASES=$CASES MAXSIM_FUZZ_SEED=$SEED python -m pytest tests/test_gpu_fuzz.py -q -x -p no:cacheprovider > gpurun_out/fuzz_c${CASES}_s${SEED}.log 2>&1
tail -3 gpurun_out/fuzz_c${CASES}_s${SEED}.log
