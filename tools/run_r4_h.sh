#!/bin/bash
set -euo pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4h_gpu_tests.log 2>&1
python tools/bench_retrieve_step.py > gpurun_out/r4h_retrieve.log 2>&1
