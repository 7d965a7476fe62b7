#!/bin/bash
# On the GPU box: the whole -m gpu suite, output to gpurun_out/<tag>_gpu_tests.log
set -euo pipefail
TAG=${1:-run}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1
