#!/bin/bash
set -euo pipefail
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r4b_gpu_tests.log 2>&1
python tools/probe_unique_pids.py > gpurun_out/r4b_probe_unique.log 2>&1
MAXSIM_BENCH_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --ndocs 100000 --steps 5 --warmup 2 > gpurun_out/r4b_bench_gloo2.json 2> gpurun_out/r4b_bench_gloo2.err
