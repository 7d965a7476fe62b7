#!/bin/bash
# round 4, call A: the new sharded-from-files tests + whole gpu suite + a 2-rank rehearsal of bench.py on one GPU (gloo)
set -euo pipefail
mkdir -p gpurun_out
python -m pytest tests/test_sharded_files.py -x -q -m gpu > gpurun_out/r4a_sharded_files.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r4a_gpu_tests.log 2>&1
MAXSIM_BENCH_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --ndocs 100000 --steps 5 --warmup 2 > gpurun_out/r4a_bench_gloo2.json 2> gpurun_out/r4a_bench_gloo2.err
tail -c 1500 gpurun_out/r4a_bench_gloo2.json
