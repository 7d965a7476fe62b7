#!/bin/bash
# On the GPU box: the default bench line + the --gpus 4 rehearsal with every rank on the one GPU over gloo (reduced index).
set -uo pipefail
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/${1:-r5}_bench.json 2> gpurun_out/${1:-r5}_bench.err
echo "bench rc=$?"; tail -c 3500 gpurun_out/${1:-r5}_bench.json
cp gpurun_out/bench_details.json gpurun_out/${1:-r5}_bench_details.json 2>/dev/null
MAXSIM_BENCH_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 4 --ndocs 100000 --steps 5 --warmup 2 > gpurun_out/${1:-r5}_bench_gloo4.json 2> gpurun_out/${1:-r5}_bench_gloo4.err
echo "gloo4 rc=$?"; tail -c 2500 gpurun_out/${1:-r5}_bench_gloo4.json; tail -5 gpurun_out/${1:-r5}_bench_gloo4.err
