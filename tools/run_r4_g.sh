#!/bin/bash
set -euo pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_fuzz.py -x -q -m gpu -k "backward or autograd or dense_score or pids or training or grad" > gpurun_out/r4g_tests.log 2>&1
python tools/bench_training_form.py --iters 10 --no-torch > gpurun_out/r4g_train.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04_train_trace2 -- python3 $GRAFT_REPO_ROOT/tools/bench_training_form.py --iters 6 --no-torch > $GRAFT_REPO_ROOT/gpurun_out/r04_train_trace2.log 2>&1
