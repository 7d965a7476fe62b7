#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
bash $R/tools/run_r4_profiles.sh "train retrieve" || exit 1
cd $R && python bench.py > gpurun_out/r4n_bench_default.json 2> gpurun_out/r4n_bench_default.err
