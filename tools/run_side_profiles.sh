#!/bin/bash
# On the GPU box: rocprofv3 kernel statistics + FETCH/WRITE counter passes (separate runs) of the side kernels:
#   retrieve : tools/bench_retrieve_step.py (k_unique_pids, work-list scan/fill, counted rerank, k_topk, k_shard_candidates)
#   train    : tools/bench_training_form.py (k_maxsim_allpairs forward, k_maxsim_bwd_dq_v8 / _index / _dd_rows backward)
# usage: tools/run_side_profiles.sh <round-tag> "<retrieve train>"
set -u
R=$GRAFT_REPO_ROOT
TAG=${1:-r05}
NAMES=${2:-"retrieve train"}
cd /tmp && export TMPDIR=/tmp
for n in $NAMES; do
  case $n in
    retrieve) CMD="python3 $R/tools/bench_retrieve_step.py" ;;
    train) CMD="python3 $R/tools/bench_training_form.py --iters 60 --no-torch" ;;   # (60 iterations: ~0.25 s, off the clock ramp)
    *) echo "unknown $n"; exit 1 ;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${n}_trace -- $CMD > $R/gpurun_out/${TAG}_${n}_trace.log 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${n}_fetch -- $CMD > $R/gpurun_out/${TAG}_${n}_fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${n}_write -- $CMD > $R/gpurun_out/${TAG}_${n}_write.log 2>&1 || exit 1
  echo "profiled $n"
done
