#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace + PMC passes of bench.py for the round's profiled workloads.
# usage: tools/run_profiles.sh <round-tag, e.g. r01>
set -u
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, bench args...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${name}_trace -- python3 $R/bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/${TAG}_${name}_trace.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_fetch -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_${name}_fetch.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_write -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_${name}_write.log 2>&1 || return 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_${name}_sq -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_${name}_sq.log 2>&1 || return 1
  echo "profiled $name"
}
C2="prof c2_f32 --workload c2 && prof c2_fp16 --workload c2 --index-dtype fp16 && prof c2_f32fast --workload c2 --fp32-mode fast && prof c2_f32bf16x3 --workload c2 --fp32-mode bf16x3"
if [ "${2:-all}" = "c2only" ]; then eval "$C2"; else eval "$C2" && prof c5_bf16 --workload c5 && prof c4_f32 --workload c4 && prof ragged_f32 --workload ragged; fi
