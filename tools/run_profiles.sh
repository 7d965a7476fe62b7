#!/bin/bash
# On the GPU box: rocprofv3 kernel-trace + PMC passes of bench.py for the round's profiled workloads.
# usage: tools/run_profiles.sh <round-tag, e.g. r03> "<names>"      names from: c2_f32 c2_fp16 c2_f32fast c2_f32bf16x3 c5_bf16
#        c4_f32 ragged_f32 ragged_f32bf16x3 ragged_fp16 dep768_fp16 mv128_fp16 mv768_fp16 c2_shard8_f32 (one rank's share of an 8-way doc-sharded job) single_query
set -u
TAG=${1:-r04}
NAMES=${2:-"c2_f32 c2_fp16 c2_f32bf16x3 c5_bf16 c4_f32 ragged_f32 dep768_fp16 c2_shard8_f32 single_query"}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# steps / warm-up per workload: ~0.5-1 s of GPU time in the traced run, the warm-ups covering the ~45 ms the clocks need to come
# back from idle (tools/probe_step_timeline.py) -- the kernel-trace statistics average EVERY call, warm-ups included, so a run of
# 3 + 20 launches (rounds 1-5a) reported the ramp: 4.17 ms for a kernel that holds 3.96-4.03
SW() { case $1 in c4_f32|mv128_fp16) echo "3000 500";; mv768_fp16) echo "800 60";; c2_fp16|ragged_fp16) echo "400 30";; c5_bf16|dep768_fp16) echo "60 5";; *) echo "200 15";; esac; }
prof() {  # name, bench args...
  local name=$1; shift
  set -- "$@" --no-cpu-baseline
  read S W <<< "$(SW $name)"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_${name}_trace -- python3 $R/bench.py "$@" --steps $S --warmup $W > $R/gpurun_out/${TAG}_${name}_trace.log 2>&1 || return 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_fetch -- python3 $R/bench.py "$@" --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_${name}_fetch.log 2>&1 || return 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_${name}_write -- python3 $R/bench.py "$@" --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_${name}_write.log 2>&1 || return 1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_${name}_sq -- python3 $R/bench.py "$@" --steps 5 --warmup $(( W < 12 ? 12 : W )) > $R/gpurun_out/${TAG}_${name}_sq.log 2>&1 || return 1
  # keep what tools/summarize_profile.py reads (gpurun merges at most 64 MiB back)
  find $R/gpurun_out/${TAG}_${name}_trace $R/gpurun_out/${TAG}_${name}_fetch $R/gpurun_out/${TAG}_${name}_write $R/gpurun_out/${TAG}_${name}_sq \
       -type f ! -name '*kernel_stats.csv' ! -name '*kernel_trace.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
  for f in $(find $R/gpurun_out/${TAG}_${name}_trace $R/gpurun_out/${TAG}_${name}_fetch $R/gpurun_out/${TAG}_${name}_write $R/gpurun_out/${TAG}_${name}_sq \
             -type f \( -name '*kernel_trace.csv' -o -name '*counter_collection.csv' \)); do
    grep -E "Kernel_Name|k_maxsim|k_topk" $f > $f.tmp; mv $f.tmp $f      # the library's kernels only (thousands of launches per traced run)
  done
  echo "profiled $name"
}
for n in $NAMES; do
  case $n in
    c2_f32) prof c2_f32 --workload c2 ;;
    c2_fp16) prof c2_fp16 --workload c2 --index-dtype fp16 ;;
    c2_f32fast) prof c2_f32fast --workload c2 --fp32-mode fast ;;
    c2_f32bf16x3) prof c2_f32bf16x3 --workload c2 --fp32-mode bf16x3 ;;
    c5_bf16) prof c5_bf16 --workload c5 ;;
    c4_f32) prof c4_f32 --workload c4 ;;
    ragged_f32) prof ragged_f32 --workload ragged ;;
    ragged_fp16) prof ragged_fp16 --workload ragged --index-dtype fp16 ;;
    ragged_f32bf16x3) prof ragged_f32bf16x3 --workload ragged --fp32-mode bf16x3 ;;
    dep768_fp16) prof dep768_fp16 --workload dep768 ;;
    mv128_fp16) prof mv128_fp16 --workload mv128 ;;
    mv768_fp16) prof mv768_fp16 --workload mv768 ;;
    c2_shard8_f32) prof c2_shard8_f32 --workload c2 --as-rank 3 --of 8 ;;
    single_query)  # the reference's online call: one rank_forward per query (tools/bench_small.py loops it)
      NDOCS=1000000 NQS=1,16 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_single_query_trace -- python3 $R/tools/bench_small.py > $R/gpurun_out/${TAG}_single_query_trace.log 2>&1 &&
      NDOCS=1000000 NQS=1,16 DTYPE=fp16 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_single_query_fp16_trace -- python3 $R/tools/bench_small.py > $R/gpurun_out/${TAG}_single_query_fp16_trace.log 2>&1 && echo "profiled single_query" ;;
    *) echo "unknown profile $n"; false ;;
  esac || exit 1
done
