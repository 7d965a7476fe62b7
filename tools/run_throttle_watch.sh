#!/bin/bash
# On the GPU box: the firmware's throttler residencies (tools/micro/throttle_watch, one line per second) beside a long loop of
# one bench workload -> gpurun_out/<tag>_throttle_watch.txt.   tools/run_throttle_watch.sh <tag> [workload] [steps] [extra bench args]
set -uo pipefail
TAG=${1:-r5}; WL=${2:-c2}; STEPS=${3:-2500}; shift 3 2>/dev/null || true
mkdir -p gpurun_out
rm -f gpurun_out/throttle_stop
tools/micro/throttle_watch 500 10 gpurun_out/throttle_stop > gpurun_out/${TAG}_throttle_watch.txt 2>&1 &
W=$!
python bench.py --workload $WL --no-cpu-baseline --steps $STEPS --warmup 5 "$@" > gpurun_out/${TAG}_throttle_bench.json 2> gpurun_out/${TAG}_throttle_bench.err
echo "bench rc=$?"
sleep 3
touch gpurun_out/throttle_stop
wait $W
rm -f gpurun_out/throttle_stop
tail -c 700 gpurun_out/${TAG}_throttle_bench.json; echo
grep -c . gpurun_out/${TAG}_throttle_watch.txt
