#!/bin/bash
# knobs live in the diagnostic build only: MAXSIM_OUT=tools/ab/diag.so colbert_amd/csrc/build.sh -DMAXSIM_DIAG
export MAXSIM_LIB=${MAXSIM_LIB:-$PWD/tools/ab/diag.so}
# On the GPU box: board power / shader clock while bench.py loops a workload.  usage: tools/power_probe.sh "<bench args>" ...
for args in "$@"; do
  echo "== $args"
  python bench.py $args --steps 6000 --warmup 3 --no-cpu-baseline > gpurun_out/pp.json 2>/dev/null &
  pid=$!
  sleep 9
  for i in 1 2 3; do
    kill -0 $pid 2>/dev/null || break
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
    sleep 0.6
  done
  wait $pid
  python tools/ms.py < gpurun_out/pp.json
done
