#!/bin/bash
# On the GPU box: board power and clocks (rocm-smi) while bench.py loops the rerank kernel, per ablation variant.
for v in 0 1 2; do
  echo "== MAXSIM_VARIANT=$v"
  MAXSIM_VARIANT=$v python bench.py --steps 5000 --warmup 3 --no-cpu-baseline > gpurun_out/power_v$v.json 2>/dev/null &
  pid=$!
  sleep 12
  for i in 1 2 3 4 5 6; do
    kill -0 $pid 2>/dev/null || break
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo
    sleep 0.7
  done
  wait $pid
  python -c "import json; r=json.load(open('gpurun_out/power_v$v.json')); print('kernel_ms', r['roofline']['kernel_ms'])"
done
