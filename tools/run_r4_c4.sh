#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $R/gpurun_out/r04_c4t_$tag -- python3 $R/tools/probe_c4_traffic.py > $R/gpurun_out/r04_c4t_$tag.log 2>&1 || echo "pass $tag failed"
done
# the bench.py form the r03 profile was taken from (1 warm-up + 3 steps of the c4 workload)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r04_c4t_benchform -- python3 $R/bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r04_c4t_benchform.log 2>&1
echo done
