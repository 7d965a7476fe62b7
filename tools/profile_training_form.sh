#!/bin/bash
# On the GPU box: per-kernel times of one forward + backward of the training-form operator (tools/bench_training_form.py
# under rocprofv3 --kernel-trace) -> gpurun_out/train_prof/kernels.txt; and the online call's latency breakdown for both
# index types -> gpurun_out/train_prof/latency_*.txt
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/train_prof
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 tools/bench_training_form.py --iters 8 > $O/kt.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("$O/kernels.txt", "w") as o:
    for r in rows[:14]:
        o.write("%-90s calls %4s avg %10.1f us  %5s %%\n" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(open("$O/kernels.txt").read())
PY
python3 tools/latency_breakdown.py 2>/dev/null > $O/latency_fp32.txt
DTYPE=fp16 python3 tools/latency_breakdown.py 2>/dev/null > $O/latency_fp16.txt
cat $O/latency_fp32.txt $O/latency_fp16.txt
