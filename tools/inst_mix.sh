#!/bin/bash
# On the GPU box: instruction mix of the rerank kernel (per-launch SQ instruction counters).  usage: tools/inst_mix.sh <bench args>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $R/gpurun_out/mix -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/mix.log 2>&1 || { tail -5 $R/gpurun_out/mix.log; exit 1; }
cd $R
python - <<'PY'
import csv, glob, os, collections
cc = max(glob.glob("gpurun_out/mix/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    if "k_maxsim" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
