#!/bin/bash
# On the GPU box: MFMA busy share and held clock of a micro benchmark's kernels (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over
# GRBM_GUI_ACTIVE / 8 XCDs; clock = GRBM_GUI_ACTIVE / 8 / dispatch ns), last launches of every kernel.
# usage: tools/profile_micro_pmc.sh tools/micro/direct_vs_dma
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/micro_pmc
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/micro_pmc -- $R/$1 > $R/gpurun_out/micro_pmc.log 2>&1 || { tail -5 $R/gpurun_out/micro_pmc.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = max(glob.glob("$R/gpurun_out/micro_pmc/**/*counter_collection.csv", recursive=True))
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, c in per.items():
    g = sorted(c["GRBM_GUI_ACTIVE"])[-5:]
    m = sorted(c["SQ_VALU_MFMA_BUSY_CYCLES"])[-5:]
    if not g: continue
    clk = sum(v / 8 / ns for _, v, ns in g) / len(g)
    busy = sum(v for _, v, _ in m) / 1024 / (sum(v for _, v, _ in g) / 8) if m else 0
    print(f"{k[:60]:60s} ms {sum(ns for _, _, ns in g) / len(g) / 1e6:.3f}  clock {clk:.3f} GHz  MFMA busy {busy:.3f}")
PY
