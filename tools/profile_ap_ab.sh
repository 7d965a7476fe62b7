#!/bin/bash
# On the GPU box: SQ counter passes of the all-pairs forward for several builds side by side.
# usage: tools/profile_ap_ab.sh lib1.so lib2.so ...   -> gpurun_out/ap_ab/<lib>_<pass>/ + a table on stdout
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/ap_ab
rm -rf $O; mkdir -p $O
for l in "$@"; do
  name=$(basename $l .so)
  export MAXSIM_LIB=$R/$l
  N=6 timeout -k 10 200 rocprofv3 --kernel-include-regex allpairs --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $O/${name}_p1 -- python3 $R/tools/bench_allpairs_fwd.py > $O/${name}_p1.log 2>&1 || echo "pass1 failed for $name"
  N=6 timeout -k 10 200 rocprofv3 --kernel-include-regex allpairs --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY --output-format csv -d $O/${name}_p2 -- python3 $R/tools/bench_allpairs_fwd.py > $O/${name}_p2.log 2>&1 || echo "pass2 failed for $name"
  N=6 timeout -k 10 200 rocprofv3 --kernel-include-regex allpairs --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/${name}_p3 -- python3 $R/tools/bench_allpairs_fwd.py > $O/${name}_p3.log 2>&1 || echo "pass3 failed for $name"
done
unset MAXSIM_LIB
python3 - "$O" "$@" <<'PY'
import csv, glob, os, sys
O = sys.argv[1]
for l in sys.argv[2:]:
    name = os.path.basename(l)[:-3]
    vals = {}
    for p in ("p1", "p2", "p3"):
        for f in glob.glob(os.path.join(O, f"{name}_{p}", "**", "*counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            byc = {}
            for r in rows:
                byc.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            for c, v in byc.items():
                v = sorted(v)[3:]                      # (the first 3 launches are warm-ups)
                if v:
                    vals[c] = sum(x[1] for x in v) / len(v)
                    vals["ns_" + p] = sum(x[2] for x in v) / len(v)
    if "GRBM_GUI_ACTIVE" in vals:
        vals["clock_GHz"] = vals["GRBM_GUI_ACTIVE"] / 8 / vals["ns_p1"]
        vals["mfma_busy"] = vals.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (vals["GRBM_GUI_ACTIVE"] / 8)
    print(name, {k: (round(v, 3) if v < 100 else int(v)) for k, v in sorted(vals.items())})
PY
