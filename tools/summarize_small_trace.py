#!/usr/bin/env python3
"""rocprofv3 --kernel-trace of tools/bench_small.py (merged back under gpurun_out/) -> profiles/<tag>_summary.json: per
(kernel, grid) median / min / p90 duration.  usage: summarize_small_trace.py <out-tag> "<label>=<trace_dir>" ..."""
import csv, glob, json, os, sys
import numpy as np


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return max(f, key=os.path.getmtime) if f else None


out = {}
for arg in sys.argv[2:]:
    label, d = arg.split("=", 1)
    kt = one(os.path.join(d, "**", "*kernel_trace.csv"))
    per = {}
    for r in csv.DictReader(open(kt)):
        if "maxsim" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        key = f"{name} grid={r['Grid_Size_X']} wg={r['Workgroup_Size_X']}"
        per.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out[label] = {k: {"launches": len(v), "median_us": round(float(np.median(v)), 3), "min_us": round(min(v), 3),
                      "p90_us": round(float(np.percentile(v, 90)), 3)} for k, v in sorted(per.items())}
out["what"] = ("rocprofv3 --kernel-trace of tools/bench_small.py (NDOCS=1000000, 180-token docs, dim 128): the rerank kernel at 1 and 16 "
               "queries x 1000 candidates back to back, then rank_forward loops (rerank + k_topk_count) and batch top-k. "
               "Algorithmic bytes: 92.16 MB per query (fp32), 46.08 MB (fp16).")
json.dump(out, open(f"profiles/{sys.argv[1]}_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
