#!/usr/bin/env python3
"""Joins tools/probe_c4_traffic.py's PHASES line with the rocprofv3 counter_collection.csv of the same run:
per phase and counter the per-launch values (rerank dispatches matched by kernel name, in dispatch order).
    python tools/summarize_c4_traffic.py <log> <pmc dir> [<log> <pmc dir> ...] > profiles/r04_c4_traffic.json"""
import csv
import glob
import json
import os
import sys

out = []
for log, d in zip(sys.argv[1::2], sys.argv[2::2]):
    meta = json.loads([l for l in open(log) if l.startswith("PHASES ")][-1][7:])
    f = max(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "k_maxsim_stream" in r["Kernel_Name"]]
    counters = sorted({r["Counter_Name"] for r in rows})
    res = {"run": os.path.basename(d), "index_ptr_mod_4096": meta["index_ptr_mod_4096"], "kernel": rows[0]["Kernel_Name"][:60], "phases": []}
    for c in counters:
        seq = [float(r["Counter_Value"]) for r in sorted((r for r in rows if r["Counter_Name"] == c), key=lambda r: int(r["Dispatch_Id"]))]
        at = 0
        for i, ph in enumerate(meta["phases"]):
            v = seq[at:at + ph["launches"]]
            at += ph["launches"]
            if len(res["phases"]) <= i:
                res["phases"].append(dict(ph))
            e = res["phases"][i]
            e[c] = [round(x, 1) for x in v]
            if c == "FETCH_SIZE":
                e["read_bytes(2*FETCH_SIZE*1024)"] = [int(2 * x * 1024) for x in v]
                e["read_over_algorithmic"] = [round(2 * x * 1024 / ph["algorithmic"], 4) for x in v]
                e["read_over_distinct_doc_bytes"] = [round(2 * x * 1024 / (dd * 4096 + 0.0), 4) for x, dd in zip(v, ph["distinct_docs"])]
        assert at == len(seq), (c, at, len(seq))
    out.append(res)
print(json.dumps(out, indent=1))
