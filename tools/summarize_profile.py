#!/usr/bin/env python3
"""Condenses rocprofv3 output merged back under gpurun_out/ into small tracked files under profiles/.

    python tools/summarize_profile.py <tag> <trace_dir> [--fetch DIR] [--write DIR] [--sq DIR] [--kernels REGEX] [--skip N]

--kernels: which kernels the per-launch PMC / steady-state summary covers (default: k_maxsim); --skip: warm-up launches
dropped from the steady-state average (default 3); --sq-skip: launches dropped from the front of the SQ pass's lists
(default 0; tools/run_profiles.sh runs that pass with >= 12 warm-ups so that the counted launches are off the clock ramp).

Writes profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats summary, kernel names shortened),
and profiles/<tag>_pmc.json (per-launch PMC values of the maxsim kernels).  HBM traffic follows
MI355X_MICROARCH.md "HBM": FETCH_SIZE is in KiB-ish units of 1024 B and reports exactly 1/2 of the bytes of a
wide coalesced 16-B/lane stream on gfx950, so read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 is exact.
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name if len(name) <= 90 else name[:87] + "..."


def one(pattern):
    f = glob.glob(pattern, recursive=True)  # a directory merged over several runs holds one file set per run: newest wins
    return max(f, key=os.path.getmtime) if f else None


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    opts = dict(zip(sys.argv[3::2], sys.argv[4::2]))
    os.makedirs("profiles", exist_ok=True)
    want = re.compile(opts.get("--kernels", "k_maxsim"))
    skip = int(opts.get("--skip", 3))
    ks = one(os.path.join(trace, "**", "*kernel_stats.csv"))
    if ks:
        rows = list(csv.DictReader(open(ks)))
        with open(f"profiles/{tag}_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows[:int(opts.get('--rows', 12))]:
                w.writerow([short(r["Name"])] + [r[k] for k in ("Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
    # steady-state average of the maxsim kernels from the per-dispatch trace (first 3 launches = bench warm-up dropped)
    steady = {}
    kt = one(os.path.join(trace, "**", "*kernel_trace.csv"))
    if kt:
        per = {}
        for r in csv.DictReader(open(kt)):
            if want.search(r["Kernel_Name"]):
                per.setdefault(short(r["Kernel_Name"]).split("(")[0], []).append(
                    (int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        for k, v in per.items():
            d = [x[1] for x in sorted(v)][skip:]
            if d:
                steady[k] = {"launches": len(d), "avg_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}
    out = {}
    sq_skip = int(opts.get("--sq-skip", 0))
    for key, flag in (("fetch", "--fetch"), ("write", "--write"), ("sq", "--sq")):
        d = opts.get(flag)
        if not d:
            continue
        cc = one(os.path.join(d, "**", "*counter_collection.csv"))
        if not cc:
            continue
        rows_cc = sorted(csv.DictReader(open(cc)), key=lambda r: int(r["Dispatch_Id"]))
        if key == "sq" and sq_skip:
            ids = sorted({int(r["Dispatch_Id"]) for r in rows_cc if want.search(r["Kernel_Name"])})
            drop = set(ids[:sq_skip])
            rows_cc = [r for r in rows_cc if int(r["Dispatch_Id"]) not in drop]
        for r in rows_cc:
            if not want.search(r["Kernel_Name"]):
                continue
            k = short(r["Kernel_Name"]).split("(")[0]
            e = out.setdefault(k, {}).setdefault(r["Counter_Name"], [])
            e.append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                out[k].setdefault("dispatch_ns(sq pass)", []).append(dur)
    summ = {}
    for k, cs in out.items():
        s = {c: sum(v) / len(v) for c, v in cs.items()}
        s["launches_sampled"] = {c: len(v) for c, v in cs.items()}
        if "FETCH_SIZE" in s:
            s["hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)"] = 2 * s["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in s:
            s["hbm_write_bytes_per_launch(WRITE_SIZE*1024)"] = s["WRITE_SIZE"] * 1024
        if "GRBM_GUI_ACTIVE" in s and "dispatch_ns(sq pass)" in s:
            s["effective_clock_GHz(GRBM_GUI_ACTIVE/8/ns)"] = s["GRBM_GUI_ACTIVE"] / 8 / s["dispatch_ns(sq pass)"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in s:
            # busy cycles summed over 1024 SIMDs / (cycles per XCD-summed GUI_ACTIVE / 8)
            s["mfma_util(SQ_VALU_MFMA_BUSY_CYCLES/1024 / (GRBM_GUI_ACTIVE/8))"] = (s["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (s["GRBM_GUI_ACTIVE"] / 8)
        summ[k] = s
    for k, v in steady.items():
        summ.setdefault(k, {})[f"kernel_trace_steady(after {skip} warm-up launches)"] = v
    with open(f"profiles/{tag}_pmc.json", "w") as f:
        json.dump(summ, f, indent=1, sort_keys=True)
    print(json.dumps(summ, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
