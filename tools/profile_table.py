#!/usr/bin/env python3
"""Prints DESIGN.md section 6's table rows from the committed profiles: for every profiled workload the rerank kernel's
steady-state rocprofv3 duration (profiles/<tag>_<name>_pmc.json), the fraction of the 8 TB/s HBM peak that duration means for
the workload's ALGORITHMIC bytes (taken from the bench record profiles/<tag>_bench_builder_run_details.json: the same
accounting the bench line uses), PMC read bytes / algorithmic, MFMA busy and the held clock.
usage: tools/profile_table.py [tag]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
det = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_builder_run_details.json")))
alg = {"c2_f32": det["roofline"]["algorithmic_bytes_per_launch"]}
for o in det.get("other_workloads", []):
    alg[{"c2_bf16x3": "c2_f32bf16x3", "c2_fp16": "c2_fp16", "ragged": "ragged_f32", "ragged_bf16x3": "ragged_f32bf16x3",
         "ragged_fp16": "ragged_fp16", "c4": "c4_f32", "c5": "c5_bf16", "dep768": "dep768_fp16", "mv128": "mv128_fp16",
         "mv768": "mv768_fp16"}[o["key"]]] = o["algorithmic_bytes_per_launch"]
sh = det.get("sharded_share", {}).get("8")
if sh:
    alg["c2_shard8_f32"] = sh["algorithmic_bytes_per_launch"]
print("| profile | kernel | rocprof kernel ms | % of 8 TB/s | PMC read / algorithmic | MFMA busy | clock GHz |")
print("|---|---|---|---|---|---|---|")
for name, a in alg.items():
    f = os.path.join(ROOT, "profiles", f"{tag}_{name}_pmc.json")
    if not os.path.exists(f):
        continue
    best = None
    for k, v in json.load(open(f)).items():
        st = [x for x in v if x.startswith("kernel_trace_steady")]
        if "k_maxsim_stream" in k and st and (best is None or v[st[0]]["avg_ns"] > best[1][best[2]]["avg_ns"]):
            best = (k, v, st[0])
    if best is None:
        continue
    k, v, st = best
    ms = v[st]["avg_ns"] / 1e6
    rd = v.get("hbm_read_bytes_per_launch(2*FETCH_SIZE*1024)")
    mf = v.get("mfma_util(SQ_VALU_MFMA_BUSY_CYCLES/1024 / (GRBM_GUI_ACTIVE/8))")
    ck = v.get("effective_clock_GHz(GRBM_GUI_ACTIVE/8/ns)")
    print(f"| {tag}_{name} | `{k.replace('void maxsim::', '')}` | {ms:.4f} | {a / ms / 8e9 * 100:.1f} | {'' if rd is None else f'{rd / a:.4f}'} | "
          f"{'' if mf is None else f'{mf:.2f}'} | {'' if ck is None else f'{ck:.2f}'} |")
