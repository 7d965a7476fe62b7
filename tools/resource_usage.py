#!/usr/bin/env python3
"""Compiles one translation unit of libmaxsim with -Rpass-analysis=kernel-resource-usage and prints one line per
kernel: demangled name, VGPRs, AGPRs, SGPRs, scratch, static LDS, occupancy.

    tools/resource_usage.py tu_stream [-D...]          table on stdout
    tools/resource_usage.py --write-table              regenerates tests/golden/kernel_resources.json (the guarded kernels)

Why: the speed of the hot kernels hangs on hipcc's register assignment (docs/experiments.md records -16 % twice when an
innocuous edit moved the fp32 rerank kernel from 186 to 214-244 VGPRs).  tests/test_kernel_resources.py compiles the
same units and compares with the committed table, so an edit (or a toolchain bump) that moves a guarded kernel fails the
CPU suite instead of silently costing the headline."""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "golden", "kernel_resources.json")
FIELDS = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch",
          "LDS Size [bytes/block]": "lds_static", "Occupancy [waves/SIMD]": "waves_per_simd", "VGPRs Spill": "vgpr_spill",
          "SGPRs Spill": "sgpr_spill"}
# translation unit -> regular expressions of the kernels the test guards (demangled, without the argument list)
GUARDED = {
    "tu_stream": [r"k_maxsim_stream<0, [0-4], 4, [12], 0, (32|48), false, (false|true), (false|true)>",
                  r"k_maxsim_stream<0, 0, 4, 1, 0, 16, false, (false|true), false>",
                  r"k_maxsim_stream<1, 0, 4, 1, 0, 32, false, false, false>", r"k_maxsim_stream_uni<[48], [12], (4|8|16), 0, (false|true)>",
                  r"k_maxsim_stream_uni16<.*>", r"k_maxsim_stream_f32h<.*>"],
    "tu_bigh_rerank": [r"k_maxsim_stream_bigh<0, [012], [12], (4|8|12), [12], false, 1, false, false, false, (false|true), (false|true)>", r"k_maxsim_bigh_uni<.*>"],
    "tu_allpairs": [r"k_maxsim_allpairs<[12], [123], [34], (false|true)>"],
}


def collect(tu, extra=()):
    """-> {demangled kernel name: {vgpr, agpr, sgpr, scratch, lds_static, waves_per_simd, vgpr_spill, sgpr_spill}}"""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "colbert_amd", "csrc"), "-fPIC", "-c", os.path.join(ROOT, "colbert_amd", "csrc", tu + ".hip"),
           "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + list(extra)
    run = subprocess.run(cmd, capture_output=True, text=True)
    if run.returncode != 0:
        raise RuntimeError(f"hipcc failed on {tu}:\n{run.stderr[-2000:]}")
    rows, cur = [], None
    for line in run.stderr.splitlines():
        m = re.search(r"remark: +([A-Za-z][A-Za-z \[\]/]*?): (.*?) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None and k in FIELDS:
            cur[FIELDS[k]] = int(v)
    dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    out = {}
    for r, d in zip(rows, dem):
        d = re.sub(r"\(.*", "", d).replace("void maxsim::", "").replace("maxsim::", "")
        out[d] = {k: v for k, v in r.items() if k != "name"}
    return out


def guarded_table():
    """What tests/test_kernel_resources.py compares: every guarded kernel of the three hot translation units."""
    with ThreadPoolExecutor(max_workers=len(GUARDED)) as ex:
        per_tu = dict(zip(GUARDED, ex.map(collect, GUARDED)))
    table = {}
    for tu, pats in GUARDED.items():
        keep = {k: v for k, v in per_tu[tu].items() if any(re.fullmatch(p, k) for p in pats)}
        table[tu] = dict(sorted(keep.items()))
    return table


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--write-table":
        table = guarded_table()
        ver = subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
        with open(TABLE, "w") as f:
            json.dump({"toolchain": ver, "flags": "-O3 --offload-arch=gfx950 -std=c++17", "kernels": table}, f, indent=1, sort_keys=True)
            f.write("\n")
        print(f"wrote {os.path.relpath(TABLE, ROOT)}: {sum(len(v) for v in table.values())} kernels")
        return
    tu = sys.argv[1] if len(sys.argv) > 1 else "tu_stream"
    for d, r in sorted(collect(tu, sys.argv[2:]).items()):
        print(f"{d:66s} v{r.get('vgpr', '?'):>4} a{r.get('agpr', '?'):>3} s{r.get('sgpr', '?'):>4} scratch {r.get('scratch', '?'):>3} "
              f"lds {r.get('lds_static', '?'):>6} occ {r.get('waves_per_simd', '?')}")


if __name__ == "__main__":
    main()
