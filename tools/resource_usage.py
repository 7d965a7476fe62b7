#!/usr/bin/env python3
"""Compiles one translation unit of libmaxsim with -Rpass-analysis=kernel-resource-usage and prints one line per
kernel: demangled name, VGPRs + AGPRs, SGPRs, scratch, LDS, occupancy.  `tools/resource_usage.py tu_stream [-D...]`.
Used to check that an edit outside the hot loop did not move the hot kernels' register allocation."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tu = sys.argv[1] if len(sys.argv) > 1 else "tu_stream"
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
       "-I" + os.path.join(ROOT, "colbert_amd", "csrc"), "-fPIC", "-c", os.path.join(ROOT, "colbert_amd", "csrc", tu + ".hip"),
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m2 = re.search(r"remark: +(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (.*?) \[-Rpass", line)
    if not m2:
        continue
    k, v = m2.group(1), m2.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for r, d in sorted(zip(rows, dem), key=lambda x: x[1]):
    d = re.sub(r"\(.*", "", d).replace("void maxsim::", "")
    print(f"{d:60s} v{r.get('VGPRs', '?'):>4} a{r.get('AGPRs', '?'):>3} s{r.get('SGPRs', '?'):>4} scratch {r.get('ScratchSize', '?'):>3} "
          f"lds {r.get('LDS', '?'):>6} occ {r.get('Occupancy', '?')}")
