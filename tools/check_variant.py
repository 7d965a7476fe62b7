#!/usr/bin/env python3
"""On the GPU box: scores of an alternative kernel variant (MAXSIM_VARIANT=<n>, diagnostic build) against the default
kernel and the float64 closed form, on ragged docs.   python tools/check_variant.py 5
(knobs are read once per process: the default kernel's scores come from a child process without the variable)"""
import os
import subprocess
import sys
import tempfile

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd  # noqa: E402

v = sys.argv[1]
os.environ.setdefault("MAXSIM_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ab", "diag.so"))
dump = os.environ.get("CHECK_VARIANT_DUMP")
if not dump:
    os.environ["MAXSIM_VARIANT"] = v
gen = torch.Generator().manual_seed(5)
ndocs = 4000
doclens = torch.randint(0, 181, (ndocs,), generator=gen).tolist()
part = F.normalize(torch.randn(sum(doclens), 128, generator=gen), dim=-1)
r = colbert_amd.ColbertRanker(parts=[part], parts_doclens=[doclens], dim=128, index_dtype=torch.float32)
Q = F.normalize(torch.randn(8, 32, 128, generator=gen), dim=-1)
cand = torch.randint(-1, ndocs, (8, 777), generator=gen)
q_len = torch.tensor([32, 31, 17, 16, 5, 1, 32, 20], dtype=torch.int32)
alt = r.score_candidates(Q, cand, q_len=q_len).cpu()
if dump:                                   # child: the default kernel's scores
    torch.save(alt, dump)
    sys.exit(0)
with tempfile.TemporaryDirectory() as td:
    env = dict(os.environ, CHECK_VARIANT_DUMP=os.path.join(td, "base.pt"))
    env.pop("MAXSIM_VARIANT", None)
    subprocess.run([sys.executable, os.path.abspath(__file__), v], env=env, check=True)
    base = torch.load(os.path.join(td, "base.pt"))
fin = torch.isfinite(base)
assert torch.equal(fin, torch.isfinite(alt))
print("max |alt - base| =", float((alt[fin] - base[fin]).abs().max()), " bitwise equal:", bool(torch.equal(alt, base)))
D = part.double()
offs = r.doclens_pfxsum
worst = 0.0
for qi in range(8):
    for c in range(0, 777, 37):
        pid = int(cand[qi, c])
        if pid < 0 or doclens[pid] == 0:
            continue
        o = int(offs[pid])
        mx = (Q[qi, :int(q_len[qi])].double() @ D[o:o + doclens[pid]].T).max(-1).values
        if int(r.d_pad_len[pid]) > doclens[pid]:
            mx = mx.clamp_min(0)
        worst = max(worst, abs(float(mx.sum()) - float(alt[qi, c])))
print("max |alt - f64| on a sample =", worst)
