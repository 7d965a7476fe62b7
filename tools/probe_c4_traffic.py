#!/usr/bin/env python3
"""C4 (multi-view: 4 M docs x 8 tokens, 256 queries x 1000 candidates) under `rocprofv3 --pmc ...`: the same rerank launch in
the two settings whose FETCH_SIZE disagreed in round 3 (profiles/r03_c4_f32_pmc.json 1.038 x algorithmic, bench.py's live
sweep 0.996 x): phase A = 4 launches back to back (the sweep child), phase B = 4 launches each followed by the top-k and
bracketed by HIP events (bench.py's timed step), phase C = A again with candidate lists WITHOUT repeated docs inside a launch
(what the L2 cannot absorb), phase D = candidates drawn from 250k docs only (many repeats).  Dispatches are matched afterwards
by kernel name and order (tools/summarize_c4_traffic.py); this script only prints what each phase should have read."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import colbert_amd  # noqa: E402

dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["c4"]
nd, L, H, LQ = wl["ndocs"], wl["ld"], wl["h"], wl["lq"]
if os.environ.get("PAD_FIRST"):                      # shift the index allocation by an odd number of KiB-blocks
    pad = torch.empty(int(os.environ["PAD_FIRST"]), dtype=torch.uint8, device=dev)
idx = bench.build_index(nd * L, H, dev, 1234, torch.float32)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [L] * nd)
g = torch.Generator(device=dev).manual_seed(1)
Q = F.normalize(torch.randn(256, LQ, H, generator=g, device=dev), dim=-1)
gc = torch.Generator(device=dev).manual_seed(2)
phases = []


def alg(c):
    return int(c.numel() * (L * H * 4 + 24) + Q.numel() * 4)


A = torch.randint(0, nd, (4, 256, 1000), generator=gc, device=dev)
for i in range(4):
    r.score_candidates(Q, A[i])
torch.cuda.synchronize()
phases.append({"phase": "A back to back", "launches": 4, "algorithmic": alg(A[0]),
               "distinct_docs": [int(torch.unique(A[i]).numel()) for i in range(4)]})
B = torch.randint(0, nd, (4, 256, 1000), generator=gc, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
for i in range(4):
    ev[2 * i].record()
    s = r.score_candidates(Q, B[i])
    ev[2 * i + 1].record()
    r.topk(s, B[i], 100)
torch.cuda.synchronize()
phases.append({"phase": "B events + top-k between", "launches": 4, "algorithmic": alg(B[0]),
               "distinct_docs": [int(torch.unique(B[i]).numel()) for i in range(4)]})
C = torch.stack([torch.randperm(nd, generator=gc, device=dev)[:256000].view(256, 1000) for _ in range(4)])
for i in range(4):
    r.score_candidates(Q, C[i])
torch.cuda.synchronize()
phases.append({"phase": "C no repeated doc inside a launch", "launches": 4, "algorithmic": alg(C[0]), "distinct_docs": [256000] * 4})
D = torch.randint(0, 250000, (4, 256, 1000), generator=gc, device=dev)
for i in range(4):
    r.score_candidates(Q, D[i])
torch.cuda.synchronize()
phases.append({"phase": "D candidates from 250k docs", "launches": 4, "algorithmic": alg(D[0]),
               "distinct_docs": [int(torch.unique(D[i]).numel()) for i in range(4)]})
print("PHASES " + json.dumps({"index_ptr_mod_4096": idx.data_ptr() % 4096, "index_ptr_mod_2MiB": idx.data_ptr() % (2 << 20), "phases": phases}), flush=True)
