#!/usr/bin/env python3
"""Context number: the reference's rerank loop written with stock torch ops and run on the SAME MI355X with the token
index already resident in HBM (more than the reference does: it gathers on the CPU and copies over PCIe per bucket,
colbert_ranker.py:105-106) -- per query: gather the candidates' tokens, mask-multiply, einsum, max, sum, sort
(BaseModel.py:41-45, colbert_ranker.py:128-130, written out here; this tool does not import the oracle).

    python tools/bench_torch_gpu_rerank.py [--nq 64]
"""
import argparse
import json
import time

import torch
import torch.nn.functional as F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=64)
    ap.add_argument("--ndocs", type=int, default=200_000)
    a = ap.parse_args()
    dev, L, h, ncand = "cuda", 180, 128, 1000
    gen = torch.Generator(device=dev).manual_seed(0)
    idx = torch.empty(a.ndocs * L, h, device=dev)
    for s in range(0, a.ndocs * L, 1 << 21):
        e = min(s + (1 << 21), a.ndocs * L)
        idx[s:e] = F.normalize(torch.randn(e - s, h, generator=gen, device=dev), dim=-1)
    offs = torch.arange(a.ndocs, device=dev) * L
    Q = F.normalize(torch.randn(a.nq, 32, h, generator=gen, device=dev), dim=-1)
    cand = torch.randint(0, a.ndocs, (a.nq, ncand), generator=gen, device=dev)
    ar = torch.arange(L, device=dev)
    qm = torch.ones(1, 32, dtype=torch.long, device=dev)
    dm = torch.ones(ncand, L, dtype=torch.long, device=dev)

    def one(i):
        rows = offs[cand[i]][:, None] + ar[None]
        D = idx[rows] * dm[..., None]
        q = Q[i:i + 1] * qm[..., None]
        s = torch.einsum("qmh,dnh->qdmn", q, D).max(-1).values.sum(-1)[0]
        top = s.sort(descending=True)
        return cand[i][top.indices[:100]], top.values[:100]

    for i in range(3):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.nq):
        one(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(json.dumps({"what": "stock torch ops on MI355X, index in HBM, 1 query x 1000 docs (32x180, dim 128, fp32) per step",
                      "queries": a.nq, "queries_per_s": round(a.nq / el, 1), "ms_per_query": round(el / a.nq * 1e3, 4)}))


if __name__ == "__main__":
    main()
