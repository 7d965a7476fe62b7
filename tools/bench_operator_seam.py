#!/usr/bin/env python3
"""Operator seam on one MI355X: colbert_amd.score(Q, D, q_mask, d_mask) against the stock torch formulation of the same
four ops (BaseModel.py:41-45, written out here -- this tool does not import the oracle) at the reference's per-bucket
call shape (colbert_ranker.py:111-112): Q [1, 32, 128] fp32, D [n, S, 128] fp32 on the device, int64 masks.

    python tools/bench_operator_seam.py [--n 1000 --s 180 --dim 128]
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def torch_score(Q, D, qm, dm):
    D = D * dm[..., None]
    Q = Q * qm[..., None]
    return torch.einsum("qmh,dnh->qdmn", Q, D).max(-1).values.sum(-1)


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--s", type=int, default=180)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    import colbert_amd
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(0)
    Q = F.normalize(torch.randn(1, 32, a.dim, generator=gen, device=dev), dim=-1)
    # a pool of buckets larger than the caches, so every call reads its D from HBM like a fresh gather would
    pool = [F.normalize(torch.randn(a.n, a.s, a.dim, generator=gen, device=dev), dim=-1) for _ in range(8)]
    qm = torch.ones(1, 32, dtype=torch.long, device=dev)
    dm = (torch.arange(a.s, device=dev)[None] < torch.randint(1, a.s + 1, (a.n, 1), generator=gen, device=dev)).long()
    i = [0]

    def ours():
        i[0] += 1
        return colbert_amd.score(Q, pool[i[0] % 8], qm, dm)

    def stock():
        i[0] += 1
        return torch_score(Q, pool[i[0] % 8], qm, dm)

    diff = float((ours() - torch_score(Q, pool[i[0] % 8], qm, dm)).abs().max())
    torch.cuda.reset_peak_memory_stats()
    t_ours = timeit(ours, a.iters)
    m_ours = torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    t_stock = timeit(stock, a.iters)
    m_stock = torch.cuda.max_memory_allocated()
    print(json.dumps({"shape": {"n": a.n, "S": a.s, "dim": a.dim, "dtype": "fp32", "mask": "int64"},
                      "ours_ms_per_call": round(t_ours, 4), "torch_ms_per_call": round(t_stock, 4),
                      "speedup": round(t_stock / t_ours, 2), "max_abs_diff": diff,
                      "peak_bytes_ours": m_ours, "peak_bytes_torch": m_stock}))


if __name__ == "__main__":
    main()
