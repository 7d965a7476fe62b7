#!/usr/bin/env python3
"""Training-form MaxSim (all-pairs score + backward) on one MI355X: colbert_amd.score under autograd against the stock
torch formulation of the same operator (mask-multiply, einsum, max, sum -- the four ops of BaseModel.py:41-45, written
out here; this tool does not import the oracle).  Shape = the reference's training step after its cross-rank all_gather
(colbert_model.py:87-90, eval.sh:17): Q [B*W, 32, dim], D [2*B*W, Ld, dim], fp16/bf16 under autocast.

    python tools/bench_training_form.py [--nq 272 --nd 544 --ld 384 --dim 768 --dtype bf16]
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
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=272)
    ap.add_argument("--nd", type=int, default=544)
    ap.add_argument("--lq", type=int, default=32)
    ap.add_argument("--ld", type=int, default=384)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--dtype", default="bf16", choices=["fp32", "fp16", "bf16"])
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-torch", action="store_true", help="skip the stock-torch comparison (profiling runs)")
    a = ap.parse_args()
    import colbert_amd
    dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    g = torch.Generator(device="cuda").manual_seed(0)
    Q = F.normalize(torch.randn(a.nq, a.lq, a.dim, generator=g, device="cuda"), dim=-1).to(dt)
    D = F.normalize(torch.randn(a.nd, a.ld, a.dim, generator=g, device="cuda"), dim=-1).to(dt)
    qm = torch.ones(a.nq, a.lq, dtype=torch.long, device="cuda")
    dl = torch.randint(a.ld // 4, a.ld + 1, (a.nd, 1), generator=g, device="cuda")
    dm = (torch.arange(a.ld, device="cuda").unsqueeze(0) < dl).long()
    w = torch.randn(a.nq, a.nd, generator=g, device="cuda")
    res = {"shape": vars(a)}

    def run(fn):
        q, d = Q.clone().requires_grad_(True), D.clone().requires_grad_(True)
        out = fn(q, d, qm, dm)
        (out.float() * w).sum().backward()
        return out, q.grad, d.grad

    o1, gq1, gd1 = run(colbert_amd.score)
    torch.cuda.reset_peak_memory_stats()
    res["ours_ms"] = round(timeit(lambda: run(colbert_amd.score), a.iters), 3)
    res["ours_peak_GB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    with torch.no_grad():
        res["ours_fwd_only_ms"] = round(timeit(lambda: colbert_amd.score(Q, D, qm, dm), a.iters), 3)
    try:
        if a.no_torch:
            raise RuntimeError("skipped (--no-torch)")
        o2, gq2, gd2 = run(torch_score)
        torch.cuda.reset_peak_memory_stats()
        res["torch_ms"] = round(timeit(lambda: run(torch_score), a.iters), 3)
        res["torch_peak_GB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
        res["max_abs_diff"] = {"out": float((o1.float() - o2.float()).abs().max()),
                               "dQ": float((gq1.float() - gq2.float()).abs().max()),
                               "dD": float((gd1.float() - gd2.float()).abs().max())}
    except Exception as e:  # the stock path keeps the [q,d,m,n] tensor: it may not fit
        res["torch_error"] = repr(e)[:200]
    flops = 2.0 * a.nq * a.nd * a.lq * a.ld * a.dim
    res["fwd_TFLOPs"] = round(flops / 1e12, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
