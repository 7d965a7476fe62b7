#!/usr/bin/env python3
"""The reference's multi-view deployment shapes (proj_conf/dense.yaml:8,29-32 with colbert_ranker.py:62's fp16 storage):
  mv128 : d_view = q_view = 8,  dim 128 (BASELINE configs[3]'s shape on the 16-bit index), 2 KiB per doc
  mv768 : d_view = q_view = 16, dim 768 (the yaml's defaults), 24 KiB per doc
Rerank kernel only: 20 launches of 256 queries x 1000 candidates back to back between two HIP events, per index dtype and
query dtype; a sample of the scores is compared with torch (fp32 math on the stored values).  WL=mv128,mv768 DT=fp16,bf16,fp32
QDT=fp32,fp16 NQ=256 UNIFORM=0 (hide the fixed-length promise from the library: the general kernels)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd  # noqa: E402

TDT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
SHAPES = {"mv128": dict(lq=8, ld=8, h=128, ndocs=4_000_000), "mv768": dict(lq=16, ld=16, h=768, ndocs=1_000_000),
          "mv128x16": dict(lq=16, ld=16, h=128, ndocs=4_000_000), "mv768x8": dict(lq=8, ld=8, h=768, ndocs=1_000_000)}


def main():
    dev = torch.device("cuda", 0)
    nq, ncand = int(os.environ.get("NQ", 256)), int(os.environ.get("NCAND", 1000))
    for name in os.environ.get("WL", "mv128,mv768").split(","):
        s = SHAPES[name]
        lq, ld, h, ndocs = s["lq"], s["ld"], s["h"], int(os.environ.get("NDOCS", s["ndocs"]))
        for dt in os.environ.get("DT", "fp16").split(","):
            dtype = TDT[dt]
            gen = torch.Generator(device=dev).manual_seed(1234)
            idx = torch.empty(ndocs * ld, h, dtype=dtype, device=dev)
            step = max(1, (1 << 28) // h)
            for a in range(0, idx.size(0), step):
                b = min(a + step, idx.size(0))
                idx[a:b] = F.normalize(torch.randn(b - a, h, generator=gen, device=dev), dim=-1).to(dtype)
            ranker = colbert_amd.ColbertRanker.from_device_tensor(idx, [ld] * ndocs)
            if os.environ.get("UNIFORM") == "0":
                ranker._iv.uniform_len = 0
            for qdt in os.environ.get("QDT", "fp32,fp16").split(","):
                if qdt != "fp32" and qdt != dt:
                    continue
                gq = torch.Generator(device=dev).manual_seed(1)
                Q = F.normalize(torch.randn(nq, lq, h, generator=gq, device=dev), dim=-1).to(TDT[qdt])
                if nq * ncand <= ndocs:     # distinct docs within a launch
                    cands = torch.stack([torch.randperm(ndocs, generator=gq, device=dev)[:nq * ncand].view(nq, ncand) for _ in range(4)])
                else:
                    cands = torch.randint(0, ndocs, (4, nq, ncand), generator=gq, device=dev)
                sc = ranker.score_candidates(Q, cands[0])
                # parity sample: fp32 math on the stored values
                qs, cs = Q[:4].float(), cands[0, :4, :64]
                rows = (cs.unsqueeze(-1) * ld + torch.arange(ld, device=dev)).view(4, -1)
                D = idx[rows].float().view(4, 64, ld, h)
                exp = torch.einsum("qmh,qdnh->qdmn", qs, D).max(-1).values.sum(-1)
                err = (sc[:4, :64] - exp).abs().max().item()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                best = None
                for rep in range(3):
                    e0.record()
                    for i in range(20):
                        ranker.score_candidates(Q, cands[i % 4])
                    e1.record()
                    e1.synchronize()
                    ms = e0.elapsed_time(e1) / 20
                    best = ms if best is None else min(best, ms)
                alg = nq * ncand * (ld * h * idx.element_size() + 24) + nq * lq * h * Q.element_size()
                print(f"{name} index {dt} q {qdt} nq {nq}: {best:.4f} ms  {alg / best / 1e6:.0f} GB/s  frac {alg / best / 1e6 / 8000:.3f}  "
                      f"max|err| {err:.2e}", flush=True)
            del ranker, idx
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
